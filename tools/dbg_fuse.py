import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from _common import calibrated_plugin, scene_np
from oracle.spec import Oracle
from oracle import geometry
from quantv2x_amd.engine import deploy
from quantv2x_amd.ptq_state import export_ptq_state
from quantv2x_amd import synth
state = export_ptq_state(calibrated_plugin())
orc, eng = Oracle(state), deploy(state=state)
sc = scene_np(2)
ot, gt = {}, {}
want = orc.forward(sc, ot); got = eng(synth.scene_to_torch(sc, "cuda"), gt)
f_g = gt["features"].cpu().numpy().reshape(ot["features"].shape)
print("features diff", np.abs(f_g - ot["features"]).max())
fu_g = gt["fused"].cpu().numpy().reshape(ot["fused"].shape)
print("fused diff", np.abs(fu_g - ot["fused"]).max())
# per agent warp check: fuse with 1 agent = warp identity
print("fused vs feature0 (gpu)", np.abs(fu_g[0] - f_g[0]).max(), " oracle:", np.abs(ot["fused"][0] - ot["features"][0]).max())
print(fu_g[0,0,0,:4], ot["fused"][0,0,0,:4], f_g[0,0,0,:4], f_g[1,0,0,:4])
codes = gt["codes"]
L_, n_, hw_ = codes.shape
exp = eng.lut_bias[None, :] + sum(eng.lut[l][codes[l].reshape(-1).long()] for l in range(L_))
print("gpu kernel vs torch gather", float((gt["features"].reshape(-1, 256) - exp).abs().max()))
print("oracle vs torch gather", np.abs(ot["features"].reshape(-1, 256) - exp.cpu().numpy()).max())
print("lut equal", np.abs(orc.lut - eng.lut.cpu().numpy()).max(), np.abs(orc.lut_bias - eng.lut_bias.cpu().numpy()).max())
print("codes equal", (ot["codes"].reshape(codes.shape) == codes.cpu().numpy()).all())
od = orc.decode(ot["codes"].reshape(3, -1))
print("oracle decode again vs torch", np.abs(od - exp.cpu().numpy()).max(), ot["features"].shape)
g = gt["features"].reshape(-1, 256)
c = [codes[l].reshape(-1).long() for l in range(3)]
T = eng.lut; b = eng.lut_bias
cands = {"b": b[None].expand_as(g), "b+T0": b + T[0][c[0]], "b+T0+T1": b + T[0][c[0]] + T[1][c[1]], "T0+T1+T2": T[0][c[0]] + T[1][c[1]] + T[2][c[2]],
         "b+T0[c1]..": b + T[0][c[1]] + T[1][c[2]] + T[2][c[0]], "b + T0[c0]*3": b + 3*T[0][c[0]]}
for k, v in cands.items(): print(k, float((g - v).abs().max()))
r = 5
print(g[r, :8]); print(exp[r, :8]); print((g[r]-exp[r])[:8])
# which rows are wrong
bad = ((g - exp).abs().max(dim=1)[0] > 1e-5)
print("bad rows", int(bad.sum()), "of", bad.numel(), bad.nonzero()[:10].flatten().tolist())
print("---- decode_rows called again after forward")
again = eng.decode_rows(gt["codes"], codes.shape[1] * codes.shape[2])
torch.cuda.synchronize()
print("again vs gather", float((again - exp).abs().max()), "again vs first", float((again - g).abs().max()))
print(eng.levels, eng.kc, eng.lut.shape, eng.lut.dtype, eng.lut.is_contiguous(), eng.lut_bias.shape, gt["codes"].shape, gt["codes"].dtype, gt["codes"].is_contiguous())
c2 = gt["codes"].clone()
again2 = eng.decode_rows(c2, codes.shape[1] * codes.shape[2]); torch.cuda.synchronize()
print("again2 (cloned codes) vs gather", float((again2 - exp).abs().max()))
