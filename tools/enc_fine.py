"""Dev tool (GPU box): phase stamps of the exact codebook-encode kernel (build: tools/build_variant.py encfine codebook_encode.hip -DQV2X_ENC_FINE).
    python tools/enc_fine.py <tag> <n_frames>      s_memtime ticks (a constant ~2.07 GHz counter; x 1.14 = shader cycles at 2.35 GHz)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantv2x_amd import lib as L
tag, n = sys.argv[1], int(sys.argv[2])
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{tag}.so")
from quantv2x_amd.engine import deploy
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
h, w = 100, 352
x = torch.randint(-128, 127, (n, h + 2, w + 2, 256), dtype=torch.int8, device="cuda")
codes = torch.empty((3, n * h * w), dtype=torch.uint8, device="cuda")
d = L.EncodeDesc(); d.n, d.h, d.w, d.levels, d.kc = n, h, w, 3, 128; d.in_zx, d.in_delta = 0, 0.05
for _ in range(3):
    L.check(eng.lib.qv2x_codebook_encode_f32(C.byref(d), L.ptr(x), eng.level_ptrs, L.ptr(codes), L.current_stream()))
torch.cuda.synchronize()
raw = C.CDLL(L.LIB_PATH)
nb = min(2048, n * 1100)
buf = np.zeros((nb, 40), np.int64)
assert raw.qv2x_debug_encode_fine(buf.ctypes.data_as(C.c_void_p), nb) == 0
names = ["stage GEMM", "store z + barrier", "qhead GEMM", "store q + barrier", "|q|^2 chains + barrier", "x2 + barrier", "distance GEMM", "key reduce + barrier",
         "argmin + barrier", "lhead GEMM", "residual + barrier"]
tot = buf[:, 33] - buf[:, 0]
print(f"{nb} workgroups of a {n}-frame launch: whole workgroup (level 0 start .. level 2 argmin) mean {tot.mean():.0f} ticks, min {tot.min()}, max {tot.max()}")
for l in range(3):
    b = buf[:, 12 * l:12 * l + 12]
    k = 12 if l < 2 else 10
    dd = np.diff(b[:, :k], axis=1)
    print(f"level {l}: " + ", ".join(f"{names[i]} {dd[:, i].mean():.0f}" for i in range(k - 1)))
