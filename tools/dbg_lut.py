import sys; sys.path.insert(0, ".")
import torch, ctypes as C
from quantv2x_amd import lib as L
l = L.load()
torch.manual_seed(0)
for (R, levels, kc) in [(8, 1, 4), (8, 3, 128), (1024, 3, 128)]:
    lut = torch.randn(levels, kc, 256, device="cuda"); bias = torch.randn(256, device="cuda")
    codes = torch.randint(0, kc, (levels, R), device="cuda", dtype=torch.uint8)
    out = torch.zeros(R, 256, device="cuda")
    rc = l.qv2x_decode_lut_f32(L.ptr(codes), R, levels, kc, L.ptr(lut), L.ptr(bias), L.ptr(out), L.current_stream())
    torch.cuda.synchronize()
    exp = bias[None] + sum(lut[i][codes[i].long()] for i in range(levels))
    print(R, levels, kc, rc, float((out - exp).abs().max()))
    if R == 8 and levels == 1:
        print(out[0, :8]); print(exp[0, :8]); print(bias[:8]); print(lut[0, codes[0,0].item(), :8])
