import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import lib as L
from quantv2x_amd.engine import deploy
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
h, w = int(sys.argv[1]), int(sys.argv[2])
x = torch.randint(-128, 127, (1, h + 2, w + 2, 256), dtype=torch.int8, device="cuda")
codes = torch.empty((3, h * w), dtype=torch.uint8, device="cuda")
d = L.EncodeDesc(); d.n, d.h, d.w, d.levels, d.kc = 1, h, w, 3, 128; d.in_zx, d.in_delta = 0, 0.05
for _ in range(5):
    L.check(eng.lib.qv2x_codebook_encode_f32(C.byref(d), L.ptr(x), eng.level_ptrs, L.ptr(codes), L.current_stream()))
torch.cuda.synchronize()
