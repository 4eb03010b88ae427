import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import lib as L
v = sys.argv[1]
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_abl{v}.so")
from quantv2x_amd.engine import deploy
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
def run(h, w, iters=10):
    x = torch.randint(-128, 127, (1, h + 2, w + 2, 256), dtype=torch.int8, device="cuda")
    codes = torch.empty((3, h * w), dtype=torch.uint8, device="cuda")
    d = L.EncodeDesc(); d.n, d.h, d.w, d.levels, d.kc = 1, h, w, 3, 128; d.in_zx, d.in_delta = 0, 0.05
    f = lambda: L.check(eng.lib.qv2x_codebook_encode_f32(C.byref(d), L.ptr(x), eng.level_ptrs, L.ptr(codes), L.current_stream()))
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    print(f"variant {v}: h={h} w={w} blocks={h*w//64} {e0.elapsed_time(e1) / iters * 1e3:.1f} us")
run(100, 352)
