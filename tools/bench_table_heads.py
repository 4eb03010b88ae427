# the QV2X_TABLE_HEADS_CELLS switch exists in builds with -DQV2X_DEV_KNOBS only (tools/build_variant.py ... -DQV2X_DEV_KNOBS; QV2X_LIB_TAG)
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np, ctypes as C
import bench
from quantv2x_amd import lib as L
if os.environ.get("QV2X_LIB_TAG"):           # a side library from tools/build_variant.py
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{os.environ['QV2X_LIB_TAG']}.so")
lib = L.load()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
hw, n = 35200, 32
for (planes, kc, c0, c1) in ((3, 128, 72, 20),):
    R = n * hw; CT = c0 + c1
    codes = torch.randint(0, kc, (planes, R), dtype=torch.uint8, device=dev)
    tab = torch.randn((planes, kc, CT), device=dev); b = torch.randn(CT, device=dev); da = torch.full((CT,), 0.05, device=dev); za = torch.full((CT,), 128.0, device=dev)
    o0 = torch.empty((n, c0, hw), device=dev); o1 = torch.empty((n, c1, hw), device=dev)
    f = lambda: L.check(lib.qv2x_table_heads_f32(L.ptr(codes), R, hw, planes, kc, c0, c1, L.ptr(tab), L.ptr(b), L.ptr(da), L.ptr(za), L.ptr(o0), L.ptr(o1), L.current_stream()), "t")
    f(); torch.cuda.synchronize()
    print(planes, kc, f"{bench.event_time_us(bench._graph_of(f), 10):.1f} us per 32 frames")
    import hashlib
    print("cells per lane", os.environ.get("QV2X_TABLE_HEADS_CELLS", "default"), "out sha1", hashlib.sha1(o0.cpu().numpy().tobytes() + o1.cpu().numpy().tobytes()).hexdigest()[:16])
