"""Dev tool (GPU box): finer phase stamps of the wide conv kernel's third item per workgroup (build with -DQV2X_WIDE_FINE -DQV2X_DEV_KNOBS).
    QV2X_WIDE_PP=0|1 python tools/wide_fine.py <tag> <n_frames>"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantv2x_amd import lib as L
tag, n = sys.argv[1], int(sys.argv[2])
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{tag}.so")
from quantv2x_amd.engine import deploy
import bench
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
dd = bench.frame_batch(1, 0, n, torch.device("cuda", 0))[1]
eng(dd); torch.cuda.synchronize()
raw = C.CDLL(L.LIB_PATH)
names = ["item start", "acc zero + first reads", "K taps 0-7 + half of 8", "vmcnt wait", "barrier", "add_psum + issue_halo", "2nd half of tap 8 (K end)",
         "window sums", "requant math", "swaps + stores issued", "slot barrier"]
for (kind, layer, x, h, w, o, c0, macs) in eng.conv_plan(n):
    if kind != "conv" or not (layer.name.startswith("shrinker") or layer.name.endswith((".0.2", ".1.2", ".2.2"))): continue
    torch.cuda.synchronize(); assert raw.qv2x_debug_wide_fine_clear() == 0
    for _ in range(2): eng._conv(layer, x, n, h, w, o)
    torch.cuda.synchronize()
    nb = 4096
    buf = np.zeros((nb, 16), np.int64)
    assert raw.qv2x_debug_wide_fine(buf.ctypes.data_as(C.c_void_p), nb) == 0
    buf = buf[(buf[:, 0] > 0) & (buf[:, 10] > 0)]
    if not len(buf):
        print(layer.name, "no workgroup reached a third item"); continue
    d = np.diff(buf[:, :11], axis=1).astype(np.float64)
    print(f"{layer.name}: {len(buf)} workgroups; third item {np.mean(buf[:, 10] - buf[:, 0]):.0f} cycles = " +
          ", ".join(f"{names[i + 1]} {d[:, i].mean():.0f}" for i in range(10)))
