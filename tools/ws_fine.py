"""Dev tool (GPU box): phase stamps of the weights-stationary conv kernel's third item per workgroup (build with -DQV2X_WS_FINE).
    python tools/ws_fine.py <tag> <n_frames>"""
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
names = ["vmcnt wait", "barrier 1", "psum + DMA issue + zero", "row 0", "lgkm + barrier 2", "row 1 (+ rowsums)", "row 2", "row 3", "row 4"]
for (kind, layer, x, h, w, o, c0, macs) in eng.conv_plan(n):
    if kind != "conv" or not layer.name.endswith((".0.2", ".0.3")): continue
    torch.cuda.synchronize(); assert raw.qv2x_debug_ws_fine_clear() == 0
    for _ in range(2): eng._conv(layer, x, n, h, w, o)
    torch.cuda.synchronize()
    nb = 4096
    buf = np.zeros((nb, 16), np.int64)
    assert raw.qv2x_debug_ws_fine(buf.ctypes.data_as(C.c_void_p), nb) == 0
    buf = buf[(buf[:, 0] > 0) & (buf[:, 9] > 0)]
    if not len(buf):
        print(layer.name, "no workgroup reached a third item"); continue
    d = np.diff(buf[:, :10], axis=1).astype(np.float64)
    print(f"{layer.name}: {len(buf)} workgroups; third item {np.mean(buf[:, 9] - buf[:, 0]):.0f} ticks = " +
          ", ".join(f"{names[i]} {d[:, i].mean():.0f}" for i in range(9)))
