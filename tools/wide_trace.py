"""Dev tool (GPU box): where a workgroup of the wide conv kernel spends its time (s_memtime stamps, build with -DQV2X_WIDE_TRACE).
    python tools/wide_trace.py <tag> <n_frames>"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantv2x_amd import lib as L
tag, n = sys.argv[1], int(sys.argv[2])
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{tag}.so")
from quantv2x_amd import synth
from quantv2x_amd.engine import deploy
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
import bench
dd = bench.frame_batch(1, 0, n, torch.device("cuda", 0))[1]          # n single-agent frames (a batch), as bench.py builds them
eng(dd); torch.cuda.synchronize()
raw = C.CDLL(L.LIB_PATH)
for (kind, layer, x, h, w, o, c0, macs) in eng.conv_plan(n):
    if kind != "conv" or not (layer.name.startswith("shrinker") or layer.name.endswith((".0.2", ".1.2", ".2.2"))): continue
    torch.cuda.synchronize(); assert raw.qv2x_debug_wide_trace_clear() == 0
    for _ in range(3): eng._conv(layer, x, n, h, w, o)
    torch.cuda.synchronize()
    nb = 8192
    buf = np.zeros((nb, 8), np.int64)
    rc = raw.qv2x_debug_wide_trace(buf.ctypes.data_as(C.c_void_p), nb); assert rc == 0, rc
    buf = buf[buf[:, 0] > 0]
    d = np.diff(buf[:, :5], axis=1).astype(np.float64)   # s_memtime ticks are 100 MHz on gfx9 (10 ns)
    t0 = buf[:, 0].min()
    print(layer.name, "blocks", len(buf), "ticks: prologue %.0f  k-loop %.0f  fold %.0f  epilogue %.0f  | total %.0f  | kernel span %.0f" %
          (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), (buf[:, 4] - buf[:, 0]).mean(), buf[:, 4].max() - t0))
    rt = (buf[:, 7] - buf[:, 6]).astype(np.float64)          # s_memrealtime: 100 MHz
    print("   shader clock seen by the workgroups: %.2f GHz (memtime ticks per 10 ns realtime tick / 10); kernel wall (realtime) %.1f us" %
          (((buf[:, 4] - buf[:, 0]) / rt).mean() / 10.0, (buf[:, 7].max() - buf[:, 6].min()) / 100.0))
    hw = buf[:, 5]
    print("   wave slot histogram:", np.bincount((hw & 15).astype(int)), " simd:", np.bincount(((hw >> 4) & 3).astype(int)), " first 16 blocks slot/cu:", [(int(h & 15), int((h >> 8) & 15)) for h in hw[:16]])
    starts = np.sort(buf[:, 0] - t0)
    print("   start ticks percentiles 0/25/50/75/100:", [int(np.percentile(starts, q)) for q in (0, 25, 50, 75, 100)])
