"""Dev tool (GPU box): phase stamps of the pixel-stationary deconv items (build: tools/build_variant.py dpsfine deconv_f32.hip -DQV2X_DPS_FINE).
    python tools/dps_fine.py <tag> <n_frames>      s_memtime ticks of the first 1024 items of every layer"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantv2x_amd import lib as L
tag, n = sys.argv[1], int(sys.argv[2])
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{tag}.so")
from quantv2x_amd.engine import deploy
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
import bench
dd = bench.frame_batch(1, 0, n, torch.device("cuda", 0))[1]
eng(dd); torch.cuda.synchronize()
items = [(layer, x, h, w, out, c0) for (kind, layer, x, h, w, out, c0, macs) in eng.conv_plan(n) if kind == "deconv"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record(); eng._deconv_batch(items, n); e1.record()
torch.cuda.synchronize()
raw = C.CDLL(L.LIB_PATH)
buf = np.zeros((3, 1024, 20), np.int64)
assert raw.qv2x_debug_dps_fine(buf.ctypes.data_as(C.c_void_p)) == 0
print(f"{n}-frame launch: {e0.elapsed_time(e1) * 1e3:.0f} us")
for li, (K, NP) in enumerate([(256, 8), (128, 8), (64, 2)]):                     # (pairs per item at a batch of 32 frames)
    b = buf[li]
    b = b[b[:, 0] > 0]
    d = np.diff(b[:, :2 + 2 * NP], axis=1)
    names = ["start -> operands ready"] + [f"{'K loop' if i % 2 == 0 else 'epilogue'} {i // 2}" for i in range(2 * NP)]
    t0 = buf[buf > 0].min()
    print(f"   {len(b)} sampled items; start times (ticks after the launch's first stamp) min {b[:, 0].min() - t0} median {int(np.median(b[:, 0])) - t0} max {b[:, 0].max() - t0}; end max {b[:, 1 + 2 * NP].max() - t0}")
    print(f"Cin {K}: item {(b[:, 1 + 2 * NP] - b[:, 0]).mean():.0f} ticks (MFMA work {NP * K * 64} cycles); " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, d.mean(axis=0))))
