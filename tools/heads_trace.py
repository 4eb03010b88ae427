"""Dev tool (GPU box): phases of the heads kernel's workgroups (build with -DQV2X_HEADS_TRACE).   python tools/heads_trace.py <tag> <n_frames>"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantv2x_amd import lib as L
tag, n = sys.argv[1], int(sys.argv[2])
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{tag}.so")
from quantv2x_amd import synth
from quantv2x_amd.engine import deploy
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
dd = synth.scene_to_torch(synth.make_scene("v2xreal", n_agents=n, seed=3, n_points=60000), "cuda")
eng(dd); torch.cuda.synchronize()
hw = eng.fh * eng.fw
codes = eng._workspace(n)["codes"]
fused = torch.randn((n, hw, 256), dtype=torch.float32, device="cuda")
for _ in range(3): eng._heads_pair(fused, n, codes, n)
torch.cuda.synchronize()
def timeit(fn, iters=20):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("pair launch us (in a graph): %.1f" % timeit(lambda: eng._heads_pair(fused, n, codes, n)))
eng._heads_pair(fused, n, codes, n); torch.cuda.synchronize()
raw = C.CDLL(L.LIB_PATH)
tiles = (n * hw + 31) // 32
tiles = (tiles + 7) // 8 * 8
nb = min(32768, 2 * tiles)
buf = np.zeros((nb, 6), np.int64)
assert raw.qv2x_debug_heads_trace(buf.ctypes.data_as(C.c_void_p), nb) == 0
for job, name in ((0, "fused-map heads (rows by DMA, 3 column tiles)"), (1, "*_single heads (rows decoded, 1 column tile)")):
    b = buf[job * tiles:(job + 1) * tiles][: max(0, nb - job * tiles)]
    b = b[b[:, 0] > 0]
    if len(b) == 0: continue
    d = np.diff(b[:, :5], axis=1).astype(np.float64)
    print(name, "blocks", len(b), "cycles: rows in %.0f  gemm %.0f  barrier %.0f  quantize+store %.0f | total %.0f" %
          (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), (b[:, 4] - b[:, 0]).mean()))
