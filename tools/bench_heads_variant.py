"""Dev tool (GPU box): heads pair launch timing with a side library.   python tools/bench_heads_variant.py <tag | main> <n_frames>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import lib as L
tag, n = sys.argv[1], int(sys.argv[2])
if tag != "main":
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{tag}.so")
from quantv2x_amd import synth
from quantv2x_amd.engine import deploy
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
dd = synth.scene_to_torch(synth.make_scene("v2xreal", n_agents=n, seed=3, n_points=60000), "cuda")
eng(dd); torch.cuda.synchronize()
hw = eng.fh * eng.fw
codes = eng._workspace(n)["codes"]
fused = torch.randn((n, hw, 256), dtype=torch.float32, device="cuda")
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
print("variant", tag, "n", n, "heads pair us: %.1f" % timeit(lambda: eng._heads_pair(fused, n, codes, n)))
