"""Dev tool (GPU box): PFN + scatter (resident canvas) timing with a side library.   python tools/bench_pfn_variant.py <tag | main> <n_frames>"""
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
def timeit(fn, iters=30):
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
def f():
    eng.pillars_to_canvas(dd["inputs_m1"], n, resident=True); eng.clear_pillars(dd["inputs_m1"], n)
print("variant", tag, "n", n, "pfn + scatter + un-scatter us: %.1f" % timeit(f))
