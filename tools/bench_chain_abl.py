"""Chain kernel ablations (dev tool, GPU box): builds libqv2x variants with -DQV2X_CHAIN_DBG=n and times the block-0 chain."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import lib as L, build as B
v = sys.argv[1]
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_chain{v}.so")
if not os.path.exists(L.LIB_PATH):
    os.makedirs(os.path.dirname(L.LIB_PATH), exist_ok=True)
    subprocess.check_call(["hipcc"] + B.FLAGS + [f"-DQV2X_CHAIN_DBG={v}", "-o", L.LIB_PATH] + B.SOURCES)
from quantv2x_amd import synth
from quantv2x_amd.engine import deploy
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
dd = synth.scene_to_torch(synth.make_scene("v2xreal", n_agents=1, seed=3, n_points=60000), "cuda")
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
for (kind, layer, x, h, w, o, c0, macs) in eng.conv_plan(1):
    if kind == "chain":
        print("variant", v, "chain us: %.1f" % timeit(lambda: eng._chain(layer, x, 1, h, w, o)))
