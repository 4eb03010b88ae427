"""Timings of the general (multi-agent) a7-a11 path at V2X-Real size (dev tool, runs on the GPU box): S scenes of N agents on the ring / line
layouts of synth.agent_poses, random code planes.      python tools/bench_fuse.py [scenes] [layout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from quantv2x_amd import synth
from quantv2x_amd import lib as L
if os.environ.get("QV2X_LIB_TAG"):
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{os.environ['QV2X_LIB_TAG']}.so")
from quantv2x_amd.engine import deploy

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
layout = sys.argv[2] if len(sys.argv) > 2 else "ring"
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
eng._workspace(1)
hw = eng.fh * eng.fw
dev = torch.device("cuda", 0)


def timeit(fn, iters=10):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


rng = np.random.default_rng(0)
for n in (1, 2, 4, 5, 8):
    Lc = max(n, 5)
    pw = torch.from_numpy(np.tile(synth.pairwise_t_matrix(synth.agent_poses(n, layout), Lc)[None], (S, 1, 1, 1, 1))).to(dev)
    # code planes in the engine's order [levels][agent-frames][H*W]; scene s = agents s*n .. s*n + n - 1
    codes = torch.from_numpy(rng.integers(0, eng.kc, size=(eng.levels, S * n, hw), dtype=np.uint8)).to(dev)
    fused = torch.empty((S, hw, 256), dtype=torch.float32, device=dev)
    offs, cnts = [s * n * hw for s in range(S)], [n] * S
    us_f = timeit(lambda: eng.fuse_scenes(L.ptr(codes), hw, S * n * hw, None, pw, offs, cnts, fused))
    us_h = timeit(lambda: eng._run_heads(eng.heads, fused, S, hw))
    us_fh = timeit(lambda: eng.fuse_heads_scenes(L.ptr(codes), hw, S * n * hw, None, pw, offs, cnts))
    print(f"agents {n} scenes {S} layout {layout}: fuse {us_f:8.1f} us ({us_f / S:6.1f} per scene, {us_f / S / n:5.1f} per agent)   heads {us_h:8.1f} us   "
          f"both in one launch {us_fh:8.1f} us ({us_fh - us_f - us_h:+.1f})", flush=True)
