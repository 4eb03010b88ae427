"""Runs the block-0 chain launch a few times (dev tool for rocprofv3 --pmc on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import synth
from quantv2x_amd.engine import deploy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
dd = synth.scene_to_torch(synth.make_scene("v2xreal", n_agents=n, seed=3, n_points=60000), "cuda")
eng(dd); torch.cuda.synchronize()
for (kind, layer, x, h, w, o, c0, macs) in eng.conv_plan(n):
    if kind == "chain":
        for _ in range(5):
            eng._chain(layer, x, n, h, w, o)
torch.cuda.synchronize()
