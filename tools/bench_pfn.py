"""(the QV2X_PFN_FORM switch exists in builds with -DQV2X_DEV_KNOBS only: python tools/build_variant.py knobs <source>.hip -DQV2X_DEV_KNOBS, then QV2X_LIB_TAG=knobs)
Dev tool: the PFN + scatter of the bench's batch (32 V2X-Real sweeps), timed by HIP events inside a graph, with a hash of the canvas it leaves.
    QV2X_PFN_FORM=64 python tools/bench_pfn.py     one lane per channel (rounds 1-5)
    QV2X_PFN_FORM=16 python tools/bench_pfn.py     sixteen lanes per pillar (round 6; the default)
The two hashes must be equal (the canvases are compared bit for bit by tests/test_hip_parity.py against the oracle)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from quantv2x_amd import lib as L
if os.environ.get("QV2X_LIB_TAG"):           # a side library from tools/build_variant.py
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{os.environ['QV2X_LIB_TAG']}.so")

dev = torch.device("cuda", 0)
_, eng, _, _ = bench.build_engine(8)
_, full, _, _ = bench.frame_batch(1, 0, 32, dev)
inp = full["inputs_m1"]
n = 32
canvas = eng.pillars_to_canvas(inp, n, resident=True)
torch.cuda.synchronize()
h = hashlib.sha1(canvas.cpu().numpy().tobytes()).hexdigest()
eng.clear_pillars(inp, n)


def both():
    eng.pillars_to_canvas(inp, n, resident=True)
    eng.clear_pillars(inp, n)


both()
torch.cuda.synchronize()
print("form", os.environ.get("QV2X_PFN_FORM", "default"), os.environ.get("QV2X_LIB_TAG", ""), "pillars", int(inp["voxel_features"].shape[0]),
      f"scatter + clear {bench.event_time_us(bench._graph_of(both), 20):.1f} us per 32 sweeps, canvas sha1 {h[:16]}")
