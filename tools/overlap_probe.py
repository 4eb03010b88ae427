"""Dev tool (GPU box): does the int8 stack of one batch run BESIDE the codebook encode of another (two streams)?
    python tools/overlap_probe.py [<lib tag>] [<frames>]      -- encode alone, the conv / deconv plan alone, both at once (wall time)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import lib as L
tag = sys.argv[1] if len(sys.argv) > 1 else ""
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if tag:
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{tag}.so")
from quantv2x_amd.engine import deploy
import bench
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
dd = bench.frame_batch(1, 0, n, torch.device("cuda", 0))[1]
eng(dd); torch.cuda.synchronize()
prio = int(os.environ.get("PRIO", "0"))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=prio)          # PRIO=-1: the int8 stack on a high-priority stream


def graph_of(fn, stream):
    with torch.cuda.stream(stream):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    return g


ga = graph_of(lambda: eng.encode_codes(n), sa)
gb = graph_of(lambda: eng.run_plan(n), sb)


def wall(graphs, reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g, s in graphs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


a, b, ab = wall([(ga, sa)]), wall([(gb, sb)]), wall([(ga, sa), (gb, sb)])
print(f"lib '{tag}', {n} frames: encode alone {a:.2f} ms, convs + deconvs + shrinker alone {b:.2f} ms, both on two streams {ab:.2f} ms (sum {a + b:.2f})")
