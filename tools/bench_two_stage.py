"""Dev tool: the two-stage exact encode against the exact encode at the bench's batch (32 V2X-Real frames of golden-like rows), HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bench
from quantv2x_amd import lib as L

def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    state, eng, _, _ = bench.build_engine(min(32, os.cpu_count() or 8))
    _, full, _, _ = bench.frame_batch(1, 0, frames, torch.device("cuda"))
    eng.encode_mode = "exact"
    eng(full)                                              # fills the workspace's shrinker output with real rows
    torch.cuda.synchronize()
    exact = eng.encode_codes(frames).clone()
    res = {}
    for mode in ("exact", "two_stage"):
        eng.encode_mode = mode
        for _ in range(3): eng.encode_codes(frames)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = eng.encode_codes(frames)
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 10
        if mode == "two_stage":
            print("equal to exact:", bool(torch.equal(out, exact)), eng.encode_refine_stats(frames))
    print(f"frames {frames}: exact {res['exact']:.3f} ms, two-stage {res['two_stage']:.3f} ms")

main()
