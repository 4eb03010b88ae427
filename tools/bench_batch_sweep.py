"""Stage timings and graph-replay throughput by batch size (dev tool, runs on the GPU box).

    python tools/bench_batch_sweep.py [B ...]          default 8 16 32

Per batch size B: the `roofline_stages` table of bench.py (live HIP-event time of every stage against its own bound) and
frames/s of the whole path replayed as one HIP graph per batch, with 1 and 2 batches in flight."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from quantv2x_amd.engine import deploy
from quantv2x_amd.ptq_state import load_ptq_state

state = load_ptq_state(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or [8, 16, 32]:
    eng = deploy(state=state)
    _, full, _, _ = bench.frame_batch(1, 0, B, dev)
    out = {"batch": B}
    for F in (1, 2):
        engines = [eng] + [deploy(state=state) for _ in range(F - 1)]
        streams = [torch.cuda.Stream() for _ in range(F)]
        reps = []
        for e, st in zip(engines, streams):
            with torch.cuda.stream(st):
                reps.append(e.capture(full))
        torch.cuda.synchronize()
        steps = max(16, 400 // B)
        for i in range(2 * F):
            with torch.cuda.stream(streams[i % F]):
                reps[i % F]()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % F]):
                reps[i % F]()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"frames_per_s_inflight{F}"] = round(B * steps / dt, 1)
        del reps, engines
    _, stages = bench.rooflines(eng, full, B, iters=10)
    out["stages"] = {k: {"us_per_frame": round(v["us_per_batch"] / B, 2), "frac": v["frac"]} for k, v in stages.items() if "us_per_batch" in v}
    print(json.dumps(out), flush=True)
    del eng
    torch.cuda.empty_cache()
