"""Throughput with F frames in flight (F engines, F graphs on F streams).  Dev tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import synth
from quantv2x_amd.engine import deploy
from quantv2x_amd.ptq_state import load_ptq_state
state = load_ptq_state(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
dd = synth.scene_to_torch(synth.make_scene("v2xreal", n_agents=1, seed=3, n_points=60000), "cuda")
for F in (1, 2, 3, 4):
    engs = [deploy(state=state) for _ in range(F)]
    streams = [torch.cuda.Stream() for _ in range(F)]
    reps = []
    for e, st in zip(engs, streams):
        with torch.cuda.stream(st):
            reps.append(e.capture(dd))
    torch.cuda.synchronize()
    K = 100
    for _ in range(10):
        for r, st in zip(reps, streams):
            with torch.cuda.stream(st): r()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        for r, st in zip(reps, streams):
            with torch.cuda.stream(st): r()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"frames in flight {F}: {F*K/dt:.1f} frames/s  ({dt/K*1e3:.3f} ms per round)")
    del engs, reps
