"""Throughput with F independent frames in flight (dev tool): python tools/bench_inflight.py [F ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import synth
from quantv2x_amd.engine import deploy
from quantv2x_amd.ptq_state import load_ptq_state
import bench

state = load_ptq_state(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
dd = synth.scene_to_torch(synth.make_scene("v2xreal", n_agents=1, seed=3, n_points=60000), "cuda")
for f in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    print(f"{f} frames in flight: {bench.frames_in_flight(state, dd, 200, f)} frames/s", flush=True)
