#!/bin/bash
out=gpurun_out/r04l; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_fuse_heads.py tests/test_hip_parity.py tests/test_maxfuse.py -q -m gpu -x > $out/fuse_heads_tests.log 2>&1; tail -5 $out/fuse_heads_tests.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $out/fuse_heads_times.log
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench
from quantv2x_amd.engine import deploy
eng = deploy(path="tools/cache/v2xreal_state.npz")
n = 32
full = bench.frame_batch(1, 0, n, torch.device("cuda"))[1]
eng(full); torch.cuda.synchronize()
hw = eng.fh * eng.fw
codes = eng._workspace(n)["codes"]
pw = full["pairwise_t_matrix"].to(torch.float64).contiguous()
from quantv2x_amd import lib as L
def two():
    fused = torch.empty((n, hw, 256), dtype=torch.float32, device="cuda")
    eng.fuse_scenes(L.ptr(codes), hw, n * hw, None, pw, [f * hw for f in range(n)], [1] * n, fused)
    return eng._run_heads(eng.heads, fused, n, hw)
def one():
    return eng.fuse_heads_scenes(L.ptr(codes), hw, n * hw, None, pw, [f * hw for f in range(n)], [1] * n)
a, b = two(), one(); torch.cuda.synchronize()
print("equal:", torch.equal(a, b))
print("two launches (fuse + heads): %.1f us per batch of 32" % bench.event_time_us(two, 20))
print("one launch: %.1f us" % bench.event_time_us(one, 20))
# multi-agent: 8 scenes of 4 agents
full4 = bench.frame_batch(4, 0, 8, torch.device("cuda"))[1]
eng(full4); torch.cuda.synchronize()
codes4 = eng._workspace(32)["codes"]
pw4 = full4["pairwise_t_matrix"].to(torch.float64).contiguous()
def two4():
    fused = torch.empty((8, hw, 256), dtype=torch.float32, device="cuda")
    eng.fuse_scenes(L.ptr(codes4), hw, 32 * hw, None, pw4, [4 * f * hw for f in range(8)], [4] * 8, fused)
    return eng._run_heads(eng.heads, fused, 8, hw)
def one4():
    return eng.fuse_heads_scenes(L.ptr(codes4), hw, 32 * hw, None, pw4, [4 * f * hw for f in range(8)], [4] * 8)
a, b = two4(), one4(); torch.cuda.synchronize()
print("4 agents x 8 scenes equal:", torch.equal(a, b), " two launches %.1f us, one launch %.1f us" % (bench.event_time_us(two4, 20), bench.event_time_us(one4, 20)))
PY
