#!/bin/bash
out=gpurun_out/r04n; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_fuse_heads.py tests/test_hip_parity.py tests/test_hip_dist_nccl.py "tests/test_hip_fullsize.py::test_bench_configuration_batch32_two_streams_exact" -q -m gpu -x > $out/tests.log 2>&1; tail -5 $out/tests.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $out/table_heads_times.log
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from quantv2x_amd.engine import deploy
from quantv2x_amd import lib as L
eng = deploy(path="tools/cache/v2xreal_state.npz")
for n in (32, 1):
    full = bench.frame_batch(1, 0, n, torch.device("cuda"))[1]
    eng(full); torch.cuda.synchronize()
    hw = eng.fh * eng.fw
    codes = eng._workspace(n)["codes"]
    pw = full["pairwise_t_matrix"].to(torch.float64).contiguous()
    def general():
        fused = torch.empty((n, hw, 256), dtype=torch.float32, device="cuda")
        eng.fuse_scenes(L.ptr(codes), hw, n * hw, None, pw, [f * hw for f in range(n)], [1] * n, fused)
        return eng._heads_pair(fused, n, codes, n)
    def tables():
        return eng._table_heads_out(codes, n)
    general(); tables(); torch.cuda.synchronize()
    print(f"{n} single-agent frame(s): fuse + heads + *_single heads %.1f us, all heads by tables %.1f us" % (bench.event_time_us(general, 20), bench.event_time_us(tables, 20)))
PY
python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 10 > $out/bench_quick.json 2> $out/bench_quick.err; python -c "
import json; d=json.load(open('$out/bench_quick.json')); print(d['value'], d['ms_per_step'], d['latency_ms_p50'])"
