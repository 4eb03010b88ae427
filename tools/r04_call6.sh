#!/bin/bash
out=gpurun_out/r04f; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I quantv2x_amd/csrc -o /tmp/qp tools/probes/q_pack4_probe.hip 2>/dev/null && /tmp/qp > $out/q_pack4_probe.log 2>&1; tail -4 $out/q_pack4_probe.log
timeout 1200 python -m pytest tests -q -m gpu -x > $out/gpu_tests.log 2>&1; grep -E "passed|failed" $out/gpu_tests.log
python tools/bench_kernels.py conv 32 > $out/conv32.log 2>&1; grep -E "conv |total" $out/conv32.log | awk '{print $2, $(NF-3)}' | tr '\n' ';'
