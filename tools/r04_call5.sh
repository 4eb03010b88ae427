#!/bin/bash
out=gpurun_out/r04e; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_conv_wide.py -q -m gpu -x > $out/conv_wide_tests.log 2>&1; tail -3 $out/conv_wide_tests.log
python tools/bench_kernels.py conv 32 > $out/conv32_ws.log 2>&1; grep -E "blocks.0|total" $out/conv32_ws.log
python tools/ws_fine.py wsfine 32 2>&1 | grep -v "amdgpu.ids" | tee -a $out/ws_fine.log
