#!/bin/bash
out=gpurun_out/r04b; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_conv_wide.py -q -m gpu -x > $out/conv_wide_tests.log 2>&1; tail -5 $out/conv_wide_tests.log
python tools/bench_kernels.py conv 32 > $out/conv32_ws.log 2>&1; grep -E "blocks.0|total" $out/conv32_ws.log
QV2X_LIB_TAG=nows python tools/bench_kernels.py conv 32 > $out/conv32_nows.log 2>&1; grep -E "blocks.0|total" $out/conv32_nows.log
python tools/bench_kernels.py conv 32 > $out/conv32_ws_b.log 2>&1; grep -E "blocks.0|total" $out/conv32_ws_b.log
