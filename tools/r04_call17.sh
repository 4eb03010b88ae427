#!/bin/bash
out=gpurun_out/r04q; mkdir -p $out
timeout 900 python tools/pytest_with_lib.py woven tests/test_hip_conv_wide.py -q -m gpu -x 2>&1 | tail -3
python tools/bench_kernels.py conv 32 > $out/conv32.log 2>&1; grep -E "conv |total" $out/conv32.log | awk '{print $2, $(NF-3)}' | tr '\n' ';'; echo
QV2X_LIB_TAG=woven python tools/bench_kernels.py conv 32 > $out/conv32_woven.log 2>&1; grep -E "conv |total" $out/conv32_woven.log | awk '{print $2, $(NF-3)}' | tr '\n' ';'; echo
