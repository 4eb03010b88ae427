#!/bin/bash
out=gpurun_out/r04i; mkdir -p $out
python tools/bench_kernels.py conv 32 > $out/conv32.log 2>&1; grep -E "conv |total" $out/conv32.log | awk '{print $2, $(NF-3)}' | tr '\n' ';'; echo
QV2X_LIB_TAG=bn128 python tools/bench_kernels.py conv 32 > $out/conv32_bn128.log 2>&1; grep -E "conv |total" $out/conv32_bn128.log | awk '{print $2, $(NF-3)}' | tr '\n' ';'; echo
QV2X_LIB_TAG=bn128 timeout 600 python tools/pytest_with_lib.py bn128 tests/test_hip_conv_wide.py -q -m gpu -x 2>&1 | tail -2
