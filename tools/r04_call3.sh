#!/bin/bash
out=gpurun_out/r04c; mkdir -p $out
export TMPDIR=/tmp
python tools/bench_kernels.py conv 32 > $out/conv32_ws.log 2>&1; echo "ws:   " $(grep -E "blocks.0.[234]" $out/conv32_ws.log | awk '{print $7}')
for k in 1 2 3 4 5; do
QV2X_LIB_TAG=wsabl$k python tools/bench_kernels.py conv 32 > $out/conv32_wsabl$k.log 2>&1; echo "abl$k: " $(grep -E "blocks.0.[234]" $out/conv32_wsabl$k.log | awk '{print $7}')
done
