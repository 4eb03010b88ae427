#!/bin/bash
out=gpurun_out/r04o; mkdir -p $out
timeout 1200 python -m pytest tests -q -m gpu -x > $out/gpu_tests.log 2>&1; grep -E "passed|failed" $out/gpu_tests.log; grep -E "^FAILED|Error" $out/gpu_tests.log | head -5
