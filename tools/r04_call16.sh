#!/bin/bash
out=gpurun_out/r04p; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_codebook_full_golden.py tests/test_hip_collapsed_encode.py "tests/test_hip_fullsize.py::test_v2xreal_frames_exact" "tests/test_hip_fullsize.py::test_opv2v_single_agent_frame_exact" -q -m gpu -x > $out/tests.log 2>&1; tail -3 $out/tests.log
for n in 32 8 2 1; do python tools/bench_kernels.py encode $n 2>&1 | grep encode; done | tee $out/encode_times.log
