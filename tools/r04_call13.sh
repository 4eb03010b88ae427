#!/bin/bash
out=gpurun_out/r04m; mkdir -p $out
timeout 1500 python -m pytest tests/test_heter.py tests/test_hip_fuse_heads.py tests/test_maxfuse.py "tests/test_hip_fullsize.py::test_v2xreal_mixed_encoder_scene_exact" "tests/test_hip_fullsize.py::test_opv2v_mixed_encoders_eight_agents_exact" -q -m gpu -x --durations=5 > $out/tests.log 2>&1; tail -15 $out/tests.log
