#!/bin/bash
out=gpurun_out/r04h; mkdir -p $out
python tools/enc_fine.py encfine 32 2>&1 | grep -v amdgpu.ids | tee $out/enc_fine_b32.log
python tools/bench_kernels.py encode 32 2>&1 | grep -v amdgpu.ids | tee -a $out/enc_fine_b32.log
