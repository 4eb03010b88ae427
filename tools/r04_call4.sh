#!/bin/bash
out=gpurun_out/r04d; mkdir -p $out
for t in wsfine wsfine1 wsfine2; do echo "== $t"; python tools/ws_fine.py $t 32 2>&1 | grep -v "amdgpu.ids" | tee -a $out/ws_fine.log; done
