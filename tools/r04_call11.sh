#!/bin/bash
out=gpurun_out/r04k; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -o /tmp/weave tools/probes/mfma_valu_weave_probe.hip 2>/dev/null && /tmp/weave | tee $out/mfma_valu_weave_probe.log
