#!/bin/bash
out=gpurun_out/r04j; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -o /tmp/roles tools/probes/mfma_valu_roles_probe.hip 2>/dev/null && /tmp/roles | tee $out/mfma_valu_roles_probe.log
