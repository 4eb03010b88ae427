#!/bin/bash
# round 4, first GPU call: everything green at the new HEAD? + baselines for the kernel work
out=gpurun_out/r04a; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x > $out/gpu_tests.log 2>&1; tail -5 $out/gpu_tests.log
timeout 600 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 1500 $out/bench_n1.json; tail -5 $out/bench_n1.err
for mode in "--link torch" "--link rccl" "--link rccl --graph-link"; do
  tagm=$(echo $mode | tr -d ' -')
  timeout 300 python bench.py --force-sharded $mode --no-extras --no-cpu-baseline --steps 40 --warmup 10 > $out/sharded_$tagm.json 2> $out/sharded_$tagm.err
  python -c "import json,sys; d=json.load(open('$out/sharded_$tagm.json')); print('$mode', d['value'], d['ms_per_step'], d['latency_ms_p50'], d.get('ego_only'))" || tail -5 $out/sharded_$tagm.err
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/cvt tools/probes/cvt_pk_u8_probe.hip 2>/dev/null && /tmp/cvt > $out/cvt_pk_u8_probe.log 2>&1; head -32 $out/cvt_pk_u8_probe.log | tail -5; tail -12 $out/cvt_pk_u8_probe.log
python tools/bench_kernels.py conv 32 > $out/conv32_fa1.log 2>&1; grep -E "shrinker|total" $out/conv32_fa1.log
QV2X_LIB_TAG=nofa1 python tools/bench_kernels.py conv 32 > $out/conv32_nofa1.log 2>&1; grep -E "shrinker|total" $out/conv32_nofa1.log
python tools/bench_kernels.py conv 32 > $out/conv32_fa1_b.log 2>&1; grep -E "shrinker|total" $out/conv32_fa1_b.log
du -sh gpurun_out
