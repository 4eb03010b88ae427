#!/bin/bash
# Round evidence on the GPU box, in three separate gpurun calls (raw profiler output is deleted before the call ends: gpurun_out/ is capped at 64 MiB):
#   gpurun --timeout 1500 -- 'bash tools/final_evidence.sh r04 tests'      GPU tests + the bench line
#   gpurun --timeout 900  -- 'bash tools/final_evidence.sh r04 trace'      rocprofv3 --kernel-trace --stats of the bench command
#   gpurun --timeout 1500 -- 'bash tools/final_evidence.sh r04 pmc'        FETCH_SIZE, WRITE_SIZE, MFMA-busy passes
# The trace runs the bench with ONE batch in flight (--inflight 1): with two streams the launches of one stream wait behind the other's
# and the trace's average duration is not the kernel's time (17.6 ms "average" inside a 16.5 ms step).  A second trace at
# the default two batches in flight is kept beside it.  The PMC passes run --steps 4 --warmup 1 (the counters are per launch).
tag=${1:-r04}; what=${2:-tests}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ $what = tests ]; then
  timeout 1200 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
  python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 300 $out/bench_n1.json
fi
if [ $what = trace ]; then
  rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o bench -- python3 bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5 --inflight 1 > $out/bench_n1_under_rocprof_inflight1.json 2> $out/prof1.err
  db=$(find /tmp/prof1 -name "*results.db" | head -1)
  python tools/rocpd_stats.py $db > $out/bench_n1_kernel_stats_inflight1.txt 2>&1
  python tools/prof_summary.py $db 60 > $out/bench_n1_kernel_stats_by_grid_inflight1.txt 2>&1
  head -30 $out/bench_n1_kernel_stats_by_grid_inflight1.txt
  python tools/step_timeline.py $db 2 > $out/bench_n1_step_timeline_inflight1.txt 2>&1; tail -3 $out/bench_n1_step_timeline_inflight1.txt
  rm -rf /tmp/prof1
  rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python3 bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5 > $out/bench_n1_under_rocprof.json 2> $out/prof.err
  db=$(find /tmp/prof -name "*results.db" | head -1)
  python tools/rocpd_stats.py $db > $out/bench_n1_kernel_stats.txt 2>&1
  python tools/prof_summary.py $db 60 > $out/bench_n1_kernel_stats_by_grid.txt 2>&1
fi
if [ $what = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/pmc_$c.err
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/pmc_mfma.err
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_n1_pmc_command.json 2> /dev/null      # (the same command un-profiled: its refined-cell count)
  python tools/pmc_dominant.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $out/bench_n1_pmc_command.json $out/pmc_dominant.json > $out/pmc_dominant.log 2>&1; cat $out/pmc_dominant.log
  python tools/pmc_by_grid.py /tmp/pmc_mfma > $out/pmc_mfma_busy.csv 2>&1
  python tools/pmc_by_grid.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $out/pmc_fetch_write_summary.csv 2>&1
  head -20 $out/pmc_mfma_busy.csv
fi
du -sh gpurun_out
