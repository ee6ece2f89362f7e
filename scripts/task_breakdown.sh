#!/bin/bash
# per-task kernel breakdown of the replayed step: rocprofv3 kernel trace of bench.py with GOAT_BENCH_TASKS=<task>  (gpurun_out/task_<task>.txt)
set -u
OUT=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  rm -rf $OUT/trace_$t
  GOAT_BENCH_NO_PER_TASK=1 GOAT_BENCH_TASKS=$t rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$t -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/task_$t.log 2>&1
  (cd /root/repo; python scripts/step_breakdown.py $OUT/trace_$t 120 6.5 > $OUT/task_$t.txt 2>&1; grep '^{"metric"' $OUT/task_$t.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $OUT/task_$t.txt)
  rm -rf $OUT/trace_$t
done
