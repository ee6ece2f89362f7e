#!/bin/bash
# Last collection of round 5 (final tree): rocprofv3 kernel stats of the default bench command with / without the roofline leg, the step
# breakdown, the default bench line, smoke().  Run on the GPU box through gpurun; summaries are copied to profiles/round5_last_*.
set -u
OUT=/root/repo/gpurun_out/r5final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GOAT_BENCH_NO_PER_TASK=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats.txt; grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json)
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats_no_roofline_leg.txt; python scripts/step_breakdown.py $OUT/trace 150 6.3 > $OUT/step_breakdown.txt 2>&1)
rm -rf $OUT/trace
cd /root/repo
unset GOAT_BENCH_NO_PER_TASK
python scripts/roofline_leg_diff.py $OUT/kernel_stats.txt $OUT/kernel_stats_no_roofline_leg.txt $OUT/bench_line_under_rocprof.json > $OUT/roofline_leg_kernel_durations.txt 2>&1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc $?" >> $OUT/smoke.txt
ls -la $OUT
