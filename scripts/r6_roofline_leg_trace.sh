#!/bin/bash
# the first kernel-trace block of collect_round6.sh again (its summary ran from the wrong directory): kernel stats of the default bench command WITH the roofline leg
set -u
OUT=/root/repo/gpurun_out/r6final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GOAT_BENCH_NO_PER_TASK=1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats.txt; grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json)
rm -rf $OUT/trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats_no_roofline_leg.txt; python scripts/step_breakdown.py $OUT/trace 150 5.2 > $OUT/step_breakdown.txt 2>&1)
rm -rf $OUT/trace
cd /root/repo
python scripts/roofline_leg_diff.py $OUT/kernel_stats.txt $OUT/kernel_stats_no_roofline_leg.txt $OUT/bench_line_under_rocprof.json > $OUT/roofline_leg_kernel_durations.txt 2>&1
cat $OUT/roofline_leg_kernel_durations.txt | head -30
