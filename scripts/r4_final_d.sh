#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r4final
mkdir -p $OUT
export GOAT_BENCH_NO_PER_TASK=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats.txt; grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json)
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
MS=$(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_under_rocprof_nrl.log | head -1 | grep -o '[0-9.]*$')
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats_no_roofline_leg.txt; python scripts/step_breakdown.py $OUT/trace 150 $MS > $OUT/step_breakdown.txt 2>&1; python scripts/gap_list.py $OUT/trace > $OUT/step_gap_list.txt 2>&1; python scripts/kernel_hist.py $OUT/trace 'ln_bwd|ln_fwd|attn2_|attn_' 150 $MS > $OUT/step_ln_attention_by_shape.txt 2>&1)
rm -rf $OUT/trace
cd /root/repo
python scripts/roofline_leg_diff.py $OUT/kernel_stats.txt $OUT/kernel_stats_no_roofline_leg.txt $OUT/bench_line_under_rocprof.json > $OUT/roofline_leg_kernel_durations.txt 2>&1
unset GOAT_BENCH_NO_PER_TASK
