#!/bin/bash
# traced / counter passes of scripts/collect_round4.sh whose window is the TAIL of the run: without bench.py's per-task timing loop behind the timed steps
set -u
OUT=/root/repo/gpurun_out/r4final
mkdir -p $OUT
export GOAT_BENCH_NO_PER_TASK=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1
(python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats.txt; grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json)
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats_no_roofline_leg.txt; python scripts/step_breakdown.py $OUT/trace 150 6.3 > $OUT/step_breakdown.txt 2>&1; python scripts/gap_list.py $OUT/trace > $OUT/step_gap_list.txt 2>&1; python scripts/kernel_hist.py $OUT/trace 'ln_bwd|ln_fwd|attn2_|attn_' 150 6.3 > $OUT/step_ln_attention_by_shape.txt 2>&1)
rm -rf $OUT/trace
cd /root/repo
python scripts/roofline_leg_diff.py $OUT/kernel_stats.txt $OUT/kernel_stats_no_roofline_leg.txt $OUT/bench_line_under_rocprof.json > $OUT/roofline_leg_kernel_durations.txt 2>&1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_headline_$c -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-extra-configs > $OUT/pmc_headline_$c.log 2>&1
done
(cd /root/repo && { python scripts/pmc_summary.py $OUT/pmc_headline_FETCH_SIZE 25; python scripts/pmc_summary.py $OUT/pmc_headline_WRITE_SIZE 25; } > $OUT/pmc_step_summary_headline.txt
 python scripts/pmc_traffic_json.py $OUT/pmc_headline_FETCH_SIZE $OUT/pmc_headline_WRITE_SIZE "--steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-extra-configs" -1 > $OUT/pmc_gemm_traffic_headline.json)
rm -rf $OUT/pmc_headline_FETCH_SIZE $OUT/pmc_headline_WRITE_SIZE
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-extra-configs > $OUT/pmc_mfma.log 2>&1
(cd /root/repo && python scripts/pmc_step_mfma.py $OUT/pmc_mfma > $OUT/pmc_step_mfma.txt 2>&1)
rm -rf $OUT/pmc_mfma
