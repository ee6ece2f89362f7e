#!/bin/bash
# the timing-sensitive parts of scripts/collect_round4.sh again (the first pass ran on a stale libgoat_hip.so: an attention-backward experiment)
set -u
OUT=/root/repo/gpurun_out/r4final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats.txt; grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json)
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats_no_roofline_leg.txt; python scripts/step_breakdown.py $OUT/trace 150 6.3 > $OUT/step_breakdown.txt 2>&1; python scripts/gap_list.py $OUT/trace > $OUT/step_gap_list.txt 2>&1; python scripts/kernel_hist.py $OUT/trace 'ln_bwd|ln_fwd|attn2_|attn_' 150 6.3 > $OUT/step_ln_attention_by_shape.txt 2>&1)
rm -rf $OUT/trace $OUT/trace_full
GOAT_BENCH_NO_PER_TASK=1 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-extra-configs > $OUT/pmc_mfma.log 2>&1
(cd /root/repo && python scripts/pmc_step_mfma.py $OUT/pmc_mfma > $OUT/pmc_step_mfma.txt 2>&1)
rm -rf $OUT/pmc_mfma
cd /root/repo
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python scripts/attn_kernel_bench.py > $OUT/attention_kernels.txt 2>&1
python bench.py --in-graph-comm --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_in_graph_comm.json 2> $OUT/bench_in_graph_comm.err
python bench.py --workload config4 --no-roofline --steps 20 > $OUT/bench_config4_workload.json 2> $OUT/bench_config4_workload.err
python scripts/ln_bench.py > $OUT/ln_bench.txt 2>&1
python scripts/roofline_leg_diff.py $OUT/kernel_stats.txt $OUT/kernel_stats_no_roofline_leg.txt $OUT/bench_line_under_rocprof.json > $OUT/roofline_leg_kernel_durations.txt 2>&1
