#!/bin/bash
# Round-5 evidence (run on the GPU box through gpurun; everything lands in gpurun_out/r5final/, summaries are copied to profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (headline + roofline leg)                      -> kernel_stats.txt
#   2. the same with --no-roofline (the training steps alone) + per-step family breakdown of the replayed steps      -> kernel_stats_no_roofline_leg.txt, step_breakdown.txt
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE), eager launches: headline, config 5, config 4 (T = 6)           -> pmc_gemm_traffic*.json
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES over the eager task cycle (own pass, kernel trace only)                        -> pmc_step_mfma.txt
#   5. the default bench line (all legs, cpu_baseline)                                                               -> bench_default.json
#   6. micro-benches: vendor GEMM comparison, grouped weight gradients, per-shape GEMM table, attention kernels, launch floor
#   7. data-parallel paths on one GPU: in-graph exchange (one-rank RCCL), config 4 workload line, RCCL capture probe
set -u
OUT=/root/repo/gpurun_out/r5final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GOAT_BENCH_NO_PER_TASK=1      # the traced / counter passes read the TAIL of the run: no per-task timing loop behind the timed steps
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats.txt; grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json)
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats_no_roofline_leg.txt; python scripts/step_breakdown.py $OUT/trace 150 6.3 > $OUT/step_breakdown.txt 2>&1; python scripts/gap_list.py $OUT/trace > $OUT/step_gap_list.txt 2>&1; python scripts/kernel_hist.py $OUT/trace 'ln_bwd|ln_fwd|attn2_|attn_' 150 6.3 > $OUT/step_ln_attention_by_shape.txt 2>&1)
rm -rf $OUT/trace
pmc_pair () {   # $1 = tag, $2 = GEMM launches to average over (-1: the repeating tail = one task cycle), rest = bench arguments
  tag=$1; last=$2; shift; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${tag}_$c -- python /root/repo/bench.py "$@" > $OUT/pmc_${tag}_$c.log 2>&1
  done
  (cd /root/repo && { python scripts/pmc_summary.py $OUT/pmc_${tag}_FETCH_SIZE 25; python scripts/pmc_summary.py $OUT/pmc_${tag}_WRITE_SIZE 25; } > $OUT/pmc_step_summary_$tag.txt
   python scripts/pmc_traffic_json.py $OUT/pmc_${tag}_FETCH_SIZE $OUT/pmc_${tag}_WRITE_SIZE "$*" $last > $OUT/pmc_gemm_traffic_$tag.json)
  rm -rf $OUT/pmc_${tag}_FETCH_SIZE $OUT/pmc_${tag}_WRITE_SIZE
}
GOAT_BENCH_NO_PER_TASK=1 pmc_pair headline -1 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-extra-configs
GOAT_BENCH_NO_NAVIGATOR=1 pmc_pair config5 -1 --leg config5 --steps 10 --no-roofline --no-graph
GOAT_BENCH_NO_NAVIGATOR=1 pmc_pair config4 -1 --leg config4 --steps 6 --no-roofline --no-graph
GOAT_BENCH_NO_PER_TASK=1 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-extra-configs > $OUT/pmc_mfma.log 2>&1
(cd /root/repo && python scripts/pmc_step_mfma.py $OUT/pmc_mfma > $OUT/pmc_step_mfma.txt 2>&1)
rm -rf $OUT/pmc_mfma
cd /root/repo
unset GOAT_BENCH_NO_PER_TASK
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python scripts/vendor_gemm_compare.py > $OUT/vendor_gemm_compare.txt 2>&1
timeout 600 python scripts/wgrad_group_bench.py > $OUT/wgrad_grouped.txt 2>&1
timeout 600 python scripts/gemm_table.py > $OUT/gemm_shape_table.txt 2>&1
timeout 600 python scripts/attn_kernel_bench.py > $OUT/attention_kernels.txt 2>&1
LD_LIBRARY_PATH=vln-goat_amd/csrc timeout 300 scripts/launch_floor.bin > $OUT/gemm_launch_floor.txt 2>&1
timeout 600 python scripts/aten_sites.py > $OUT/aten_sites.txt 2>&1
timeout 600 python bench.py --in-graph-comm --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_in_graph_comm.json 2> $OUT/bench_in_graph_comm.err
timeout 600 python bench.py --workload config4 --no-roofline --steps 20 > $OUT/bench_config4_workload.json 2> $OUT/bench_config4_workload.err
rm -f $OUT/rccl_capture_probe.txt
for n in all_reduce all_to_all_single all_gather_into_tensor reduce_scatter_tensor broadcast; do
  timeout 120 python scripts/rccl_capture_probe.py $n > $OUT/_probe.txt 2>&1; echo "$n: exit code $? $(grep -c CAPTURE_OK $OUT/_probe.txt) capture(s) replayed" >> $OUT/rccl_capture_probe.txt
done
rm -f $OUT/_probe.txt
python scripts/roofline_leg_diff.py $OUT/kernel_stats.txt $OUT/kernel_stats_no_roofline_leg.txt $OUT/bench_line_under_rocprof.json > $OUT/roofline_leg_kernel_durations.txt 2>&1
timeout 600 python scripts/ln_bench.py > $OUT/ln_bench.txt 2>&1
timeout 600 python scripts/gemm_persist_ab.py all > $OUT/gemm_persistent_ab.txt 2>&1
GOAT_FLOOR_RANDOM=1 LD_LIBRARY_PATH=vln-goat_amd/csrc timeout 300 scripts/launch_floor.bin > $OUT/gemm_launch_floor_random_operands.txt 2>&1
GOAT_NAV_T=15 timeout 600 python bench.py --leg config4 --steps 12 --no-roofline > $OUT/bench_config4_T15.json 2> $OUT/bench_config4_T15.err
ls -la $OUT
