#!/bin/bash
# Round-3 evidence (run on the GPU box through gpurun; everything lands in gpurun_out/r3final/, summaries are copied to profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench command, headline + its roofline leg        -> kernel_stats.txt
#   2. the same with --no-roofline (the training steps alone: the roofline leg = 1 minus 2)               -> kernel_stats_no_roofline_leg.txt
#   3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), eager launches: headline, config 5, config 4     -> pmc_gemm_traffic*.json
#   4. the default bench line (all legs, cpu_baseline)                                                    -> bench_default.json
#   5. kernel micro-benches: vendor GEMM comparison, grouped weight gradients, per-shape GEMM table
set -u
OUT=/root/repo/gpurun_out/r3final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats.txt; grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json)
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats_no_roofline_leg.txt)
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
pmc_pair headline -1 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-extra-configs
GOAT_BENCH_NO_NAVIGATOR=1 pmc_pair config5 -1 --leg config5 --steps 10 --no-roofline --no-graph
GOAT_BENCH_NO_NAVIGATOR=1 pmc_pair config4 -1 --leg config4 --steps 6 --no-roofline --no-graph
cd /root/repo
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python scripts/vendor_gemm_compare.py > $OUT/vendor_gemm_compare.txt 2>&1
python scripts/wgrad_group_bench.py > $OUT/wgrad_grouped.txt 2>&1
python scripts/gemm_table.py > $OUT/gemm_shape_table.txt 2>&1
python scripts/attn_kernel_bench.py > $OUT/attention_kernels.txt 2>&1
ls -la $OUT
