#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the GEMM family on the final tree (separate --pmc passes, eager launches with the branch streams forked as in the captured
# steps so that the grouped weight-gradient launches are the captured ones) -> gpurun_out/pmcfinal/pmc_gemm_traffic.json
set -u
OUT=/root/repo/gpurun_out/pmcfinal
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GOAT_BENCH_NO_PER_TASK=1 GOAT_BRANCH_STREAMS=always
ARGS="--steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-extra-configs"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python /root/repo/bench.py $ARGS > $OUT/pmc_$c.log 2>&1
done
cd /root/repo
{ python scripts/pmc_summary.py $OUT/pmc_FETCH_SIZE 25; python scripts/pmc_summary.py $OUT/pmc_WRITE_SIZE 25; } > $OUT/pmc_step_summary.txt 2>&1
python scripts/pmc_traffic_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE "$ARGS (GOAT_BRANCH_STREAMS=always)" -1 > $OUT/pmc_gemm_traffic.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
cat $OUT/pmc_gemm_traffic.json | head -16
