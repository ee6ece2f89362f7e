#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the GEMM family for the config 5 and config 4 legs on the final tree (separate --pmc passes, eager launches with the
# branch streams forked as in the captured bodies) -> gpurun_out/pmcfinal/pmc_gemm_traffic_<leg>.json
set -u
OUT=/root/repo/gpurun_out/pmcfinal
mkdir -p $OUT
export GOAT_BENCH_NO_NAVIGATOR=1 GOAT_BRANCH_STREAMS=always
for leg in config5 config4; do
  if [ $leg = config5 ]; then ARGS="--leg config5 --steps 10 --no-roofline --no-graph"; else ARGS="--leg config4 --steps 6 --no-roofline --no-graph"; fi
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${leg}_$c -- python /root/repo/bench.py $ARGS > $OUT/pmc_${leg}_$c.log 2>&1
  done
  cd /root/repo
  python scripts/pmc_traffic_json.py $OUT/pmc_${leg}_FETCH_SIZE $OUT/pmc_${leg}_WRITE_SIZE "$ARGS (GOAT_BRANCH_STREAMS=always)" -1 > $OUT/pmc_gemm_traffic_$leg.json
  rm -rf $OUT/pmc_${leg}_FETCH_SIZE $OUT/pmc_${leg}_WRITE_SIZE
  python -c "import json; d=json.load(open('$OUT/pmc_gemm_traffic_$leg.json')); print('$leg', d['launches_counted'], round(d['traffic_bytes_per_launch']))"
done
