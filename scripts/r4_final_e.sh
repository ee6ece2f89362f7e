#!/bin/bash
cd /root/repo
OUT=/root/repo/gpurun_out/r4final; mkdir -p $OUT
GOAT_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; cut -c1-200 $OUT/bench_2rank_gloo.json
GOAT_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --workload config4 --no-roofline --steps 6 --warmup 2 > $OUT/bench_2rank_gloo_c4.json 2> $OUT/bench_2rank_gloo_c4.err; cut -c1-200 $OUT/bench_2rank_gloo_c4.json
bash scripts/collect_round4.sh > $OUT/collect.log 2>&1
tail -n 3 $OUT/collect.log
