#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r4dp
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_dp_two_rank_gpu.py -x -q > $OUT/pytest_dp.txt 2>&1
timeout 300 python scripts/in_graph_comm_check.py > $OUT/in_graph.txt 2>&1
timeout 600 python bench.py --in-graph-comm --no-cpu-baseline --no-extra-configs --no-roofline --steps 30 > $OUT/bench_in_graph.json 2> $OUT/bench_in_graph.err
timeout 600 python bench.py --workload config4 --no-cpu-baseline --steps 20 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
timeout 600 python bench.py --workload config4 --in-graph-comm --wire bf16 --no-roofline --steps 20 > $OUT/bench_c4_ig.json 2> $OUT/bench_c4_ig.err
GOAT_DIST_BACKEND=gloo timeout 900 python bench.py --workload config4 --gpus 2 --no-roofline --steps 10 > $OUT/bench_c4_2rank_gloo.json 2> $OUT/bench_c4_2rank_gloo.err
GOAT_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --wire bf16 --steps 10 > $OUT/bench_c2_2rank_gloo_bf16.json 2> $OUT/bench_c2_2rank_gloo_bf16.err
tail -5 $OUT/*.err
