#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r4dp2
mkdir -p $OUT
cd /root/repo
timeout 300 python -X faulthandler scripts/in_graph_comm_check.py > $OUT/in_graph.txt 2>&1
timeout 600 python -X faulthandler bench.py --workload config4 --in-graph-comm --wire bf16 --no-roofline --steps 20 > $OUT/bench_c4_ig.json 2> $OUT/bench_c4_ig.err
timeout 600 python -X faulthandler bench.py --workload config4 --in-graph-comm --no-roofline --steps 20 > $OUT/bench_c4_ig_f32.json 2> $OUT/bench_c4_ig_f32.err
timeout 600 python -X faulthandler bench.py --in-graph-comm --wire bf16 --no-cpu-baseline --no-extra-configs --no-roofline --steps 30 > $OUT/bench_in_graph_bf16.json 2> $OUT/bench_in_graph_bf16.err
timeout 600 python -m pytest tests/test_dp_two_rank_gpu.py -x -q -k finetune > $OUT/pytest_dp.txt 2>&1
