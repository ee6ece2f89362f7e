#!/bin/bash
OUT=/root/repo/gpurun_out/r4dp5; mkdir -p $OUT; cd /root/repo
timeout 600 python -m pytest tests/test_dp_two_rank_gpu.py -x -q > $OUT/pytest_dp.txt 2>&1
timeout 600 python bench.py --workload config4 --in-graph-comm --wire bf16 --no-roofline --steps 20 > $OUT/bench_c4_ig.json 2> $OUT/bench_c4_ig.err
timeout 600 python bench.py --in-graph-comm --wire bf16 --no-cpu-baseline --no-extra-configs --no-roofline --steps 30 > $OUT/bench_in_graph_bf16.json 2> $OUT/bench_in_graph_bf16.err
tail -3 $OUT/pytest_dp.txt; tail -3 $OUT/*.err
