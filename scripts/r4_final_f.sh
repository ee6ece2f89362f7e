#!/bin/bash
cd /root/repo
OUT=/root/repo/gpurun_out/r4final; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python -m pytest tests/test_train_step_gpu.py tests/test_static_batch.py -q -m gpu > $OUT/pytest_static.txt 2>&1; tail -n 2 $OUT/pytest_static.txt | cut -c1-200
