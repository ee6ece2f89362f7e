#!/bin/bash
cd /root/repo
OUT=/root/repo/gpurun_out/r4final; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu_final.txt 2>&1; tail -n 3 $OUT/pytest_gpu_final.txt | cut -c1-200
timeout 900 python bench.py > $OUT/bench_default_final.json 2> $OUT/bench_default_final.err
