#!/bin/bash
OUT=/root/repo/gpurun_out/r4t3; mkdir -p $OUT; cd /root/repo
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -n 6 $OUT/pytest_gpu.txt
grep "nav bf16 gradients" $OUT/pytest_gpu.txt
