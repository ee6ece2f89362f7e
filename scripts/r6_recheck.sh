#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_train_step_gpu.py tests/test_rollout_gpu.py -q -m gpu 2>&1 | tail -6 > gpurun_out/r6/recheck.txt
cat gpurun_out/r6/recheck.txt
