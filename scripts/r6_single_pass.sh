#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_rollout_gpu.py -v -m gpu -k "single_pass" 2>&1 | grep -n "PASSED\|FAILED\|Fatal\|rollout.py\|test_rollout_gpu.py\|hipops\|streams.py" | tail -400 > gpurun_out/r6/single_pass_test.txt
cat gpurun_out/r6/single_pass_test.txt | grep -v Warning | grep -n "PASSED\|FAILED\|Fatal\|rollout.py\|test_rollout_gpu.py\|hipops\|streams.py" | tail -40
