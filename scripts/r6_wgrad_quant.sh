#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 900 python scripts/r6_wgrad_round_quantisation.py > gpurun_out/r6/wgrad_round_quantisation.txt 2>&1
cat gpurun_out/r6/wgrad_round_quantisation.txt
