#!/bin/bash
OUT=/root/repo/gpurun_out/r4dbg; mkdir -p $OUT; cd /root/repo
T="tests/test_model_parity_gpu.py::test_losses_and_grads_match_oracle"
timeout 600 python -m pytest "$T" -q -m gpu -k "reverie_small and sap" > $OUT/fan_3.txt 2>&1; grep -h "passed\|failed\|sap_fuse_linear" $OUT/fan_3.txt | cut -c1-300
