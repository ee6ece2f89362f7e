#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -k "wgrad or linear" 2>&1 | tail -2
B=/root/repo/vln-goat_amd/csrc/ab/libgoat_bsum_behind.so
{
for i in 1 2; do
echo "--- interleaved (default)"; WG_N=12,14 timeout 600 python scripts/r6_wgrad_round_quantisation.py 2>&1 | grep rows
echo "--- behind the MFMAs (variant)"; GOAT_HIP_LIB=$B WG_N=12,14 timeout 600 python scripts/r6_wgrad_round_quantisation.py 2>&1 | grep rows
done
echo "--- without bias gradient"; WG_NO_BIAS=1 WG_N=12,14 timeout 600 python scripts/r6_wgrad_round_quantisation.py 2>&1 | grep rows
} > gpurun_out/r6/wgrad_bias_interleave.txt
cat gpurun_out/r6/wgrad_bias_interleave.txt
