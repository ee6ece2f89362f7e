#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -k "wgrad or gemm_bf16 or linear" 2>&1 | tail -3
{
echo "--- with bias gradient (v_dot2c behind the MFMAs, row blocks split over the four waves of a group)"; WG_N=7,12,14 timeout 600 python scripts/r6_wgrad_round_quantisation.py 2>&1 | grep rows
echo "--- without bias gradient"; WG_NO_BIAS=1 WG_N=7,12,14 timeout 600 python scripts/r6_wgrad_round_quantisation.py 2>&1 | grep rows
} > gpurun_out/r6/wgrad_bias_ab_dot2_split.txt
cat gpurun_out/r6/wgrad_bias_ab_dot2_split.txt
