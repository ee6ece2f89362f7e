#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
OLD=/root/repo/vln-goat_amd/csrc/ab/libgoat_bsum_behind.so
{
for i in 1 2; do
echo "--- 4-byte bias loads (variant library: the tree before this change)"; GOAT_HIP_LIB=$OLD timeout 600 python scripts/r6_bias_epilogue_probe.py 2>&1 | grep " x "
echo "--- 16-byte bias loads (default)"; timeout 600 python scripts/r6_bias_epilogue_probe.py 2>&1 | grep " x "
done
} > gpurun_out/r6/bias_epilogue_probe.txt
cat gpurun_out/r6/bias_epilogue_probe.txt
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -k "gemm or linear or ffn" 2>&1 | tail -2
OUT=gpurun_out/r6/biasvec; mkdir -p $OUT
for i in 1 2; do
  GOAT_HIP_LIB=$OLD timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > $OUT/old_$i.json 2>/dev/null
  timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > $OUT/new_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('/root/repo/gpurun_out/r6/biasvec/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-8s %.3f ms  %.0f  family %.4f' % (os.path.basename(f)[:-5], d['ms_per_step'], d['value'], d['roofline']['frac']))
PY
