#!/bin/bash
# same-box A/B of the bias-gradient column sums inside the grouped weight-gradient tiles: v_dot2c_f32_bf16 (default) against conversions + adds (variant library)
cd /root/repo; OUT=gpurun_out/r6/bsum; mkdir -p $OUT
OLD=/root/repo/vln-goat_amd/csrc/ab/libgoat_bsum_old.so
for i in 1 2; do
  GOAT_HIP_LIB=$OLD timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > $OUT/old_$i.json 2>/dev/null
  timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > $OUT/new_$i.json 2>/dev/null
done
GOAT_HIP_LIB=$OLD GOAT_BENCH_LARGE_BATCHES=256 timeout 900 python bench.py --leg large_batch > $OUT/old_B256.json 2>/dev/null
GOAT_BENCH_LARGE_BATCHES=256 timeout 900 python bench.py --leg large_batch > $OUT/new_B256.json 2>/dev/null
GOAT_HIP_LIB=$OLD GOAT_BENCH_LARGE_BATCHES=256 timeout 900 python bench.py --leg large_batch > $OUT/old_B256_2.json 2>/dev/null
GOAT_BENCH_LARGE_BATCHES=256 timeout 900 python bench.py --leg large_batch > $OUT/new_B256_2.json 2>/dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('/root/repo/gpurun_out/r6/bsum/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    if 'B256' in d:
        r=d['B256']; print('%-16s B256 %.3f ms  %.0f  family %.4f  step %.4f' % (os.path.basename(f)[:-5], r['ms_per_step'], r['value'], r['roofline']['frac'], r['step_mfma_frac']))
    else:
        print('%-16s %.3f ms  %.0f  family %.4f' % (os.path.basename(f)[:-5], d['ms_per_step'], d['value'], d['roofline']['frac']))
PY
