#!/bin/bash
# round 6: linear_bank (all layers' cross-attention K|V of one attended sequence as one GEMM) — tests + same-box A/B of the headline step
set -x
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "linear_bank or attention_fwd_bwd" 2>&1 | tail -15 > gpurun_out/r6/kvbank_tests.txt
timeout 1500 python -m pytest tests/test_model_parity_gpu.py tests/test_nav_parity_gpu.py -x -q -m gpu 2>&1 | tail -15 >> gpurun_out/r6/kvbank_tests.txt
for i in 1 2; do
  GOAT_NO_KV_BANK=1 timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > gpurun_out/r6/bench_nobank_$i.json 2> gpurun_out/r6/bench_nobank_$i.err
  timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > gpurun_out/r6/bench_bank_$i.json 2> gpurun_out/r6/bench_bank_$i.err
done
cat gpurun_out/r6/kvbank_tests.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('launches_per_cycle'))
    except Exception as e:
        print(f, 'ERR', e)
PY
