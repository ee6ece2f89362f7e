#!/bin/bash
set -x
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "layer_norm or linear_bank" 2>&1 | tail -8 > gpurun_out/r6/ln768_tests.txt
cat gpurun_out/r6/ln768_tests.txt
timeout 300 python scripts/ln_bench.py > gpurun_out/r6/ln_bench_768.txt 2>&1
GOAT_LN_BWD_GENERIC=1 timeout 300 python scripts/ln_bench.py > gpurun_out/r6/ln_bench_generic.txt 2>&1
tail -30 gpurun_out/r6/ln_bench_768.txt; tail -30 gpurun_out/r6/ln_bench_generic.txt
timeout 600 python scripts/r6_layout_rate.py > gpurun_out/r6/layout_rate.txt 2>&1
cat gpurun_out/r6/layout_rate.txt
for i in 1 2; do
  GOAT_LN_BWD_GENERIC=1 timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > gpurun_out/r6/bench_lngeneric_$i.json 2> gpurun_out/r6/bench_lngeneric_$i.err
  timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > gpurun_out/r6/bench_ln768_$i.json 2> gpurun_out/r6/bench_ln768_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6/bench_ln*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('launches_per_cycle'))
    except Exception as e:
        print(f, 'ERR', e)
PY
