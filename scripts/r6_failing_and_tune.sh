#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r6
mkdir -p $OUT
cd /root/repo
timeout 1200 python -m pytest tests/test_rollout_gpu.py "tests/test_nav_parity_gpu.py::test_nav_bf16_gradients_against_the_pinned_float32_gradients" tests/test_train_step_gpu.py::test_captured_steps_launch_only_measured_configurations -q -m gpu -s 2>&1 | tail -150 > $OUT/failing.txt
grep -n "^E \|FAILED\|passed\|failed\|nav bf16\|door gate\|bf16 projections" $OUT/failing.txt | head -60
GOAT_SAVE_TUNED=$OUT/tuned_r6.json timeout 2400 python bench.py > $OUT/bench_full_1.json 2> $OUT/bench_full_1.err
tail -c 600 $OUT/bench_full_1.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r6/bench_full_1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('launches_per_cycle'), d.get('gemm_shapes_autotuned_in_this_run'))
for k in ('fresh_batch','train_loop','with_optimizer','config5_reverie','large_batch'):
    print(k, json.dumps(d.get(k))[:600])
c4=d.get('config4_nav') or {}
print('config4', {k:(v if not isinstance(v,dict) else '...') for k,v in c4.items()})
PY
ls $OUT | grep tuned
