#!/bin/bash
# last check of round 6 on the final tree: whole GPU suite, smoke(), the driver's command
OUT=/root/repo/gpurun_out/r6final2
mkdir -p $OUT
cd /root/repo
timeout 2400 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 > $OUT/gpu_suite_tail.txt
tail -3 $OUT/gpu_suite_tail.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc $?" >> $OUT/smoke.txt; tail -3 $OUT/smoke.txt
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err ) 2> $OUT/bench_wall.txt
tail -3 $OUT/bench_wall.txt
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r6final2/bench_driver_cmd.json').read().strip().splitlines()[-1])
r=d['roofline']; dg=d['config4_nav']['navigator']['dagger_iteration']
print(d['ms_per_step'], d['value'], r['frac'], r['launches_per_cycle'], r['traffic'], r['traffic_source'], 'dagger', dg['best_form'], dg['best_ms_per_iteration'], dg['forms_ms'])
PY
