#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6
timeout 1500 python bench.py --leg config4 > gpurun_out/r6/config4_leg.json 2> gpurun_out/r6/config4_leg.err
tail -c 1500 gpurun_out/r6/config4_leg.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r6/config4_leg.json').read().strip().splitlines()[-1])
nav=d.get('navigator',{}); dg=nav.get('dagger_iteration',{})
print('episode', d.get('ms_per_episode'), 'forms', dg.get('forms_ms'), 'best', dg.get('best_form'))
print(json.dumps((dg.get('two_pass') or {}).get('single_pass_captured'))[:900])
print(json.dumps({k:v for k,v in ((dg.get('two_pass') or {}).get('pass1_captured') or {}).items() if k!='teacher_overlapped'})[:600])
PY
