#!/bin/bash
# one-off: time the candidate GEMM configurations of the T = 15 sampled-episode graphs (two-pass and one-pass forms) and save the table
cd /root/repo
mkdir -p gpurun_out/r6
GOAT_BENCH_TUNE_TWO_PASS=1 GOAT_SAVE_TUNED=/root/repo/gpurun_out/r6/tuned_t15.json timeout 2400 python bench.py --leg config4 > gpurun_out/r6/config4_leg_tuning_run.json 2> gpurun_out/r6/config4_leg_tuning_run.err
tail -c 400 gpurun_out/r6/config4_leg_tuning_run.err
ls -la gpurun_out/r6/ | grep tuned_t15
python scripts/merge_tuned.py gpurun_out/r6/tuned_t15.json
# and the leg again on the merged table (no tuning inside the run)
timeout 1500 python bench.py --leg config4 > gpurun_out/r6/config4_leg_tuned.json 2> gpurun_out/r6/config4_leg_tuned.err
cp vln-goat_amd/tuned_gfx950.json gpurun_out/r6/tuned_gfx950_after_t15.json
python - <<'PY'
import json
for f in ('config4_leg_tuning_run','config4_leg_tuned'):
    d=json.loads(open('/root/repo/gpurun_out/r6/%s.json'%f).read().strip().splitlines()[-1])
    dg=d['navigator']['dagger_iteration']
    print(f, 'episode', d.get('ms_per_episode'), 'family', d['roofline']['frac'], 'forms', dg.get('forms_ms'))
PY
