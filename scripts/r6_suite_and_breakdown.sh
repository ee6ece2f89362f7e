#!/bin/bash
# round 6: the whole GPU suite on the split tree + a kernel-trace step breakdown of the headline step
set -u
OUT=/root/repo/gpurun_out/r6
mkdir -p $OUT
cd /root/repo
timeout 2400 python -m pytest tests -q -m gpu --maxfail=12 --durations=15 2>&1 | tail -60 > $OUT/gpu_suite.txt
tail -25 $OUT/gpu_suite.txt
cd /tmp && export TMPDIR=/tmp
GOAT_BENCH_NO_PER_TASK=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats_no_roofline_leg.txt; python scripts/step_breakdown.py $OUT/trace 150 5.2 > $OUT/step_breakdown.txt 2>&1; python scripts/kernel_hist.py $OUT/trace 'ln_bwd|ln_fwd|attn2_|attn_' 150 5.2 > $OUT/step_ln_attention_by_shape.txt 2>&1)
rm -rf $OUT/trace
head -45 $OUT/step_breakdown.txt
cd /root/repo
# sensitivity of the headline to the length of the timed region (the driver runs --steps 20 --warmup 5)
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-roofline > $OUT/bench_s20_$i.json 2>/dev/null
  timeout 600 python bench.py --no-extra-configs --no-cpu-baseline --no-roofline > $OUT/bench_s96_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6/bench_s*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['value'], d.get('ms_per_task_step'))
    except Exception as e:
        print(f, 'ERR', e)
PY
