#!/bin/bash
OUT=/root/repo/gpurun_out/r4navbr; mkdir -p $OUT; cd /root/repo
for i in 1 2; do
GOAT_BENCH_NO_DAGGER=1 timeout 300 python bench.py --leg config4 --steps 20 --no-roofline > $OUT/new_$i.json 2> $OUT/err.txt; python -c "import json; d=json.loads([l for l in open('$OUT/new_$i.json') if l.startswith('{')][-1]); n=d['navigator']; print('branches', d['ms_per_episode'], {k:v for k,v in n.items() if k not in ('what','dagger_iteration')})"
done
