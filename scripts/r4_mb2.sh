#!/bin/bash
OUT=/root/repo/gpurun_out/r4mb; mkdir -p $OUT; cd /root/repo
F="--no-cpu-baseline --no-extra-configs --no-roofline --warmup 20"
python bench.py $F --steps 300 > $OUT/b48.json 2> $OUT/b48.err
python bench.py $F --steps 300 --batch 24 > $OUT/b24.json 2> $OUT/b24.err
python bench.py $F --steps 900 --batch 24 > $OUT/b24_p1.json 2> $OUT/b24_p1.err &
python bench.py $F --steps 900 --batch 24 > $OUT/b24_p2.json 2> $OUT/b24_p2.err &
wait
for f in b48 b24 b24_p1 b24_p2; do python - $OUT/$f.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'])
PY
done
