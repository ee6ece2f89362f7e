#!/bin/bash
OUT=/root/repo/gpurun_out/r4lnab; mkdir -p $OUT; cd /root/repo
F="--no-cpu-baseline --no-extra-configs --no-roofline --steps 60"
for i in 1 2 3; do
for v in lnw4 lnw2 lnw4p; do
GOAT_HIP_LIB=/root/repo/vln-goat_amd/csrc/ab/libgoat_$v.so python bench.py $F > $OUT/${v}_$i.json 2>> $OUT/err.txt; python -c "import json,sys; d=json.loads([l for l in open('$OUT/${v}_$i.json') if l.startswith('{')][-1]); print('$v', d['ms_per_step'], d['ms_per_task_step'])"
done
done
