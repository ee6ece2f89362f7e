#!/bin/bash
OUT=/root/repo/gpurun_out/r4wgab; mkdir -p $OUT; cd /root/repo
F="--no-cpu-baseline --no-extra-configs --no-roofline --steps 60"
for i in 1 2 3; do
for v in base defer defer24 max24; do
case $v in base) E="";; defer) E="GOAT_WGRAD_DEFER_ALL=1";; defer24) E="GOAT_WGRAD_DEFER_ALL=1 GOAT_WGRAD_GROUP_MAX=24";; max24) E="GOAT_WGRAD_GROUP_MAX=24";; esac
env $E python bench.py $F > $OUT/${v}_$i.json 2>> $OUT/err.txt; python -c "import json,sys; d=json.loads([l for l in open('$OUT/${v}_$i.json') if l.startswith('{')][-1]); print('$v', d['ms_per_step'], d['ms_per_task_step'], d['gemm_shapes_autotuned_in_this_run'])"
done
done
