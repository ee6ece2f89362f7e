#!/bin/bash
# round 6: HIP runtime knobs that touch hipGraph launch / parallel-branch queues, same box, headline step (bench.py --no-extra-configs --no-cpu-baseline --no-roofline)
OUT=/root/repo/gpurun_out/r6/knobs
mkdir -p $OUT
cd /root/repo
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-extra-configs --no-cpu-baseline --no-roofline > $OUT/$name.json 2> $OUT/$name.err; }
run base_1 X=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run hwq2 GPU_MAX_HW_QUEUES=2
run hwq8 GPU_MAX_HW_QUEUES=8
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run gq2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run gq4 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run gq8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run dynq0 DEBUG_HIP_DYNAMIC_QUEUES=0
run base_2 X=1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('/root/repo/gpurun_out/r6/knobs/*.json'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print('%-14s %.3f ms  %.0f  %s' % (os.path.basename(f)[:-5], d['ms_per_step'], d['value'], d.get('ms_per_task_step')))
    except Exception as e:
        print(os.path.basename(f), 'ERR', open(f[:-5]+'.err').read()[-200:].replace('\n',' | '))
PY
