#!/bin/bash
# one sample of the driver's command (headline only) on whatever box this call lands on, with the box's clocks / power beside it
cd /root/repo; mkdir -p gpurun_out/r6/boxes
tag=$(date +%H%M%S)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline > gpurun_out/r6/boxes/line_$tag.json 2> /dev/null
rocm-smi --showclocks --showpower --showtemp --json > gpurun_out/r6/boxes/smi_$tag.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r6/boxes/line_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
PY
