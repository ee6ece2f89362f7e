#!/bin/bash
OUT=/root/repo/gpurun_out/r4rd; mkdir -p $OUT; cd /root/repo
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "rowdot or linear" > $OUT/pytest_ops.txt 2>&1; tail -n 2 $OUT/pytest_ops.txt
timeout 1800 python -m pytest tests/test_model_parity_gpu.py tests/test_nav_parity_gpu.py tests/test_train_step_gpu.py -q -m gpu > $OUT/pytest_model.txt 2>&1; tail -n 3 $OUT/pytest_model.txt | cut -c1-200
timeout 600 python scripts/aten_sites.py > $OUT/aten_sites.txt 2>&1; grep "==" $OUT/aten_sites.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra-configs --no-roofline --steps 60 > $OUT/b_$i.json 2> $OUT/err.txt; python -c "import json; d=json.loads([l for l in open('$OUT/b_$i.json') if l.startswith('{')][-1]); print('new', d['ms_per_step'], d['ms_per_task_step'])"
GOAT_NO_ROWDOT=1 python bench.py --no-cpu-baseline --no-extra-configs --no-roofline --steps 60 > $OUT/n_$i.json 2>> $OUT/err.txt; python -c "import json; d=json.loads([l for l in open('$OUT/n_$i.json') if l.startswith('{')][-1]); print('norowdot', d['ms_per_step'], d['ms_per_task_step'])"
done
