#!/bin/bash
OUT=/root/repo/gpurun_out/r4nav2; mkdir -p $OUT; cd /root/repo
timeout 1200 python -m pytest tests/test_nav_parity_gpu.py tests/test_rollout_gpu.py tests/test_speaker.py -q -m gpu > $OUT/pytest.txt 2>&1; tail -n 3 $OUT/pytest.txt | cut -c1-200
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_model_parity_gpu.py -q -m gpu -k "door or bacl or cross_entropy_rows or linear" > $OUT/pytest_ops.txt 2>&1; tail -n 2 $OUT/pytest_ops.txt | cut -c1-200
timeout 400 python scripts/aten_sites_nav.py > $OUT/aten_nav.txt 2>&1; head -n 2 $OUT/aten_nav.txt | tail -n 1
for i in 1 2; do
GOAT_BENCH_NO_DAGGER=1 timeout 300 python bench.py --leg config4 --steps 20 --no-roofline > $OUT/new_$i.json 2> $OUT/err.txt; python -c "import json; d=json.loads([l for l in open('$OUT/new_$i.json') if l.startswith('{')][-1]); n=d['navigator']; print('config4', d['ms_per_episode'], n['ms_per_episode'], n['vs_frozen_episode'])"
done
