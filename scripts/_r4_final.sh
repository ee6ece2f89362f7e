#!/bin/bash
cd /root/repo
OUT=/root/repo/gpurun_out/r4final; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu_final.txt 2>&1; tail -n 3 $OUT/pytest_gpu_final.txt | cut -c1-200
timeout 1800 bash scripts/collect_round4.sh > $OUT/collect.log 2>&1
timeout 400 python scripts/aten_sites_nav.py > $OUT/aten_sites_nav.txt 2>&1
tail -n 2 $OUT/collect.log
