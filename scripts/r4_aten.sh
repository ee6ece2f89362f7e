#!/bin/bash
OUT=/root/repo/gpurun_out/r4aten; mkdir -p $OUT; cd /root/repo
LD_LIBRARY_PATH=vln-goat_amd/csrc timeout 300 scripts/launch_floor.bin > $OUT/launch_floor.txt 2>&1
timeout 300 python scripts/fanin_sites.py > $OUT/fanin.txt 2>&1
timeout 600 python scripts/aten_sites.py > $OUT/aten_sites.txt 2>&1
timeout 900 python -m pytest tests/test_train_step_gpu.py tests/test_static_batch.py tests/test_model_parity_gpu.py -x -q -m gpu > $OUT/pytest_sub.txt 2>&1
tail -n 3 $OUT/pytest_sub.txt
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --no-roofline --steps 30 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-250 $OUT/bench.json
