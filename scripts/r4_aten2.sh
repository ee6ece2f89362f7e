#!/bin/bash
OUT=/root/repo/gpurun_out/r4aten3; mkdir -p $OUT; cd /root/repo
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "add_n or zero_ranges or already_padded or layer_norm or linear" > $OUT/pytest_ops.txt 2>&1; tail -n 3 $OUT/pytest_ops.txt
timeout 1500 python -m pytest tests/test_model_parity_gpu.py tests/test_nav_parity_gpu.py tests/test_train_step_gpu.py -x -q -m gpu > $OUT/pytest_model.txt 2>&1; tail -n 3 $OUT/pytest_model.txt
timeout 600 python scripts/aten_sites.py > $OUT/aten_sites.txt 2>&1
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --no-roofline --steps 60 > $OUT/bench_$i.json 2> $OUT/bench.err; cut -c100-260 $OUT/bench_$i.json
GOAT_NO_FANOUT=1 timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --no-roofline --steps 60 > $OUT/bench_nofan_$i.json 2>> $OUT/bench.err; cut -c100-260 $OUT/bench_nofan_$i.json
done
