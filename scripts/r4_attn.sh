#!/bin/bash
OUT=/root/repo/gpurun_out/r4attn; mkdir -p $OUT; cd /root/repo
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "attn or attention" > $OUT/pytest_attn.txt 2>&1; tail -n 3 $OUT/pytest_attn.txt
timeout 300 python scripts/attn_kernel_bench.py > $OUT/attn_seq.txt 2>&1
GOAT_ATTN_NO_SEQ=1 timeout 300 python scripts/attn_kernel_bench.py > $OUT/attn_noseq.txt 2>&1
grep "p=0.1" $OUT/attn_seq.txt | cut -c1-200; echo ---; grep "p=0.1" $OUT/attn_noseq.txt | cut -c1-200
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --no-roofline --steps 60 > $OUT/bench_$i.json 2> $OUT/bench.err; cut -c100-260 $OUT/bench_$i.json
GOAT_ATTN_NO_SEQ=1 timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --no-roofline --steps 60 > $OUT/bench_noseq_$i.json 2>> $OUT/bench.err; cut -c100-260 $OUT/bench_noseq_$i.json
done
timeout 1500 python -m pytest tests/test_model_parity_gpu.py tests/test_nav_parity_gpu.py -q -m gpu > $OUT/pytest_model.txt 2>&1; tail -n 4 $OUT/pytest_model.txt
