#!/bin/bash
OUT=/root/repo/gpurun_out/r4t4; mkdir -p $OUT; cd /root/repo
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 2 $OUT/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -n 4 $OUT/pytest_gpu.txt | cut -c1-200
grep "nav bf16 gradients" $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
