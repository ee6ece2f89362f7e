#!/bin/bash
OUT=/root/repo/gpurun_out/r4dp6; mkdir -p $OUT; cd /root/repo
timeout 600 python -m pytest tests/test_dp_two_rank_gpu.py -x -q -k captured > $OUT/pytest_dp.txt 2>&1
tail -n 3 $OUT/pytest_dp.txt
grep -n "wire\|IN_GRAPH\|what()" $OUT/pytest_dp.txt | head
