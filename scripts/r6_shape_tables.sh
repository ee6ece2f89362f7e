#!/bin/bash
OUT=/root/repo/gpurun_out/r6
mkdir -p $OUT
cd /root/repo
timeout 900 python scripts/gemm_table.py 256 > $OUT/gemm_shape_table_B256.txt 2>&1
timeout 900 python scripts/gemm_table.py 48 > $OUT/gemm_shape_table_B48.txt 2>&1
head -45 $OUT/gemm_shape_table_B256.txt
head -30 $OUT/gemm_shape_table_B48.txt
