#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r4dp3
mkdir -p $OUT
cd /root/repo
run () { tag=$1; shift; env "$@" timeout 200 python -X faulthandler scripts/in_graph_comm_check.py > $OUT/$tag.txt 2>&1; echo "$tag rc=$?" >> $OUT/summary.txt; grep -h "capturing\|captured\|wire\|IN_GRAPH" $OUT/$tag.txt >> $OUT/summary.txt; }
run f32_all CHK_WIRES=f32
run f32_sap CHK_WIRES=f32 CHK_TASKS=sap
run f32_mlm_waitall CHK_WIRES=f32 CHK_TASKS=mlm CHK_WAIT_ALL=1
run f32_train CHK_WIRES=f32 CHK_TRAIN=1
run bf16_sap CHK_WIRES=bf16 CHK_TASKS=sap
run bf16_sap_waitall CHK_WIRES=bf16 CHK_TASKS=sap CHK_WAIT_ALL=1
cat $OUT/summary.txt
