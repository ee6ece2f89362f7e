#!/bin/bash
OUT=/root/repo/gpurun_out/r4dp4; mkdir -p $OUT; cd /root/repo
for n in all_reduce all_to_all_single all_gather_into_tensor reduce_scatter_tensor broadcast all_gather_list; do
  timeout 120 python scripts/rccl_capture_probe.py $n > $OUT/$n.txt 2>&1; echo "$n rc=$? $(grep -c CAPTURE_OK $OUT/$n.txt)" >> $OUT/summary.txt
done
cat $OUT/summary.txt
