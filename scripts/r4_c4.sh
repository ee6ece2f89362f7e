#!/bin/bash
OUT=/root/repo/gpurun_out/r4c4; mkdir -p $OUT; cd /root/repo
timeout 1200 python bench.py --leg config4 --steps 20 > $OUT/c4.json 2> $OUT/c4.err
tail -n 5 $OUT/c4.err | cut -c1-300
