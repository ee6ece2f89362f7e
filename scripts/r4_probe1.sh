#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r4probe1
mkdir -p $OUT
cd /root/repo
LD_LIBRARY_PATH=vln-goat_amd/csrc timeout 300 scripts/launch_floor.bin > $OUT/launch_floor.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
cd /root/repo
python scripts/prof_stats.py $OUT/trace 90 > $OUT/kernel_stats_no_roofline_leg.txt
python scripts/step_breakdown.py $OUT/trace > $OUT/step_breakdown.txt 2>&1
rm -rf $OUT/trace
