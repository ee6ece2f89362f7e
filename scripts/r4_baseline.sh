#!/bin/bash
# Round-4 session baseline: default bench line, kernel stats of the steps alone, vendor compare, shape table, attention kernels.
set -u
OUT=/root/repo/gpurun_out/r4base
mkdir -p $OUT
cd /root/repo
python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
(python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats_no_roofline_leg.txt)
rm -rf $OUT/trace
cd /root/repo
python scripts/vendor_gemm_compare.py > $OUT/vendor_gemm_compare.txt 2>&1
python scripts/gemm_table.py > $OUT/gemm_shape_table.txt 2>&1
python scripts/attn_kernel_bench.py > $OUT/attention_kernels.txt 2>&1
ls -la $OUT
