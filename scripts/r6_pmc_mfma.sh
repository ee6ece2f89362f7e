#!/bin/bash
# round 6: --pmc SQ_VALU_MFMA_BUSY_CYCLES over the eager task cycle (own pass, kernel trace only) -> pmc_step_mfma.txt
set -u
OUT=/root/repo/gpurun_out/r6final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GOAT_BENCH_NO_PER_TASK=1 GOAT_BRANCH_STREAMS=always timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-extra-configs > $OUT/pmc_mfma.log 2>&1
(cd /root/repo && python scripts/pmc_step_mfma.py $OUT/pmc_mfma > $OUT/pmc_step_mfma.txt 2>&1)
rm -rf $OUT/pmc_mfma
cat $OUT/pmc_step_mfma.txt
