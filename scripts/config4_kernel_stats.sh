set -u
OUT=/root/repo/gpurun_out/c4prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GOAT_BENCH_NO_NAVIGATOR=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --leg config4 --steps 20 --no-roofline > $OUT/bench.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 45 > $OUT/kernel_stats.txt)
rm -rf $OUT/trace
head -50 $OUT/kernel_stats.txt | cut -c1-170
grep -o '"ms_per_episode": [0-9.]*' $OUT/bench.log | head -2
