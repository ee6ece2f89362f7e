#!/bin/bash
# Collection of round 6 (final tree): the whole GPU suite, smoke(), rocprofv3 kernel stats of the default bench command with / without the roofline
# leg, the step breakdown, the two --pmc traffic passes of the GEMM family, the driver's own command (--steps 20 --warmup 5) and the default bench
# line.  Run on the GPU box through gpurun; summaries are copied to profiles/round6_final_* by hand afterwards.
set -u
OUT=/root/repo/gpurun_out/r6final
mkdir -p $OUT
cd /root/repo
timeout 2400 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -40 > $OUT/gpu_suite_tail.txt
tail -4 $OUT/gpu_suite_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc $?" >> $OUT/smoke.txt
cd /tmp && export TMPDIR=/tmp
export GOAT_BENCH_NO_PER_TASK=1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats.txt; grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json)
rm -rf $OUT/trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-configs --no-roofline > $OUT/bench_under_rocprof_nrl.log 2>&1
(cd /root/repo && python scripts/prof_stats.py $OUT/trace 70 > $OUT/kernel_stats_no_roofline_leg.txt; python scripts/step_breakdown.py $OUT/trace 150 5.2 > $OUT/step_breakdown.txt 2>&1; python scripts/kernel_hist.py $OUT/trace 'ln_bwd|ln_fwd|attn2_|attn_' 150 5.2 > $OUT/step_ln_attention_by_shape.txt 2>&1)
rm -rf $OUT/trace
cd /root/repo
python scripts/roofline_leg_diff.py $OUT/kernel_stats.txt $OUT/kernel_stats_no_roofline_leg.txt $OUT/bench_line_under_rocprof.json > $OUT/roofline_leg_kernel_durations.txt 2>&1
# PMC traffic of the GEMM family (separate passes; eager launches with the branch streams forked as in the captured steps)
cd /tmp
export GOAT_BRANCH_STREAMS=always
ARGS="--steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-extra-configs"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python /root/repo/bench.py $ARGS > $OUT/pmc_$c.log 2>&1
done
unset GOAT_BRANCH_STREAMS
cd /root/repo
{ python scripts/pmc_summary.py $OUT/pmc_FETCH_SIZE 25; python scripts/pmc_summary.py $OUT/pmc_WRITE_SIZE 25; } > $OUT/pmc_step_summary.txt 2>&1
python scripts/pmc_traffic_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE "$ARGS (GOAT_BRANCH_STREAMS=always)" -1 > $OUT/pmc_gemm_traffic.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
unset GOAT_BENCH_NO_PER_TASK
# the driver's command, then the default line (all legs)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rocm-smi --showclocks --showpower --json > $OUT/smi_after.json 2>/dev/null
ls -la $OUT
python - <<'PY'
import json
for f in ('bench_driver_cmd','bench_default'):
    try:
        d=json.loads(open('/root/repo/gpurun_out/r6final/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['launches_per_cycle'], d.get('step_mfma_frac'))
    except Exception as e:
        print(f, 'ERR', e)
PY
