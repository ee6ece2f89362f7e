"""GEMM-family launches of bench.py's roofline leg alone = (kernel stats of the default bench command) - (kernel stats of the same command
with --no-roofline), both from rocprofv3 --kernel-trace --stats via scripts/prof_stats.py.  Prints per kernel and in total: launches, time,
average duration — the number `roofline.avg_launch_us` of the bench line has to agree with.
    python scripts/roofline_leg_diff.py profiles/roundN_rocprofv3_kernel_stats_bench.txt profiles/roundN_..._no_roofline_leg.txt [bench line json]"""
import json
import re
import sys


def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l.rstrip())
        if m:
            d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return d


full, nrl = load(sys.argv[1]), load(sys.argv[2])
rows, tc, tt = [], 0, 0.0
for k, (c, t) in full.items():
    if any(x in k for x in ('gemm2_', 'gemm_nt_kernel', 'pp_kernel', 'pp_group_kernel')) and not k.startswith('GEMM family'):      # (prof_stats' own family row is a sum of these)
        c0, t0 = nrl.get(k, (0, 0.0))
        if c - c0 > 0:
            rows.append((t - t0, c - c0, k))
            tc += c - c0
            tt += t - t0
print('%-110s %8s %12s %9s' % ('GEMM-family kernel, roofline leg only (full - no_roofline)', 'calls', 'total_us', 'avg_us'))
for t, c, k in sorted(rows, reverse=True):
    print('%-110s %8d %12.1f %9.2f' % (k[:110], c, t, t / c))
print('%-110s %8d %12.1f %9.2f' % ('all GEMM-family launches of the leg', tc, tt, tt / max(tc, 1)))
if len(sys.argv) > 3:
    r = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])['roofline']
    print('bench line of the profiled run: avg_launch_us %.2f over %d launches per cycle, gemm_ms_per_cycle %.3f, frac %.4f'
          % (r['avg_launch_us'], r['launches_per_cycle'], r['gemm_ms_per_cycle'], r['frac']))
