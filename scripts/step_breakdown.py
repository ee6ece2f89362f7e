"""Per-step kernel-time breakdown of the TIMED hipGraph replays of bench.py from a rocprofv3 --kernel-trace CSV: takes the last `window_ms`
of the trace (the replayed steps are the last thing the default --no-extra-configs --no-roofline --no-cpu-baseline run does), groups kernels
into families and prints time per step, share, launches per step, plus the busy / overlap profile of the window.
    python scripts/step_breakdown.py <trace dir> [window_ms=150] [ms_per_step=5.85]"""
import csv, glob, re, sys, collections
d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
msps = float(sys.argv[3]) if len(sys.argv) > 3 else 5.85
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
t_end = max(e for _, e, _ in ev)
ev = [x for x in ev if x[0] >= t_end - win * 1e6]
nsteps = win / msps
# one goat_zero_ranges launch opens every captured step (the gradient arena's fills): count the steps of the window from it when present
marks = sum(1 for _, _, n in ev if 'zero_ranges_kernel' in n)
if marks >= 3:
    nsteps, msps = float(marks), win / marks


def fam(n):
    if 'pp_group_kernel' in n or 'gemm2_group_kernel' in n: return 'GEMM grouped wgrad' + (' (pp)' if 'pp_' in n else ' (g2)')
    if 'pp_kernel' in n: return 'GEMM fwd/dgrad (pp)'
    if 'gemm2_kernel' in n or 'gemm_nt_kernel' in n: return 'GEMM fwd/dgrad/split (g2)'
    if 'ln_bwd' in n or 'ln_reduce' in n: return 'LayerNorm bwd'
    if 'ln_fwd' in n: return 'LayerNorm fwd'
    if 'attn' in n and 'bwd' in n: return 'attention bwd'
    if 'attn' in n and 'pool' not in n: return 'attention fwd'
    if 'at::native' in n or 'rocclr' in n or 'aten' in n: return 'ATen / runtime copies+fills'
    if 'ce_' in n: return 'cross-entropy'
    if 'embed' in n: return 'embedding'
    return 'other HIP kernels'


tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in ev:
    tot[fam(n)] += (e - s) / 1e3; cnt[fam(n)] += 1
allk = sum(tot.values())
print('window %.0f ms (~%.1f steps at %.2f ms), %d kernels, kernel time %.1f ms = %.2f ms per step' % (win, nsteps, msps, len(ev), allk / 1e3, allk / 1e3 / nsteps))
for k, v in tot.most_common():
    print('  %-34s %8.3f ms/step  %5.1f %%   %6.1f launches/step  avg %7.1f us' % (k, v / 1e3 / nsteps, 100 * v / allk, cnt[k] / nsteps, v / cnt[k]))
# busy / overlap profile
pts = []
for s, e, _ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
lvl = 0; last = pts[0][0]; hist = collections.Counter()
for t, dlt in pts:
    hist[min(lvl, 4)] += t - last
    last = t; lvl += dlt
span = pts[-1][0] - pts[0][0]
print('concurrency: ' + '  '.join('%d kernels %.1f %%' % (k, 100 * v / span) for k, v in sorted(hist.items())))
# top individual kernels
per = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in ev:
    a = per[n]; a[0] += 1; a[1] += (e - s) / 1e3
print('top kernels (us per step, calls per step, avg us):')
for n, (c, t) in sorted(per.items(), key=lambda x: -x[1][1])[:28]:
    nm = re.sub(r'\(anonymous namespace\)::|_ZN12_GLOBAL__N_1|void ', '', n)[:96]
    print('  %8.1f  %6.1f  %7.1f  %s' % (t / nsteps, c / nsteps, t / c, nm))
