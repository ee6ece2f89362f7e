"""MFMA-pipe utilisation of the step from two rocprofv3 --pmc passes over the SAME eager task cycle (SQ_VALU_MFMA_BUSY_CYCLES in one pass, SQ_BUSY_CYCLES
/ GRBM_GUI_ACTIVE in the other) joined with the kernel trace of the first: per kernel family, over the last task cycle of the run,
  mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES      (summed over the SIMDs that ran the kernel's waves; 4 SIMDs x 256 CUs = 1024 at most)
  share     = mfma_busy / (1024 x kernel duration x sclk) — the fraction of the chip's MFMA-pipe cycles the kernel used while it ran
and the same for the whole cycle — the counter-side check of bench.py's step_mfma_frac (algorithmic FLOPs / time / peak).
    python scripts/pmc_step_mfma.py <dir of the MFMA pass> [sclk_ghz=2.4]"""
import csv, glob, sys, collections
d = sys.argv[1]
sclk = float(sys.argv[2]) if len(sys.argv) > 2 else 2.4
rows = []
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
trace = {}
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        trace[r['Dispatch_Id']] = (int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'])


def fam(n):
    if any(k in n for k in ('gemm2_', 'gemm_nt_kernel', 'pp_kernel', 'pp_group_kernel')):
        return 'GEMM family'
    if 'attn' in n and 'pool' not in n:
        return 'attention'
    if 'ln_' in n:
        return 'LayerNorm'
    if 'at::native' in n or 'rocclr' in n:
        return 'torch / runtime'
    return 'other HIP kernels'


per = collections.OrderedDict()
for r in rows:
    if r['Counter_Name'] != 'SQ_VALU_MFMA_BUSY_CYCLES':
        continue
    did = r['Dispatch_Id']
    per[did] = per.get(did, 0.0) + float(r['Counter_Value'])
ids = sorted(per, key=lambda x: int(x))
# the repeating tail: last third of the dispatches is at least one full task cycle of a 6-step eager run with 3 warm-up steps
tail = ids[len(ids) * 2 // 3:]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for did in tail:
    if did not in trace:
        continue
    s, e, n = trace[did]
    a = agg[fam(n)]
    a[0] += 1
    a[1] += per[did]
    a[2] += (e - s)
tot_busy = sum(a[1] for a in agg.values())
tot_ns = sum(a[2] for a in agg.values())
print('%-22s %9s %16s %12s %10s' % ('kernel family', 'launches', 'MFMA busy cyc', 'kernel ms', 'MFMA share'))
for k, (n, busy, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-22s %9d %16.0f %12.3f %10.3f' % (k, n, busy, ns / 1e6, busy / (1024.0 * ns * sclk) if ns else 0.0))
print('%-22s %9d %16.0f %12.3f %10.3f   (all kernels of the window, serial sum of kernel durations)' %
      ('whole window', sum(a[0] for a in agg.values()), tot_busy, tot_ns / 1e6, tot_busy / (1024.0 * tot_ns * sclk)))
print('sclk assumed %.2f GHz; 1024 SIMDs; a v_mfma_f32_32x32x16_bf16 keeps its SIMD\'s pipe busy 8 passes x 4 cycles' % sclk)
