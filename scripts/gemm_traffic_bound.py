"""How far can the tile ORDER move the fabric-side traffic of the GEMM family?  (measured 72.4 MB per launch against 42.0 MB
algorithmic = 1.73x, unchanged since round 1.)

MI355X has eight XCDs with private 4 MiB L2s.  A workgroup's operand panels are fetched once per XCD that needs them (the Infinity Cache
serves the repeats, but FETCH_SIZE counts them: it sits on the L2's fabric side).  For C[M,N] = A[M,K]·B[N,K]^T on bm x bn tiles:
  algorithmic reads = (M + N) K 2 bytes                                  (every operand byte once: ONE shared L2)
  order model       = sum over XCDs of (distinct tile rows * bm + distinct tile columns * bn) K 2   under the kernel's own map
                      (xcd_chunk_position + group_m, csrc/gemm2.hip pick_group_m): what perfect sharing INSIDE each L2 gives
  blocking bound    = min over a * b = 8 of (b M + a N) K 2              (the eight L2s as an a x b grid of rectangular blocks of C:
                      the least any assignment of whole tiles to eight equal XCD shares can fetch, up to divisibility)
Reads the shapes and tile choices of one mlm+sap+cfp cycle from a gemm shape table (scripts/gemm_table.py output) and prints the three
numbers per shape and for the cycle, next to the PMC measurement.    python scripts/gemm_traffic_bound.py [table] [pmc json]"""
import ast, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
table = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'round4_gemm_shape_table.txt')
pmc = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'profiles', 'round4_pmc_gemm_traffic.json')


def pick_group_m(tm_, tn_, bm, bn):
    nwg = tm_ * tn_
    q, r = nwg >> 3, nwg & 7
    best, best_cost = 1, 1e300
    for gm in range(1, min(tm_, 64) + 1):
        cost, pos = 0.0, 0
        for x in range(8):
            cnt = q + 1 if x < r else q
            sm, sn = set(), set()
            for _ in range(cnt):
                gsz = gm * tn_
                grp, gi = divmod(pos, gsz)
                h = min(tm_ - grp * gm, gm)
                tn = gi // h
                sm.add(grp * gm + (gi - tn * h)); sn.add(tn)
                pos += 1
            cost += len(sm) * bm + len(sn) * bn
        if cost < best_cost - 1e-9:
            best_cost, best = cost, gm
    return best, best_cost


rows = []
for line in open(table):
    m = re.match(r"\((\d+), (\d+), (\d+), (\d+), (\d+), '([^']*)'\)\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
    if not m:
        continue
    M, N, K, epi, split, cfg, n, tot_us, avg_us = m.groups()
    M, N, K, epi, split, n = int(M), int(N), int(K), int(epi), int(split), int(n)
    t = re.search(r'(\d+)x(\d+)', cfg)
    if not t:
        continue
    bm, bn = int(t.group(1)), int(t.group(2))
    f32 = 't11' in cfg                                   # weight gradients: float32 results
    rows.append((M, N, K, epi, split, cfg, n, bm, bn, f32, float(avg_us)))

tot = dict(alg_r=0.0, model_r=0.0, bound_r=0.0, w=0.0, n=0)
print('%-34s %4s %9s | reads MB: %8s %8s %8s | writes %6s | (reads+writes)/algorithmic: %6s %6s' % (
    'shape (M, N, K) tile', 'n', 'tiles', 'algo', 'order', 'bound', 'MB', 'order', 'bound'))
for M, N, K, epi, split, cfg, n, bm, bn, f32, avg in sorted(rows, key=lambda r: -r[6] * r[10]):
    tm_, tn_ = -(-M // bm), -(-N // bn)
    gm, cost = pick_group_m(tm_, tn_, bm, bn)
    alg_r = (M + N) * K * 2
    model_r = cost * K * 2 * max(1, split) / max(1, split)            # (a split launch fetches every panel once per XCD as well: K is covered once in total)
    bound_r = min((b * M + a * N) * K * 2 for a, b in ((1, 8), (2, 4), (4, 2), (8, 1)))
    bound_r = max(bound_r, alg_r)
    w = M * N * (4 if f32 else 2) * (2 if epi in (1, 2) else 1) + (M * N * 2 if epi in (3, 4) else 0)     # aux store / aux load counted with the writes column
    for k, v in (('alg_r', alg_r), ('model_r', model_r), ('bound_r', bound_r), ('w', w)):
        tot[k] += v * n
    tot['n'] += n
    print('%-34s %4d %4dx%-4d | %18.1f %8.1f %8.1f | %13.1f | %31.2f %6.2f' % (
        '(%d, %d, %d) %s' % (M, N, K, cfg.split()[-2] + ' ' + cfg.split()[-1]), n, tm_, tn_, alg_r / 1e6, model_r / 1e6, bound_r / 1e6, w / 1e6,
        (model_r + w) / (alg_r + w), (bound_r + w) / (alg_r + w)))
N_ = tot['n']
print('\ncycle: %d launches, per launch: algorithmic %.1f MB (reads %.1f + writes %.1f), order model %.1f MB = %.2fx, blocking bound %.1f MB = %.2fx' % (
    N_, (tot['alg_r'] + tot['w']) / N_ / 1e6, tot['alg_r'] / N_ / 1e6, tot['w'] / N_ / 1e6, (tot['model_r'] + tot['w']) / N_ / 1e6,
    (tot['model_r'] + tot['w']) / (tot['alg_r'] + tot['w']), (tot['bound_r'] + tot['w']) / N_ / 1e6, (tot['bound_r'] + tot['w']) / (tot['alg_r'] + tot['w'])))
if os.path.exists(pmc):
    j = json.load(open(pmc))
    print('PMC (%s): reads %.1f MB + writes %.1f MB = %.1f MB per launch over %d launches' % (
        os.path.basename(pmc), j['read_bytes_per_launch'] / 1e6, j['write_bytes_per_launch'] / 1e6, j['traffic_bytes_per_launch'] / 1e6, j['launches_counted']))
