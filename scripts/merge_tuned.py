"""Merge the GEMM tables written by a re-tuning bench run (GOAT_NO_TUNED=1 GOAT_SAVE_TUNED=<path> python bench.py: <path>, <path>.config5,
<path>.config4) into vln-goat_amd/tuned_gfx950.json.  New measurements replace old entries of the same key; keys the run did not meet are kept
unless --fresh is given.    python scripts/merge_tuned.py gpurun_out/tuned_r4.json [--fresh]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, 'vln-goat_amd', 'tuned_gfx950.json')
base = sys.argv[1]
tab = {} if '--fresh' in sys.argv else json.load(open(dst))
n_old = len(tab)
for suf in ('', '.config5', '.config4', '.large_batch'):
    p = base + suf
    if os.path.exists(p):
        new = json.load(open(p))
        changed = sum(1 for k, v in new.items() if tab.get(k) != v)
        tab.update(new)
        print('%s: %d entries, %d new or changed' % (p, len(new), changed))
with open(dst, 'w') as f:
    json.dump(dict(sorted(tab.items())), f, indent=0, sort_keys=True)
print('%s: %d -> %d entries' % (dst, n_old, len(tab)))
