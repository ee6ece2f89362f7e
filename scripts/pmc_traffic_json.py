"""fabric-side bytes per GEMM launch from two rocprofv3 --pmc output directories (FETCH_SIZE pass, WRITE_SIZE pass) -> JSON on stdout.
    python scripts/pmc_traffic_json.py <fetch dir> <write dir> "<bench arguments of the passes>" """
import csv
import glob
import json
import sys


def gemm_rows(d, counter):
    rows = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r['Kernel_Name'] for k in ('gemm2_', 'gemm_nt_kernel', 'pp_kernel', 'pp_group_kernel')) and r['Counter_Name'] == counter:
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    rows.sort()
    return rows


def period(rows):
    """length of the repeating tail of the GEMM dispatch sequence (one task cycle / one episode): smallest P >= 32 with the last
    P kernel names equal to the P before them.  0 if the tail does not repeat."""
    names = [r[1] for r in rows]
    for P in range(32, len(names) // 2 + 1):
        if names[-P:] == names[-2 * P:-P]:
            return P
    return 0


def fam_avg(rows, last):
    rows = rows[-last:] if last else rows
    n = len(rows)
    return n, (sum(r[2] for r in rows) / n if n else None)


LAST = int(sys.argv[4]) if len(sys.argv) > 4 else 0       # N: the last N GEMM dispatches; -1: the repeating tail (auto-detected)
RF, RW = gemm_rows(sys.argv[1], 'FETCH_SIZE'), gemm_rows(sys.argv[2], 'WRITE_SIZE')
if LAST < 0:
    LAST = period(RF)
nf, f = fam_avg(RF, LAST)
nw, w = fam_avg(RW, LAST)
nf_all, f_all = fam_avg(RF, 0)
nw_all, w_all = fam_avg(RW, 0)
print(json.dumps({'kernel': 'gemm2_kernel + gemm2_group_kernel + pp_kernel + pp_group_kernel + gemm_nt_kernel', 'launches_counted': nf, 'selection': ('the last %d GEMM dispatches = the repeating tail of the dispatch sequence (one task cycle / episode; warm-up excluded)' % LAST) if LAST else 'every GEMM dispatch of the run',
                  'all_dispatches': {'launches': nf_all, 'traffic_bytes_per_launch': f_all * 1024 * 2 + w_all * 1024},
                  'fetch_kb_per_launch_reported': f, 'write_kb_per_launch_reported': w,
                  'read_bytes_per_launch': f * 1024 * 2, 'write_bytes_per_launch': w * 1024,
                  'traffic_bytes_per_launch': f * 1024 * 2 + w * 1024,
                  'corrections': 'FETCH_SIZE (KB) doubled: gfx950 rocprofv3 tallies 128-B requests at 64 B for 16-B/lane reads (MI355X_MICROARCH.md, HBM); '
                                 'WRITE_SIZE (KB) x1, calibrated on act_bwd (23,307 KB reported for a 23,040 KB store). FETCH_SIZE counts L2-miss requests on the '
                                 'fabric side, Infinity-Cache hits included: an upper bound on HBM bytes.',
                  'command': 'rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python bench.py %s (two separate passes)' % sys.argv[3]}, indent=1))
