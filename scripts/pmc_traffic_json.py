"""fabric-side bytes per GEMM launch from two rocprofv3 --pmc output directories (FETCH_SIZE pass, WRITE_SIZE pass) -> JSON on stdout.
    python scripts/pmc_traffic_json.py <fetch dir> <write dir> "<bench arguments of the passes>" """
import csv
import glob
import json
import sys


def fam_avg(d, counter):
    n = s = 0
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if ('gemm2_' in r['Kernel_Name'] or 'gemm_nt_kernel' in r['Kernel_Name']) and r['Counter_Name'] == counter:
                n += 1
                s += float(r['Counter_Value'])
    return n, (s / n if n else None)


nf, f = fam_avg(sys.argv[1], 'FETCH_SIZE')
nw, w = fam_avg(sys.argv[2], 'WRITE_SIZE')
print(json.dumps({'kernel': 'gemm2_kernel + gemm2_group_kernel + gemm_nt_kernel', 'launches_counted': nf,
                  'fetch_kb_per_launch_reported': f, 'write_kb_per_launch_reported': w,
                  'read_bytes_per_launch': f * 1024 * 2, 'write_bytes_per_launch': w * 1024,
                  'traffic_bytes_per_launch': f * 1024 * 2 + w * 1024,
                  'corrections': 'FETCH_SIZE (KB) doubled: gfx950 rocprofv3 tallies 128-B requests at 64 B for 16-B/lane reads (MI355X_MICROARCH.md, HBM); '
                                 'WRITE_SIZE (KB) x1, calibrated on act_bwd (23,307 KB reported for a 23,040 KB store). FETCH_SIZE counts L2-miss requests on the '
                                 'fabric side, Infinity-Cache hits included: an upper bound on HBM bytes.',
                  'command': 'rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python bench.py %s (two separate passes)' % sys.argv[3]}, indent=1))
