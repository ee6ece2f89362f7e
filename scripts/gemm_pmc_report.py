"""Join scripts/gemm_pmc.py's launch order with rocprofv3 --pmc counter CSVs (FETCH_SIZE pass and WRITE_SIZE pass)."""
import csv, glob, json, sys
order = json.load(open(sys.argv[1]))
def series(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'gemm2_' in r['Kernel_Name'] and r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return [float(r['Counter_Value']) for r in rows]
fs, ws = series(sys.argv[2], 'FETCH_SIZE'), series(sys.argv[3], 'WRITE_SIZE')
print('gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE (KB) x2 for 16-B/lane reads; WRITE_SIZE (KB) x1 '
      '(calibrated here on act_bwd: 23,307 KB reported for a 23,040 KB store)')
print('%-44s %10s %10s %7s %10s %10s %7s' % ('(ta,tb,M,N,K,epi,split,bm,stages)', 'algo_rd_MB', 'L2miss_MB', 'x', 'algo_wr_MB', 'wr_MB', 'x'))
i = 0
for o in order:
    n = o['reps']
    f = sum(fs[i:i + n]) / n * 1024 * 2 / 1e6
    w = sum(ws[i:i + n]) / n * 1024 / 1e6
    i += n
    ar, aw = o['algo_read_bytes'] / 1e6, o['algo_write_bytes'] / 1e6
    print('%-44s %10.1f %10.1f %7.2f %10.1f %10.1f %7.2f' % (str(tuple(o['shape'])), ar, f, f / ar, aw, w, w / aw))
