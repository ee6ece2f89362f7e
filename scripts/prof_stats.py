"""Summarise a rocprofv3 --kernel-trace --stats CSV run: per-kernel calls / total / avg / percent."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv under", d); sys.exit(1)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms over %d kernel names" % (tot / 1e6, len(rows)))
print("%-100s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-100s %8s %12.1f %10.2f %6.2f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                             float(r["AverageNs"]) / 1e3, float(r["Percentage"])))

fam = {}
for r in rows:
    n = r["Name"]
    k = "GEMM family (gemm2_kernel* + pp_kernel* + gemm_nt_kernel*)" if any(x in n for x in ("gemm2_", "gemm_nt_kernel", "pp_kernel", "pp_group_kernel")) else None
    if k:
        a = fam.setdefault(k, [0, 0.0])
        a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
for k, (c, t) in fam.items():
    print("%-100s %8d %12.1f %10.2f %6.2f" % (k, c, t / 1e3, t / 1e3 / c, 100 * t / tot))
