"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: python scripts/prof_summary.py gpurun_out/profN [out.md]"""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
lines = ["| kernel | calls | avg us | total ms | % |", "|---|---:|---:|---:|---:|"]
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:18]:
    lines.append("| `%s` | %s | %.1f | %.2f | %.1f |" % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
lines.append("\ntotal kernel time %.2f ms" % (tot/1e6))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
