"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: python scripts/prof_summary.py gpurun_out/profN [out.md]
Adds the launch-weighted average of the quantised mat-vec family (k_mmq* + k_mmvq*), the figure bench.py's
roofline.avg_launch_us has to agree with."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
lines = ["| kernel | calls | avg us | total ms | % |", "|---|---:|---:|---:|---:|"]
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    lines.append("| `%s` | %s | %.1f | %.2f | %.1f |" % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
lines.append("\ntotal kernel time %.2f ms" % (tot/1e6))
fam = [r for r in rows if 'k_mmq<' in r['Name'] or 'k_mmvq<' in r['Name']]
if fam:
    n = sum(int(r['Calls']) for r in fam); t = sum(float(r['TotalDurationNs']) for r in fam)
    lines.append("quantised mat-vec family (k_mmq* + k_mmvq*): %d launches, %.2f ms, average %.2f us per launch, %.1f%% of kernel time" % (n, t/1e6, t/1e3/n, 100*t/tot))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
