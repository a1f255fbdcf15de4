"""Timeline view of a rocprofv3 --kernel-trace CSV: per-kernel busy time, the idle gap in front of every dispatch, and the
busy / idle split of the steady-state part of the run.
    python scripts/trace_timeline.py gpurun_out/profN [first_fraction] [out.md]
`first_fraction` (default 0.5): dispatches before that fraction of the trace (model upload, tiling, warm-up) are skipped."""
import csv, glob, sys, collections, re
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(len(rows) * frac):]
def short(n):
    n = re.sub(r'\(.*', '', n); n = n.replace('void ', '')
    return n[:60]
busy = collections.defaultdict(float); gap = collections.defaultdict(float); cnt = collections.Counter()
prev_end = None; tot_busy = 0; tot_gap = 0; big_gaps = 0; big_gap_t = 0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp']); n = short(r['Kernel_Name'])
    busy[n] += e - s; cnt[n] += 1; tot_busy += e - s
    if prev_end is not None:
        g = max(0, s - prev_end)
        if g > 50000: big_gaps += 1; big_gap_t += g          # host-side waits (sampling, round boundaries), not launch gaps
        else: gap[n] += g; tot_gap += g
    prev_end = max(prev_end or 0, e)
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
out = ["| kernel | calls | avg busy us | avg gap before us | busy ms | gap ms |", "|---|---:|---:|---:|---:|---:|"]
for n in sorted(busy, key=lambda n: -(busy[n] + gap[n])):
    out.append("| `%s` | %d | %.2f | %.2f | %.2f | %.2f |" % (n, cnt[n], busy[n]/cnt[n]/1e3, gap[n]/cnt[n]/1e3, busy[n]/1e6, gap[n]/1e6))
out.append("\n%d dispatches over %.2f ms: busy %.2f ms (%.1f%%), launch gaps %.2f ms (%.1f%%), %d host waits > 50 us totalling %.2f ms" %
           (len(rows), span/1e6, tot_busy/1e6, 100*tot_busy/span, tot_gap/1e6, 100*tot_gap/span, big_gaps, big_gap_t/1e6))
txt = "\n".join(out); print(txt)
if len(sys.argv) > 3: open(sys.argv[3], "w").write(txt + "\n")
