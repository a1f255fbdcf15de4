"""Summaries of the rocprofv3 kernel trace bench.py's child run leaves in gpurun_out/bench_rocprof_kernel_trace.csv, cut at the plugin's
marker kernels (the K timed-equivalent rounds only):
    python scripts/bench_trace_summary.py gpurun_out/bench_rocprof_kernel_trace.csv BENCH.json profiles/r02
writes profiles/r02_bench_kernel_stats.md (per-kernel table) and profiles/r02_rounds_roofline.json (mat-vec time per round, algorithmic
bytes, HBM fraction) -- the figures bench.py's roofline.{achieved, frac} are computed from."""
import csv, json, sys, collections, re
trace, bench_json, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_profile_mark" in r["Kernel_Name"]]
sel = rows[marks[-2] + 1:marks[-1]]
b = json.loads(open(bench_json).read().strip().splitlines()[-1])
steps = b["steps"]
MV = ("k_mmt<", "k_mmt2<", "k_mmt_bb<", "k_mmq<", "k_mmvq<")
def short(n): return re.sub(r"\(.*", "", n).replace("void ", "")[:70]
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
lines = [f"rocprofv3 --kernel-trace of `python bench.py` (child run), the {steps} rounds between the plugin's marker kernels\n",
         "| kernel | calls | calls/round | avg us | total ms | % |", "|---|---:|---:|---:|---:|---:|"]
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("| `%s` | %d | %.1f | %.2f | %.3f | %.1f |" % (k, n, n / steps, t / n / 1e3, t / 1e6, 100 * t / tot))
mv = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in sel if any(m in r["Kernel_Name"] for m in MV)]
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
rp = b["roofline"]
bpl = rp["algorithmic_bytes_per_launch"]
avg_us = sum(mv) / len(mv) / 1e3
lines.append("\nall kernels: %.3f ms per round (%d dispatches per round); wall under the profiler %.3f ms per round" % (tot / 1e6 / steps, len(sel) / steps, span / 1e6 / steps))
lines.append("quantised mat-vec family: %d launches (%.0f per round), average %.2f us, %.3f ms per round, %.1f%% of kernel time" % (len(mv), len(mv) / steps, avg_us, sum(mv) / 1e6 / steps, 100 * sum(mv) / tot))
lines.append("algorithmic bytes per launch %d -> %.1f GB/s = %.4f of 8 TB/s" % (bpl, bpl / avg_us / 1e3, bpl / avg_us / 1e3 / 8000))
open(prefix + "_bench_kernel_stats.md", "w").write("\n".join(lines) + "\n")
json.dump({"rounds": steps, "matvec_launches": len(mv), "matvec_avg_us": round(avg_us, 3), "matvec_ms_per_round": round(sum(mv) / 1e6 / steps, 4),
           "all_kernel_ms_per_round": round(tot / 1e6 / steps, 4), "algorithmic_bytes_per_launch": bpl, "algorithmic_bytes_per_round": round(bpl * len(mv) / steps),
           "achieved_GBps": round(bpl / avg_us / 1e3, 1), "frac_of_8TBps": round(bpl / avg_us / 1e3 / 8000, 4),
           "bench_line_frac": rp["frac"], "bench_line_value_tokens_per_s": b["value"]}, open(prefix + "_rounds_roofline.json", "w"), indent=1)
print("\n".join(lines[-3:]))
