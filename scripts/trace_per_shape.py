"""Per-shape duration of the mat-vec / attention launches in bench.py's rocprofv3 trace (gpurun_out/bench_rocprof_kernel_trace.csv): the kernel
names alone mix shapes (wo, q|k|v, ffn_down all run `k_mmt<12, false, true, ...>`); the position of a launch in the layer tells them apart.
    python scripts/trace_per_shape.py gpurun_out/bench_rocprof_kernel_trace.csv [rounds] > profiles/r02_per_shape_launch_us.md
This split is what exposed the RoPE epilogue (q|k|v took as long as the twice-as-big gate|up launch)."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_profile_mark' in r['Kernel_Name']]
sel = rows[marks[-2] + 1:marks[-1]]
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")
n = len(sel) // R
agg = collections.defaultdict(list)
for r in range(R):
    seq = sel[r * n:(r + 1) * n]
    names = [short(x['Kernel_Name']) for x in seq]
    durs = [(int(x['End_Timestamp']) - int(x['Start_Timestamp'])) / 1000 for x in seq]
    first_verify = next((i for i, nm in enumerate(names) if nm.startswith('k_get_rows_f16_rows') and i > 8), 41)      # the verification's embedding gather follows the chain
    for i, (nm, d) in enumerate(zip(names, durs)):
        prev = names[i - 1] if i > 0 else ''; nxt = names[i + 1] if i + 1 < len(names) else ''
        key = None
        if nm.startswith('k_mmt2'): key = 'q|k (Q4_K) + v (Q6_K), 32.6 MB'
        elif nm.startswith('k_mmt<12, false'):
            if nxt.startswith('k_attn'): key = 'q|k|v Q4_K, 28.3 MB'
            elif prev.startswith('k_attn'): key = 'wo Q4_K, 9.4 MB'
            elif prev.startswith('k_mmt<12, true'): key = 'ffn_down Q4_K, 25.4 MB'
            else: key = 'fc Q4_K (EAGLE), 18.9 MB'
        elif nm.startswith('k_mmt<12, true'): key = 'gate|up Q4_K, 50.7 MB'
        elif nm.startswith('k_mmt<14'): key = 'ffn_down Q6_K, 37.0 MB' if prev.startswith('k_mmt<12, true') else 'LM head Q6_K, 107.5 MB'
        elif nm.startswith('k_attn'): key = 'attention'
        if key: agg[(key, 'chain step, 1 token' if i < first_verify else 'verification, 6 tokens')].append(d)
print("rocprofv3 durations (they include the ~1.3 us boundary to the next launch), Vicuna-7B Q4_K_M + EAGLE head, %d rounds\n" % R)
print("| launch | where | per round | average us |\n|---|---|---:|---:|")
for k in sorted(agg): print('| %s | %s | %.0f | %.2f |' % (k[0], k[1], len(agg[k]) / R, sum(agg[k]) / len(agg[k])))
