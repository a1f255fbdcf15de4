"""HBM traffic of the quantised mat-vec launches from a rocprofv3 --pmc FETCH_SIZE pass:
    python scripts/pmc_summary.py gpurun_out/pmcN profiles/r02_pmc_traffic.json
FETCH_SIZE is reported in KiB; on gfx950 it tallies 128-byte requests at 64 bytes for wide streaming reads, so it is
doubled (MI355X_MICROARCH.md, HBM section).  Writes are negligible for this kernel (outputs are T*rows*4 bytes)."""
import csv, glob, json, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
n = 0; kib = 0.0; per = {}
for r in csv.DictReader(open(f)):
    if r.get('Counter_Name') != 'FETCH_SIZE': continue
    name = r['Kernel_Name']
    if not any(m in name for m in ('k_mmt<', 'k_mmt2<', 'k_mmt_bb<', 'k_mmq<', 'k_mmvq<')): continue
    v = float(r['Counter_Value']); n += 1; kib += v
    key = name.split('(')[0]; a = per.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += v
res = {"hbm_bytes_per_launch": round(kib * 1024 * 2 / max(1, n)), "launches": n, "counter": "FETCH_SIZE (KiB) x 1024 x 2 (gfx950 correction)",
       "per_kernel": {k: {"launches": a[0], "hbm_bytes_per_launch": round(a[1] * 1024 * 2 / a[0])} for k, a in per.items()}}
print(json.dumps(res, indent=1))
if len(sys.argv) > 2: json.dump(res, open(sys.argv[2], "w"), indent=1)
