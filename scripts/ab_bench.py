"""A/B of environment knobs on one box: python scripts/ab_bench.py "KNOB=1" ["KNOB2=1" ...] -- runs bench.py alternately without / with each
setting (same process order A B A B) and prints tokens/s, ms/round and the mat-vec average."""
import os; os.environ.setdefault("EH_LAB_PLUGIN", "1")      # lab knobs / stamp kernels live in the --lab build of the plugin only
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [("base", {})] + [(v, dict([v.split("=", 1)])) for v in sys.argv[1:]]
reps = int(os.environ.get("AB_REPS", "2"))
for r in range(reps):
    for name, env in variants:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--no-cpu-baseline", "--no-rocprof", "--no-extra"], env=e, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print("%-40s %8.1f tok/s  %6.3f ms/round  plain %6.1f  mat-vec raw %.2f us" % (name, d["value"], d["ms_per_step"], d["plain_decode_tokens_per_s"], d["roofline"]["hip_events"]["avg_launch_us_raw_event_pair"]), flush=True)
        except Exception as ex:
            print(name, "FAILED", ex, out.stderr[-500:], flush=True)
