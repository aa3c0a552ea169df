"""gpurun_out/bench_round/*.json (tools/bench_round.sh) -> profiles/<tag>_bench_round.json: {run name: bench record}.
  python tools/collect_bench_round.py r04"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
out = {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "bench_round", "*.json"))):
    lines = [ln for ln in open(f) if ln.startswith("{")]
    if lines:
        out[os.path.basename(f)[:-5]] = json.loads(lines[-1])
with open(os.path.join(ROOT, "profiles", f"{tag}_bench_round.json"), "w") as fh:
    json.dump(out, fh, indent=1)
for k, d in out.items():
    t = d.get("train") or {}
    print(f"{k:28s} {d['ms_per_step']:8.3f} ms  median {d['median_ms_per_step']:8.3f}  frac {d['roofline']['frac']:.3f}  train-leg {t.get('ms_per_step')}")
