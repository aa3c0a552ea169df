"""Per-kernel time of the LAST fraction of a rocprofv3 kernel trace (steady state, after MIOpen's find phase).
  python tools/trace_tail.py <kernel_trace.csv> [fraction=0.25]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t1 - (t1 - t0) * frac
agg = defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) >= cut:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"window {(t1 - cut) / 1e6:.2f} ms, kernel time {tot / 1e6:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{v[1] / tot * 100:6.2f}% {v[1] / 1e6:8.3f} ms {v[0]:6d} x {v[1] / v[0] / 1e3:8.1f} us  {k[:110]}")
