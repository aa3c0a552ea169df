"""Print selected rows of a rocprofv3 kernel-stats CSV: python tools/kstats.py <stats.csv> [substring ...]"""
import csv
import sys

rows = list(csv.reader(open(sys.argv[1])))[1:]
keys = sys.argv[2:]
for r in rows:
    if not keys or any(k in r[0] for k in keys):
        print(f"{r[0][:64]:64s} n={r[1]:>5s} avg_us={float(r[3]) / 1e3:9.1f} total_us={float(r[2]) / 1e3:10.1f}")
