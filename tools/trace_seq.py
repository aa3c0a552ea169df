"""Durations of one kernel's launches in issue order over the tail of a rocprofv3 kernel trace (to tell apart the variants
of a kernel that share a name, e.g. forward-saving and gradient-chain launches of fstack_bf16_kernel).
  python tools/trace_seq.py <kernel_trace.csv> <name substring> [last N=80]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 80
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-n:]]
print(" ".join(f"{x:.1f}" for x in d))
