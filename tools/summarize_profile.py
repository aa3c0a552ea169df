"""Turns gpurun_out/prof (tools/profile_round.sh) into profiles/<tag>_{kernel_stats.csv,train_kernel_stats.csv,rocprof_summary.json}.
  python tools/summarize_profile.py r01"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = None  # the dominant kernel of the forward run (first row of its kernel stats)


def stats(path):
    return list(csv.DictReader(open(path)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof")
    dst = os.path.join(ROOT, "profiles")
    fwd = glob.glob(os.path.join(src, "fwd", "**", "*kernel_stats.csv"), recursive=True)[0]
    trn = glob.glob(os.path.join(src, "train", "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(fwd, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    shutil.copy(trn, os.path.join(dst, f"{tag}_train_kernel_stats.csv"))
    global KERNEL
    KERNEL = stats(fwd)[0]["Name"].split("(")[0]
    pmc = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            pmc[k] = sum(v) / len(v)
    notes = {
        "command": "tools/profile_round.sh: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 "
                   "--no-cpu-baseline ; separate --kernel-trace --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*); train: bench.py --train",
        "units": "FETCH_SIZE/WRITE_SIZE in KiB per launch; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of a "
                 "wide coalesced 16 B/lane stream -> doubled",
        "dominant_kernel": KERNEL,
    }
    if "persist" in KERNEL:  # one launch = the whole trajectory (B=64, T=10, rk4: 180 layers)
        notes.update({"algorithmic_flop_per_launch": 180 * 1207959552, "executed_mfma_flop_per_launch": 180 * 536870912,
                      "algorithmic_bytes_per_launch_fused_ideal": 64 * 11 * 65536 + 5 * 147712,
                      "algorithmic_bytes_per_launch_per_layer_io": 180 * 8536064})
    else:
        notes.update({"algorithmic_bytes_per_launch": 8536064, "algorithmic_flop_per_launch": 1207959552,
                      "executed_mfma_flop_per_launch": 536870912})
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    notes["source_sha256"] = bench.source_digest()      # bench.py reports `traffic` only while the kernel sources still match
    notes["source_files"] = list(bench.TRAFFIC_SOURCES)
    try:
        notes["git_head"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        notes["git_head"] = None
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        notes["hbm_bytes_per_launch"] = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
    out = {"forward_kernel_stats": stats(fwd)[:12], "train_kernel_stats": stats(trn)[:16], "pmc_dominant_kernel_per_launch": pmc,
           "notes": notes}
    json.dump(out, open(os.path.join(dst, f"{tag}_rocprof_summary.json"), "w"), indent=1)
    line = [l for l in open(os.path.join(src, "fwd.log")) if l.startswith("{")]
    if line:
        open(os.path.join(dst, f"{tag}_bench_line_under_rocprof.json"), "w").write(line[-1])
    print(json.dumps({"pmc": pmc, "hbm_bytes_per_launch": notes.get("hbm_bytes_per_launch"),
                      "fwd_top": {k: stats(fwd)[0][k] for k in ("Name", "Calls", "AverageNs")}}, indent=1))


if __name__ == "__main__":
    main()
