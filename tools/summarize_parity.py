"""gpurun_out/parity_observed.jsonl (tests/conftest.py::record, appended by a `pytest -m gpu` run) + that run's log ->
profiles/<tag>_parity_observed.json: the observed value of every record()ed parity assertion (max over repeats), the tail of the
suite's log and the commit it ran on (VERDICT r03 #1b).
  python tools/summarize_parity.py r04 gpurun_out/<suite log> [git head of the run]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, log = sys.argv[1], sys.argv[2]
head = sys.argv[3] if len(sys.argv) > 3 else subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
obs = {}
with open(os.path.join(ROOT, "gpurun_out", "parity_observed.jsonl")) as fh:
    for ln in fh:
        r = json.loads(ln)
        k = f"{r['test']}::{r['name']}"
        obs[k] = max(obs.get(k, 0.0), r["value"])
tail = [ln.rstrip("\n") for ln in open(os.path.join(ROOT, log))][-6:]
out = {"note": "observed value of every record()ed parity assertion of the GPU suite (max over parametrisations and repeats); the tolerances are "
               "stated in the tests", "command": "python -m pytest tests -m gpu -x -q (one process, fresh MI355X box through gpurun)",
       "git_head": head, "suite_log_tail": tail, "observed": dict(sorted(obs.items()))}
with open(os.path.join(ROOT, "profiles", f"{tag}_parity_observed.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(head, tail[-1], len(obs), "records")
