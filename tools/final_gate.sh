#!/bin/bash
# The round's closing GPU run (VERDICT r03 #1b): the full GPU suite in one process, smoke(), a 2-rank rehearsal of bench.py's N > 1 path on
# the one GPU (gloo), the default bench line.  Steps are joined with &&: nothing runs behind a failure.
#   gpurun --timeout 1200 -- 'bash tools/final_gate.sh <short head>'   ->  gpurun_out/final_*  (then tools/summarize_parity.py)
set -o pipefail
H=${1:-unknown}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/final_suite_$H.log 2>&1
rc=$?
tail -3 gpurun_out/final_suite_$H.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1 || { tail -5 gpurun_out/final_smoke.log; exit 1; }
tail -2 gpurun_out/final_smoke.log
ODEHIP_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-model > gpurun_out/final_rehearsal.log 2>&1 || { tail -5 gpurun_out/final_rehearsal.log; exit 1; }
tail -1 gpurun_out/final_rehearsal.log | cut -c1-300
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -5 gpurun_out/final_bench.err; exit 1; }
cut -c1-500 gpurun_out/final_bench.json
