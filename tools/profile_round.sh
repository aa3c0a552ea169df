#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline object (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench (forward, BASELINE configs[1]) and of --train;
#   2. separate --pmc passes (never combined with trace domains other than kernel-trace): FETCH_SIZE, WRITE_SIZE, SQ_*.
# Output: gpurun_out/prof/{fwd,train,pmc_*}; tools/summarize_profile.py turns them into profiles/<round>_*.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config0 --no-train-leg --no-model"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fwd -o fwd -- $B > $OUT/fwd.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- python3 $ROOT/bench.py --train --steps 5 --warmup 2 --no-cpu-baseline > $OUT/train.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config0 --no-train-leg --no-model > $OUT/pmc_$N.log 2>&1 || exit 1
done
echo profile_round done
