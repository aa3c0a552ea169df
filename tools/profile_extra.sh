#!/bin/bash
# rocprofv3 kernel stats of the secondary configurations (dopri5 forward + backward; bf16 forward and forward + backward at B=128,
# T=40; the adaptive adjoint of configs[2]).
# Output: gpurun_out/prof_extra/{dopri5,bf16}/..._kernel_stats.csv + the bench lines printed under the profiler.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_extra
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dopri5 -o m -- python3 $ROOT/bench.py --method dopri5 --train --steps 10 --warmup 2 --no-cpu-baseline --no-model > $OUT/dopri5.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf16 -o m -- python3 $ROOT/bench.py --dtype bf16 --batch 128 --frames 40 --train --steps 5 --warmup 2 --no-cpu-baseline --no-model > $OUT/bf16.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf16_fwd -o m -- python3 $ROOT/bench.py --dtype bf16 --batch 128 --frames 40 --steps 10 --warmup 2 --no-cpu-baseline --no-model --no-config0 --no-train-leg > $OUT/bf16_fwd.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adjoint -o m -- python3 $ROOT/bench.py --method dopri5 --train --adjoint --rtol 1e-5 --steps 10 --warmup 2 --no-cpu-baseline --no-model > $OUT/adjoint.log 2>&1 || exit 1
rm -f $OUT/*/m_kernel_trace.csv $OUT/*/*/m_kernel_trace.csv
echo profile_extra done
