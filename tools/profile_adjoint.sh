#!/bin/bash
# rocprofv3 kernel stats of BASELINE configs[2]: dopri5 rtol=atol=1e-5, forward + adaptive adjoint backward (seminorm).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_adj
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $ROOT/bench.py --method dopri5 --train --adjoint --rtol 1e-5 --atol 1e-5 --steps 10 --warmup 2 --no-cpu-baseline --no-model > $OUT/bench.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adj -o m -- python3 $ROOT/bench.py --method dopri5 --train --adjoint --rtol 1e-5 --atol 1e-5 --steps 10 --warmup 2 --no-cpu-baseline --no-model > $OUT/adj.log 2>&1 || exit 1
rm -f $OUT/adj/*/m_kernel_trace.csv $OUT/adj/m_kernel_trace.csv
echo done
