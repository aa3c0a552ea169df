#!/bin/bash
# rocprofv3 kernel stats of ODEConvGRUCell.forward alone (B=64, T_in=10, 64 channels): which kernels make up a frame.
# Output: gpurun_out/prof_encoder/..._kernel_stats.csv
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_encoder
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o m -- python3 $ROOT/tools/encoder_bench.py --steps 20 > $OUT/encoder.log 2>&1 || exit 1
rm -f $OUT/m_kernel_trace.csv $OUT/*/m_kernel_trace.csv
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/encoder_kernel_stats.csv
echo profile_encoder done
