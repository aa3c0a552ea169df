#!/bin/bash
# rocprofv3 kernel stats of the end-to-end ODEConvGRU training step (tools/model_bench.py --only train), fp32 and bf16.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_model
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/model_bench.py --only train --steps 2 > $OUT/warm.log 2>&1 || exit 1   # fills MIOpen's user find-db so its search kernels stay out of the stats
for D in f32 bf16; do
  python3 $ROOT/tools/model_bench.py --only train --steps 20 --dtype $D > $OUT/$D.plain.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$D -o m -- python3 $ROOT/tools/model_bench.py --only train --steps 10 --dtype $D > $OUT/$D.log 2>&1 || exit 1
  rm -f $OUT/$D/*/m_kernel_trace.csv $OUT/$D/m_kernel_trace.csv
done
echo profile_model done
