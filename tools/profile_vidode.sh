#!/bin/bash
# rocprofv3 kernel stats of VidODE's training step at the per-GPU batch of BASELINE configs[3] (tools/vidode_bench.py --only train).
# The first, un-profiled run warms MIOpen's find cache (a cold cache shows up as naive convolution kernels in the profile).
#   -> gpurun_out/prof_vidode/{bench.json, p/m_kernel_stats.csv}
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_vidode
mkdir -p $OUT
python3 $ROOT/tools/vidode_bench.py --steps 6 > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o m -- python3 $ROOT/tools/vidode_bench.py --steps 5 --only train > $OUT/prof.log 2>&1 || { tail $OUT/prof.log; exit 1; }
rm -f $OUT/p/*/m_kernel_trace.csv $OUT/p/m_kernel_trace.csv
echo done
