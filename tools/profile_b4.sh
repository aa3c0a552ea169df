set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_b4
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_hip_conv.py tests/test_hip_vidode.py tests/test_hip_errors.py tests/test_hip_encoder.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
python tools/model_bench.py --batch 4 --method dopri5 --steps 20 > $OUT/b4.json 2>$OUT/b4.err || { tail $OUT/b4.err; exit 1; }
cat $OUT/b4.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o m -- python3 $ROOT/tools/model_bench.py --batch 4 --method dopri5 --steps 20 --only train > $OUT/prof.log 2>&1 || { tail $OUT/prof.log; exit 1; }
rm -f $OUT/p/*/m_kernel_trace.csv $OUT/p/m_kernel_trace.csv
echo done
