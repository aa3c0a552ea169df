#!/bin/bash
# Every BASELINE.json config's single-GPU share + the context lines quoted in DESIGN.md section 5, one JSON line each.
# Run on the GPU box from the repo root: tools/bench_round.sh  ->  gpurun_out/bench_round/<name>.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bench_round
mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; timeout -k 10 280 python3 bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "$name FAILED"; python3 - $OUT/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])   # (gloo prints a connection line to stdout in the rehearsal mode)
    t = d.get("train") or {}
    print(f"{sys.argv[2]:28s} {d['ms_per_step']:8.3f} ms  median {d['median_ms_per_step']:8.3f}  {d['value']:10.0f} frames/s  frac {d['roofline']['frac']:.3f}  kernel {d['roofline']['kernel'][:40]}  train-leg {t.get('ms_per_step')}")
except Exception as e:
    print(sys.argv[2], "unreadable:", e)
PY
}
run config1_headline
run config0_b4 --batch 4 --no-config0 --no-cpu-baseline --no-model
run config1_train --train --no-cpu-baseline --no-model
run dopri5_fwd --method dopri5 --no-cpu-baseline --no-config0 --no-model
run dopri5_train --method dopri5 --train --no-cpu-baseline --no-model
# configs[2]: rtol 1e-5 as BASELINE.json states; atol is unstated there = the reference's DiffEqSolver default 1e-5.
# Both adjoint norms: the seminorm, and torchdiffeq's default MIXED norm (every parameter tensor's error ratio steers the steps: an
# order of magnitude more backward steps by construction, in torchdiffeq as here)
run config2_adjoint --method dopri5 --train --adjoint --rtol 1e-5 --no-cpu-baseline --steps 20 --no-model
run config2_adjoint_mixed --method dopri5 --train --adjoint --adjoint-norm mixed --max-accept 700 --rtol 1e-5 --no-cpu-baseline --steps 3 --warmup 1 --no-model --no-config0
run config2_adjoint_atol1e-6 --method dopri5 --train --adjoint --rtol 1e-5 --atol 1e-6 --no-cpu-baseline --steps 20 --no-model --no-config0
run config3_vidode_fwd --shape V --no-cpu-baseline --no-config0 --no-model
run config3_vidode_train --shape V --train --no-cpu-baseline --no-model
run config3_vidode_dopri5_fwd --shape V --method dopri5 --no-cpu-baseline --no-config0 --no-model
run config3_vidode_dopri5_train --shape V --method dopri5 --train --no-cpu-baseline --no-config0 --no-model
run config4_bf16_fwd --dtype bf16 --batch 128 --frames 40 --no-cpu-baseline --no-config0 --steps 20 --no-model
run config4_bf16_train --dtype bf16 --batch 128 --frames 40 --train --no-cpu-baseline --steps 10 --no-model
run bf16_b64_t10_fwd --dtype bf16 --no-cpu-baseline --no-config0 --no-model
run bf16_b64_t10_train --dtype bf16 --train --no-cpu-baseline --no-model
run f32_b128_t40_fwd --batch 128 --frames 40 --no-cpu-baseline --no-config0 --steps 10 --no-train-leg --no-model
ODEHIP_PERSISTENT=0 run config1_per_layer --no-cpu-baseline --no-config0 --no-train-leg --no-model
ODEHIP_BENCH_REHEARSAL=1 run rehearsal_2ranks_one_gpu --gpus 2 --steps 10 --no-cpu-baseline --no-model
echo bench_round done
