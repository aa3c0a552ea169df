#!/bin/bash
# Ablation builds of the one-launch bf16 kernels (timing only: results are wrong by construction).  `build` here, `run` on the GPU box.
# Variants = macros BTRAJ_ABLATE_<v> in csrc/btraj_bf16.hip: no_bias (no bias-gradient sums), no_gstore (conv-output gradients not
# stored), no_mask (saved masks not loaded), no_go (grad_out not loaded).  (The forward kernel's ablations -- row barriers 1.45 us,
# tile rewrites 1.3 us, stage epilogue 0.6 us, activation-fragment latency 0.5 us per evaluation -- were taken with macros that
# have since been removed from fstack_bf16.hip: DESIGN.md section 4.2c.)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
AB=$ROOT/gpurun_ab/abl
VARIANTS="${VARIANTS:-base no_bias no_gstore no_mask no_go}"
case "${1:-}" in
build)
  for v in $VARIANTS; do
    d=$AB/$v; rm -rf "$d"; mkdir -p "$d/ode-rl_amd/csrc" "$d/include"
    cp "$ROOT"/ode-rl_amd/csrc/*.hip "$ROOT"/ode-rl_amd/csrc/*.h "$ROOT"/ode-rl_amd/csrc/Makefile "$d/ode-rl_amd/csrc/"
    cp "$ROOT"/include/*.h "$d/include/"
    make -s -C "$d/ode-rl_amd/csrc" -j8 OUT_DIR="$d/lib" CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -mllvm -amdgpu-kernarg-preload-count=8 -DFTRAJ_ABLATE_$v -DBTRAJ_ABLATE_$v" >/dev/null 2>&1 || { echo "$v failed"; exit 1; }
    rm -rf "$d/ode-rl_amd/csrc/build"; echo built $v
  done ;;
run)
  cd "$ROOT"
  for v in $VARIANTS; do
    ODEHIP_LIB=$AB/$v/lib/libodecgru_hip.so python bench.py --dtype bf16 --batch ${2:-128} --frames 40 --steps 10 --no-cpu-baseline --no-config0 ${3:-} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['median_ms_per_step'],3), 'ms;', round(d['median_ms_per_step']*1e3/156,2), 'us per f; train leg', d['train'] and round(d['train']['median_ms_per_step'],3))"
  done ;;
esac
