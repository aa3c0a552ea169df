#!/bin/bash
# Mutation check of the parity suite ("a deliberately wrong tableau coefficient turns the suite red").
# Builds mutated copies of the library under gpurun_ab/mut/ (git-ignored scratch that travels to the GPU box) -- run with
# `tools/mutation_check.sh build` HERE (hipcc cross-compiles), then `tools/mutation_check.sh run` on the GPU box: every mutant
# must FAIL the forward parity tests, the unmutated library must pass them.  Summary -> gpurun_out/mutation_check.txt.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
MUT=$ROOT/gpurun_ab/mut
declare -A SED
SED[rk4_stage3]='s/c.c1\[0\] = -third;/c.c1[0] = -0.25f;/'                                   # x3 = y + h (k2 - k1/3): -1/3 -> -1/4
SED[rk4_weights_classic]='s/c.c2\[0\] = 0.125f;/c.c2[0] = 0.1666667f;/;s/c.c2\[3\] = 0.125f;/c.c2[3] = 0.1666667f;/;s/c.c2\[1\] = 0.375f;/c.c2[1] = 0.3333333f;/;s/c.c2\[2\] = 0.375f;/c.c2[2] = 0.3333333f;/'
SED[midpoint_half]='s/c.c1\[0\] = 0.5f;/c.c1[0] = 0.45f;/'
SED[codec_convt_tap]='s/if (parity == 0) { k = t == 0 ? 1 : 3;/if (parity == 0) { k = t == 0 ? 3 : 1;/'        # frame decoder: kernel rows of the even output rows swapped
SED[codec_enc_slope]='s/v = w2\[((size_t)co \* kEncMid + 4 \* kq + j) \* 9 + tap\];/v = w2[((size_t)co * kEncMid + 4 * kq + j) * 9 + (8 - tap)];/'  # frame encoder: second conv's filter flipped
SED[wino5_bt_coef]='s/out\[0\] = fma2(4.0f, in\[0\], fma2(-5.0f, in\[2\], in\[4\]));/out[0] = fma2(4.0f, in[0], fma2(-4.0f, in[2], in[4]));/'      # F(2x2,5x5): B^T row 0: -5 -> -4
SED[wino5_at_coef]='s/^  return i == 0 ? 0.0f : (i == 1 ? 1.0f : (i == 2 ? -1.0f : (i == 3 ? 2.0f : (i == 4 ? -2.0f : 1.0f))));/  return i == 0 ? 0.0f : (i == 1 ? 1.0f : (i == 2 ? -1.0f : (i == 3 ? 2.0f : (i == 4 ? -1.0f : 1.0f))));/'    # F(2x2,5x5): A^T row 1: -2 -> -1
SED[wgrad_wino_g]='s/const float hs = 0.5f \* (u\[1\]\[j\] + u\[2\]\[j\]), hd = 0.5f \* (u\[1\]\[j\] - u\[2\]\[j\]);/const float hs = 0.5f * (u[1][j] + u[2][j]), hd = 0.4f * (u[1][j] - u[2][j]);/'   # Winograd-domain weight gradient: one G entry 0.5 -> 0.4
SED[wgrad_wino_at]='s/\*(f32x4\*)(wr + (4 \* i + 2) \* kW2Plane) = t\[i\]\[0\] - t\[i\]\[1\];/*(f32x4*)(wr + (4 * i + 2) * kW2Plane) = t[i][0] + t[i][1];/'   # ... and one sign of A dY A^T
# ---- round 3: the adaptive walk, the device-driven adjoint, the saving forward, the 16-workgroup walk
SED[adapt_combine_ccur]='s/      d_cA = m.c1\[np\];/      d_cA = m.c1[np] * 1.01f;/'                                  # adaptive walk: weight of the stage's own k in an order-1 combine
SED[adapt_ew_coef]='s/    const float c1 = (m.c_dev ? ((ConstF\*)m.c_dev)\[j\] : m.c1\[j\]) \* hs;/    const float c1 = (m.c_dev ? ((ConstF*)m.c_dev)[j] : m.c1[j]) * hs * 1.01f;/'   # elementwise rows of the walk
SED[adjoint_dense_weight]='s/  const double C2 = d7 - 4.0 \* d1 - 5.0 \* b + 16.0 \* m;\n  return x \* d1/XX/;s/^__device__ double adj_dense_weight(const AdjCtl\* st, int s, double x) {  \/\/ dp5::dense_weight/__device__ double adj_dense_weight(const AdjCtl* st, int s, double x) { x *= 0.97;/'   # device controller: dense-output weights evaluated at the wrong point
SED[walk16_out_transform]='s/      val = pk_sub(pk_sub(\*(const f32x4\*)(x + 1024) + bias4, \*(const f32x4\*)(x + 2048)), \*(const f32x4\*)(x + 3072));/      val = pk_sub(pk_sub(*(const f32x4*)(x + 1024) + bias4, *(const f32x4*)(x + 3072)), *(const f32x4*)(x + 2048)) * 1.001f;/'   # 16-workgroup walk: second half of the output transform
SED[saving_slot_offset]='s/  fa.off_y1 = (long long)(6 \* BL.st);/  fa.off_y1 = (long long)(5 * BL.st);/'                # saving forward: y1 read from the wrong stage input of the slot
SED[wgrad_wino5_bt]='s/  out\[0\] = w5_fma2(4.0f, in\[0\], w5_fma2(-5.0f, in\[2\], in\[4\]));/  out[0] = w5_fma2(4.0f, in[0], w5_fma2(-4.0f, in[2], in[4]));/'   # Winograd F(2x2,5x5) weight gradient: B^T row 0
SED[codec_bwd_dec_tap]='s/          const int pa = (ky \& 1) ^ 1, ta = ky >> 1, pb = (kx \& 1) ^ 1, tb = kx >> 1;/          const int pa = (ky \& 1) ^ 1, ta = ky >> 1, pb = (kx \& 1), tb = kx >> 1;/'   # decoder backward: column parity of the second layer'"'"'s weights
SED[codec_bwd_enc_dx]='s/        const int dx = kx == 0 ? 1 : 0;/        const int dx = kx == 2 ? 1 : 0;/'   # encoder backward: which odd-column tap reads the next source column
SED[codec_bwd_dw1_centre]='s/        const float\* const bp = gm + ((2 \* iy + ky) \* kMW + 8 \* kq + kx) \* kMPix + nn;/        const float* const bp = gm + ((2 * iy + ky) * kMW + 8 * kq + kx + 1) * kMPix + nn;/'   # decoder backward: dW1 operand shifted by one column
# ---- round 4: the flow decoder's upsampling, the split 5x5 launches, the seal of an asynchronous solve
SED[upsample_src_index]='s/  float s = 0.5f \* ((float)dst + 0.5f) - 0.5f;/  float s = 0.5f * ((float)dst + 0.5f) - 0.45f;/'           # bilinear x2: source index off by 0.05 px
SED[split5_drop_partial]='s/      for (int e = 0; e < 4; ++e) tot\[e\] = k == 0 ? pk\[e\] : tot\[e\] + pk\[e\];/      for (int e = 0; e < 4; ++e) tot[e] = k <= 1 ? pk[e] : tot[e] + pk[e];/'   # split 5x5 conv: the first split'"'"'s partial is dropped
SED[seal_no_nan]='s/  for (long long i = (long long)blockIdx.x \* 256 + threadIdx.x; i < n; i += (long long)gridDim.x \* 256) o\[i\] = nan;/  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) o[i] = 0.0f;/'   # sealed async solve: unreached frames zero instead of NaN
SED[bn_bwd_mean_term]='s/  m2\[c\] = (float)(b \/ count);/  m2[c] = (float)(a \/ count);/'   # fused BatchNorm backward: the xhat term gets the wrong mean
SED[bn_running_var_biased]='s/    const double unbiased = count > 1.0 ? var \* count \/ (count - 1.0) : var;/    const double unbiased = var;/'   # running_var updated with the biased variance
SED[relu_launders_nan]='s/__device__ __forceinline__ float relu_f(float v) { return v < 0.0f ? 0.0f : v; }/__device__ __forceinline__ float relu_f(float v) { return fmaxf(v, 0.0f); }/'   # the ReLU of every epilogue back to v_max_f32: NaN -> 0
SED[evalwalk_no_nan_fill]='s/      nan_fill_row16(\*(const ConvArgs\*)((ConstArgs\*)table + l), b, cq, rq, pa.reloc, pa.out_nchw);/      (void)0;/'   # sixteen-workgroup walk: a launch whose wait gave up leaves its outputs as they are
SED[dopri5_beta32]='s|{44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0}|{44.0 / 45, -56.0 / 15, 31.0 / 9, 0, 0, 0}|'
TESTS="tests/test_hip_backward.py::test_backward_strict_on_kink_free_dynamics tests/test_hip_frame_codec.py::test_encoder_matches_reference_fixture tests/test_hip_frame_codec.py::test_decoder_matches_reference_fixture tests/test_hip_conv.py::test_winograd5_conv_matches_torch tests/test_hip_full_size.py::test_cell_and_encoder_full_channels tests/test_hip_odeint.py::test_fixed_grid_on_vigorous_dynamics_matches_reference_fixture tests/test_hip_odeint.py::test_dopri5_on_vigorous_dynamics_matches_reference_fixture tests/test_hip_odeint.py::test_fixed_grid_matches_golden_and_oracle tests/test_hip_odeint.py::test_full_size_against_oracle tests/test_hip_backward.py::test_dopri5_backward_matches_autograd_through_oracle tests/test_hip_backward.py::test_dopri5_adjoint_matches_oracle_adjoint tests/test_hip_backward.py::test_dopri5_saving_forward_equals_reintegration tests/test_hip_reference_configs.py::test_config0_as_stated_b4 tests/test_hip_encoder_backward.py::test_convgru_cell_backward_matches_autograd_through_oracle tests/test_hip_frame_codec.py::test_backward_matches_fp64_autograd tests/test_hip_vidode.py::test_upsample2x_matches_torch tests/test_hip_vidode.py::test_bn_relu_up_matches_torch tests/test_hip_backward.py::test_async_dopri5_forward_matches_the_synchronous_one tests/test_hip_errors.py::test_a_non_finite_state_is_not_laundered tests/test_hip_errors.py::test_a_lost_partner_in_a_single_evaluation_walk_is_loud"
case "${1:-}" in
build)
  for m in "${!SED[@]}"; do
    if [ -n "${MUT_ONLY:-}" ] && [[ ! " $MUT_ONLY " =~ " $m " ]]; then continue; fi
    d=$MUT/$m
    rm -rf "$d"; mkdir -p "$d/ode-rl_amd/csrc" "$d/include"
    # sources AND the unmutated build's objects with their timestamps: make then recompiles only what the mutation touches
    cp -p "$ROOT"/ode-rl_amd/csrc/*.hip "$ROOT"/ode-rl_amd/csrc/*.h "$ROOT"/ode-rl_amd/csrc/Makefile "$d/ode-rl_amd/csrc/"
    cp -p "$ROOT"/include/*.h "$d/include/"
    if [ -d "$ROOT/ode-rl_amd/csrc/build" ]; then mkdir -p "$d/ode-rl_amd/csrc/build"; cp -p "$ROOT"/ode-rl_amd/csrc/build/*.o "$d/ode-rl_amd/csrc/build/"; fi
    sed -i "${SED[$m]}" "$d"/ode-rl_amd/csrc/*.hip "$d"/ode-rl_amd/csrc/*.h
    changed=0
    for f in "$d"/ode-rl_amd/csrc/*.hip "$d"/ode-rl_amd/csrc/*.h; do
      diff -q "$f" "$ROOT/ode-rl_amd/csrc/$(basename "$f")" >/dev/null || changed=$((changed + 1))
    done
    if [ $changed -ne 1 ]; then echo "mutant $m: the pattern changed $changed files (expected exactly 1)" >&2; exit 1; fi
    make -s -C "$d/ode-rl_amd/csrc" -j8 OUT_DIR="$d/lib" >/dev/null 2>&1 || { echo "mutant $m failed to build" >&2; exit 1; }
    rm -rf "$d/ode-rl_amd/csrc/build"
    echo "built $m"
  done ;;
run)
  # MUT_ONLY="name name ...": only those mutants, appended to the summary (a GPU lease is too short for all nineteen in one go)
  out=$ROOT/gpurun_out/mutation_check.txt
  cd "$ROOT"
  bad=0
  if [ -z "${MUT_ONLY:-}" ]; then
    : > "$out"
    python -m pytest $TESTS -q > gpurun_out/mut_baseline.log 2>&1; rc=$?
    echo "unmutated library: pytest rc=$rc ($(tail -1 gpurun_out/mut_baseline.log))" | tee -a "$out"
    [ $rc -ne 0 ] && bad=1
  fi
  for d in "$MUT"/*/; do
    m=$(basename "$d")
    if [ -n "${MUT_ONLY:-}" ] && [[ ! " $MUT_ONLY " =~ " $m " ]]; then continue; fi
    ODEHIP_LIB=$d/lib/libodecgru_hip.so python -m pytest $TESTS -q > gpurun_out/mut_$m.log 2>&1; rc=$?
    echo "mutant $m: pytest rc=$rc ($(tail -1 gpurun_out/mut_$m.log)) failed: $(grep -c '^FAILED' gpurun_out/mut_$m.log)" | tee -a "$out"
    grep '^FAILED' gpurun_out/mut_$m.log | sed 's/ - .*//' >> "$out"
    [ $rc -eq 0 ] && { echo "  !! mutant $m SURVIVED" | tee -a "$out"; bad=1; }
    # the mutant must fail on PARITY (an assertion of a test), not because it could not be loaded or crashed
    if grep -qE "is missing: build it|reports ABI version|cannot open shared object|undefined symbol|AttributeError: .*odehip_|OSError|Segmentation|core dumped" gpurun_out/mut_$m.log; then
      echo "  !! mutant $m failed for the wrong reason (library not loaded / crash): rebuild the mutants" | tee -a "$out"; bad=1
    fi
  done
  exit $bad ;;
baseline)   # the unmutated library alone (starts a fresh summary; the mutants then follow with MUT_ONLY=... run, one lease at a time)
  out=$ROOT/gpurun_out/mutation_check.txt
  cd "$ROOT"
  python -m pytest $TESTS -q > gpurun_out/mut_baseline.log 2>&1; rc=$?
  echo "unmutated library: pytest rc=$rc ($(tail -1 gpurun_out/mut_baseline.log))" | tee "$out"
  exit $rc ;;
*) echo "usage: $0 build|baseline|run" >&2; exit 2 ;;
esac
