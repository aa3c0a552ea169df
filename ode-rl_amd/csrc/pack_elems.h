// pack_elems.h -- the element routine of the Winograd F(2x2,5x5) weight pack, shared by its own launch (conv_wino5.hip) and the
// many-tensors-in-one-launch pack (layout.hip, odehip_pack_conv_weights kind 2).
#pragma once
#include <stddef.h>

namespace odehip {

// U[ct][chunk][xi 36][quad 2][co 32][4] = (G g G^T)[xi / 6][xi % 6] of filter (co = 32 ct + i, ci = 8 chunk + 4 quad + s);
// transpose_flip: the input-gradient convolution's filter g'[ci][co][ky][kx] = g[co][ci][4 - ky][4 - kx]
__device__ __forceinline__ void pack_winograd5_elem(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int transpose_flip,
                                                    int idx) {
  int r = idx;
  const int s = r & 3; r >>= 2;
  const int i = r & 31; r >>= 5;
  const int quad = r & 1; r >>= 1;
  const int xi = r % 36; r /= 36;
  const int nc = cin / 8;
  const int c = r % nc, ct = r / nc;
  const int co = ct * 32 + i, ci = 8 * c + 4 * quad + s;
  // G rows: [1, p, p^2, p^3, p^4] / prod_{q != p} (p - q) for p = 0, 1, -1, 2, -2; inf: [0, 0, 0, 0, 1]
  const double G[6][5] = {{1.0 / 4, 0, 0, 0, 0},
                          {-1.0 / 6, -1.0 / 6, -1.0 / 6, -1.0 / 6, -1.0 / 6},
                          {-1.0 / 6, 1.0 / 6, -1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 2.0 / 24, 4.0 / 24, 8.0 / 24, 16.0 / 24},
                          {1.0 / 24, -2.0 / 24, 4.0 / 24, -8.0 / 24, 16.0 / 24},
                          {0, 0, 0, 0, 1}};
  const int ur = xi / 6, uc = xi - 6 * ur;
  double acc = 0.0;
  for (int ky = 0; ky < 5; ++ky)
    for (int kx = 0; kx < 5; ++kx) {
      const float g = transpose_flip ? w[((size_t)ci * cout + co) * 25 + (4 - ky) * 5 + (4 - kx)] : w[((size_t)co * cin + ci) * 25 + ky * 5 + kx];
      acc += G[ur][ky] * G[uc][kx] * (double)g;
    }
  out[idx] = (float)acc;
}

}  // namespace odehip
