// pack.hpp -- per-element bodies of the weight-packing kernels, shared by the one-tensor launches (conv_mfma.hip, conv_wino.hip)
// and by the batched launch of a whole plan (plan.hip pack_batch_kernel: ~140 launches of 4-5 us per optimisation step -> 3).
#pragma once
#include "conv_wino.hpp"

namespace mcedm {

// value of element i of the packed table dst[((chunk*taps + tap)*KC + cil) * CoutP + co] (zero-padded in ci and co)
__device__ __forceinline__ float pack_conv_value(const float* __restrict__ w, size_t i, int Cout, int Cin, int taps, int KC, int coutp,
                                                 int qkv_heads, int transpose_flip) {
  const int co = (int)(i % coutp);
  size_t t = i / coutp;
  const int cil = (int)(t % KC); t /= KC;
  const int tap = (int)(t % taps);
  const int chunk = (int)(t / taps);
  const int ci = chunk * KC + cil;
  float v = 0.f;
  if (co < Cout && ci < Cin) {
    if (!transpose_flip) {
      int cs = co;
      if (qkv_heads > 0) {  // packed row (head, which, c) <- reference row (head, c, which), adm_blocks.py:175
        const int per = Cout / qkv_heads, d = per / 3;
        const int h = co / per, rr = co % per, which = rr / d, c = rr % d;
        cs = h * per + c * 3 + which;
      }
      v = w[((size_t)cs * Cin + ci) * taps + tap];
    } else {
      // dgrad: the GEMM's output channels are the conv's input channels and its K index runs over the conv's
      // output channels; w is stored [K = Cin][Cout][taps], taps mirrored.  For the qkv conv the incoming
      // gradient rows are in packed (head, which, c) order, so K index ci reads reference row (head, c, which).
      int kk = ci;
      if (qkv_heads > 0) {
        const int per = Cin / qkv_heads, d = per / 3;
        const int h = ci / per, rr = ci % per, which = rr / d, c = rr % d;
        kk = h * per + c * 3 + which;
      }
      v = w[((size_t)kk * Cout + co) * taps + (taps - 1 - tap)];
    }
  }
  return v;
}

// U = G g G^T for one (cout, cin) = element idx of [coutp][nch * WKC]: G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
// tflip: the data-gradient weights, w'[co][ci][a][b] = w[ci][co][2 - a][2 - b] (w is then [Cin][Cout][3][3])
__device__ __forceinline__ void wino_pack_elem(const float* __restrict__ w, float* __restrict__ dst, int idx, int Cout, int Cin, int coutp,
                                               int tflip) {
  const int co = idx % coutp, ci = idx / coutp;
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float v = 0.f;
      if (co < Cout && ci < Cin) v = tflip ? w[((size_t)ci * Cout + co) * 9 + (2 - a) * 3 + (2 - b)] : w[((size_t)co * Cin + ci) * 9 + a * 3 + b];
      g[a][b] = v;
    }
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
    t[3][b] = g[2][b];
  }
  const int chunk = ci / WKC, k = ci % WKC, mb = co / 32, lane = (co & 31) + 32 * (k & 1), s = k >> 1;
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) {
    // column nu = 2 is stored NEGATED: the kernel's input transform produces -V[xi][2] (t1 - t2 instead of t2 - t1, which lets
    // the four outputs of a row come out of two packed additions), and (-U) * (-V) = U * V bit for bit
    const float u[4] = {t[xi][0], 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]), -(0.5f * (t[xi][0] - t[xi][1] + t[xi][2])), t[xi][2]};
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
      dst[((((size_t)chunk * (coutp / 32) + mb) * 16 + 4 * xi + nu) * 64 + lane) * 4 + s] = u[nu];
  }
}

}  // namespace mcedm
