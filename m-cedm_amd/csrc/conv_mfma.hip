// conv_mfma.hip -- K2/K3/K4: 3x3 (pad 1) and 1x1 convolution as an implicit GEMM on the
// fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain, 64 FLOP/clk/SIMD).
//
// Replaces, per call, the ATen sequence of models/adm_blocks.py:57-82 plus the pointwise ops
// the reference runs around it (adm_blocks.py:161,166,170-172,179):
//     out = conv(resample(act((cat(xa, xb) - mean) * scale + offset))) + bias [+ resample(res)]
// so GroupNorm-apply, FiLM, SiLU, the channel concat, the 2x up/down resampling, the bias and
// the residual add never touch HBM as separate tensors.
//
// GEMM view: D[co][pixel] = sum_{tap, ci} Wp[tap][ci][co] * X[ci][pixel + tap]
//   A operand (M = 32 output channels)  <- LDS weight slab  [tap][ci_local][MT]
//   B operand (N = 32 pixels)           <- LDS input tile   [ci_local][PH+2][PW+2]   (halo'd)
//   K is walked in chunks of KC input channels; one MFMA consumes 2 channels of one tap.
// A workgroup = 4 waves = MT output channels x (PH x PW) pixels of one sample.  HBM layout
// stays NCHW: lanes run along W, so both the staging loads and the epilogue stores are
// contiguous 128-byte row segments.
#include "common.hpp"
#include "prof.hpp"

namespace mcedm {

template <int MT_, int PH_, int PW_, int WM_, int WN_, int TAPS_, int KC_>
struct ConvCfg {
  static constexpr int MT = MT_, PH = PH_, PW = PW_, WM = WM_, WN = WN_, TAPS = TAPS_, KC = KC_;
  static constexpr int HALO = (TAPS == 9) ? 1 : 0;
  static constexpr int PITCH = PW + 2 * HALO;
  static constexpr int ROWS = PH + 2 * HALO;
  static constexpr int PLANE = ROWS * PITCH;
  static constexpr int NPIX = PH * PW;
  static constexpr int TM = MT / WM / 32;    // 32x32 accumulator tiles per wave along M
  static constexpr int TN = NPIX / WN / 32;  // ... along N
  static constexpr int XL = KC * PLANE;      // floats of the input tile
  static constexpr int WL = TAPS * KC * MT;  // floats of the weight slab
  static constexpr int NWAVE = WM * WN;       // waves that own accumulators (the rest only help staging)
  static_assert(NWAVE >= 1 && NWAVE <= 4, "at most 4 compute waves per workgroup");
  static_assert(TM >= 1 && TN >= 1 && MT % (WM * 32) == 0 && NPIX % (WN * 32) == 0, "tile shape");
  static_assert(MT % 4 == 0 && KC % 2 == 0, "vector widths");
};

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

__device__ __forceinline__ float apply_coef(float v, const Coef& c, int act) {
  float t = (v - c.mean) * c.scale + c.offset;
  return act ? silu_f(t) : t;
}

// Stage one KC-channel slab of the (transformed, resampled, zero-padded) input into LDS.
// The channel loop is outermost so that the per-(sample, channel) coefficient row and the source
// plane pointer are wave-uniform (scalar loads); all global loads of the slab are issued before
// the first one is consumed.
template <class C, int RS>
__device__ __forceinline__ void stage_input(const ConvArgs& p, float* xl, int n, int c0, int y0, int x0, int tid) {
  const int Cin = p.Ca + p.Cb;
  constexpr int NL = (RS == RS_DOWN) ? 4 : 1;
  constexpr int SUB = (C::PLANE + 255) / 256;
  float raw[C::KC][SUB][NL];
  const size_t src_plane = (size_t)p.Hs * p.Ws;
#pragma unroll
  for (int cil = 0; cil < C::KC; ++cil) {
    const int ci = c0 + cil;
    const bool in_a = ci < p.Ca;
    const float* src = in_a ? p.xa : p.xb;
    const int cc = in_a ? ci : ci - p.Ca;
    const int CC = in_a ? p.Ca : p.Cb;
    const bool chan_ok = (ci < Cin) && (src != nullptr);
    const float* plane = chan_ok ? src + ((size_t)n * CC + cc) * src_plane : nullptr;
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const int e = tid + sub * 256;
      const int r = e / C::PITCH;
      const int c = e - r * C::PITCH;
      const int y = y0 + r - C::HALO;
      const int x = x0 + c - C::HALO;
      const bool ok = chan_ok && (e < C::PLANE) && ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
#pragma unroll
      for (int q = 0; q < NL; ++q) raw[cil][sub][q] = 0.f;
      if (ok) {
        if (RS == RS_NONE) {
          raw[cil][sub][0] = plane[(size_t)y * p.Ws + x];
        } else if (RS == RS_UP) {
          raw[cil][sub][0] = plane[(size_t)(y >> 1) * p.Ws + (x >> 1)];
        } else {
          const float* q0 = plane + (size_t)(2 * y) * p.Ws + 2 * x;
          raw[cil][sub][0] = q0[0];
          raw[cil][sub][1] = q0[1];
          raw[cil][sub][2] = q0[p.Ws];
          raw[cil][sub][3] = q0[p.Ws + 1];
        }
      }
    }
  }
#pragma unroll
  for (int cil = 0; cil < C::KC; ++cil) {
    const int ci = c0 + cil;
    const bool chan_ok = (ci < Cin) && ((ci < p.Ca ? p.xa : p.xb) != nullptr);
    Coef cf{0.f, 1.f, 0.f, 0.f};
    if (chan_ok && p.coef) cf = p.coef[(p.coef_batch ? (size_t)n * Cin : 0) + ci];
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const int e = tid + sub * 256;
      const int r = e / C::PITCH;
      const int c = e - r * C::PITCH;
      const int y = y0 + r - C::HALO;
      const int x = x0 + c - C::HALO;
      const bool ok = chan_ok && ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
      float v = 0.f;
      if (ok) {
        if (RS == RS_DOWN) {
          // 2x2 box filter of the ACTIVATED source (adm_blocks.py:75-77 runs after silu(norm(x)))
          v = 0.25f * ((apply_coef(raw[cil][sub][0], cf, p.act) + apply_coef(raw[cil][sub][1], cf, p.act)) +
                       (apply_coef(raw[cil][sub][2], cf, p.act) + apply_coef(raw[cil][sub][3], cf, p.act)));
        } else {
          v = apply_coef(raw[cil][sub][0], cf, p.act);
        }
      }
      if (e < C::PLANE) xl[cil * C::PLANE + e] = v;
    }
  }
}

template <class C>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs p, int tiles_x, int tiles_y, int mtiles,
                                                        int nchunks, int coutp) {
  __shared__ __attribute__((aligned(16))) float xl[C::XL];
  __shared__ __attribute__((aligned(16))) float wl[C::WL];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / C::WN;
  const int wn = wave % C::WN;

  int bid = blockIdx.x;
  const int mt = bid % mtiles; bid /= mtiles;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * C::PH, x0 = tx * C::PW;
  const int m0 = mt * C::MT;

  f32x16 acc[C::TM][C::TN];
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int boff[C::TN];
#pragma unroll
  for (int j = 0; j < C::TN; ++j) {
    const int pix = (wn * C::TN + j) * 32 + (lane & 31);
    boff[j] = (lane >> 5) * C::PLANE + (pix / C::PW) * C::PITCH + (pix % C::PW);
  }
  const int aoff = (lane >> 5) * C::MT + wm * C::TM * 32 + (lane & 31);

  for (int ch = 0; ch < nchunks; ++ch) {
    // ---- weights: rows of MT floats out of the packed [chunk][tap][ci_local][CoutP] table
    {
      constexpr int V4_PER_ROW = C::MT / 4;
      constexpr int NV4 = C::WL / 4;
      const float* wbase = p.wpk + (size_t)ch * (C::TAPS * C::KC) * coutp + m0;
#pragma unroll
      for (int it = 0; it < (NV4 + 255) / 256; ++it) {
        const int i = tid + it * 256;
        if (i < NV4) {
          const int row = i / V4_PER_ROW;
          const int c4 = i - row * V4_PER_ROW;
          const float4 v = *reinterpret_cast<const float4*>(wbase + (size_t)row * coutp + c4 * 4);
          reinterpret_cast<float4*>(wl)[i] = v;
        }
      }
    }
    // ---- inputs
    if (p.resample == RS_NONE) stage_input<C, RS_NONE>(p, xl, n, ch * C::KC, y0, x0, tid);
    else if (p.resample == RS_UP) stage_input<C, RS_UP>(p, xl, n, ch * C::KC, y0, x0, tid);
    else stage_input<C, RS_DOWN>(p, xl, n, ch * C::KC, y0, x0, tid);
    __syncthreads();

    if (wave < C::NWAVE) {
#pragma unroll
    for (int tap = 0; tap < C::TAPS; ++tap) {
      const int ky = (C::TAPS == 9) ? tap / 3 : 0;
      const int kx = (C::TAPS == 9) ? tap % 3 : 0;
#pragma unroll
      for (int kk = 0; kk < C::KC / 2; ++kk) {
        float a[C::TM], b[C::TN];
#pragma unroll
        for (int i = 0; i < C::TM; ++i) a[i] = wl[aoff + (tap * C::KC + 2 * kk) * C::MT + i * 32];
#pragma unroll
        for (int j = 0; j < C::TN; ++j) b[j] = xl[boff[j] + 2 * kk * C::PLANE + ky * C::PITCH + kx];
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
          for (int j = 0; j < C::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    }
    __syncthreads();
  }
  if (wave >= C::NWAVE) return;

  // ---- epilogue: + bias, + (resampled) residual, store NCHW
  const size_t HW = (size_t)p.H * p.W;
#pragma unroll
  for (int j = 0; j < C::TN; ++j) {
    const int pix = (wn * C::TN + j) * 32 + (lane & 31);
    const int y = y0 + pix / C::PW;
    const int x = x0 + pix % C::PW;
    if (y >= p.H || x >= p.W) continue;
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + (wm * C::TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.Cout) {
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[co];
          if (p.res) {
            const float* rp = p.res;
            if (p.res_mode == RS_NONE) {
              v += rp[((size_t)n * p.Cout + co) * HW + (size_t)y * p.W + x];
            } else if (p.res_mode == RS_UP) {
              const int Wr = p.W >> 1;
              v += rp[((size_t)n * p.Cout + co) * (HW >> 2) + (size_t)(y >> 1) * Wr + (x >> 1)];
            } else {
              const int Wr = p.W * 2;
              const float* q0 = rp + ((size_t)n * p.Cout + co) * (HW * 4) + (size_t)(2 * y) * Wr + 2 * x;
              v += 0.25f * ((q0[0] + q0[1]) + (q0[Wr] + q0[Wr + 1]));
            }
          }
          p.out[((size_t)n * p.Cout + co) * HW + (size_t)y * p.W + x] = v;
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// host side
int conv_kc_for(int taps) { return taps == 9 ? 8 : 16; }
static int cout_padded(int Cout) { return (Cout + 31) / 32 * 32; }

size_t conv_packed_floats(int Cout, int Cin, int taps) {
  const int KC = conv_kc_for(taps);
  return (size_t)ceil_div(Cin, KC) * taps * KC * cout_padded(Cout);
}

// dst[((chunk*taps + tap)*KC + cil) * CoutP + co]; zero-padded in ci and co.
__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int taps,
                                 int KC, int coutp, int qkv_heads, int transpose_flip, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
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
        // dgrad: out channel of this GEMM = conv input channel; src w is [Cin_gemm = conv Cout][.. ]
        // here (Cout, Cin) are the GEMM's: w is stored [Cin][Cout][taps]; taps mirrored
        v = w[((size_t)ci * Cout + co) * taps + (taps - 1 - tap)];
      }
    }
    dst[i] = v;
  }
}

int launch_pack_conv(const float* w, float* dst, int Cout, int Cin, int taps, int qkv_heads, int transpose_flip,
                     hipStream_t stream) {
  const int KC = conv_kc_for(taps);
  const size_t total = conv_packed_floats(Cout, Cin, taps);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_conv_kernel, dim3(blocks), dim3(256), 0, stream, w, dst, Cout, Cin, taps, KC,
                     cout_padded(Cout), qkv_heads, transpose_flip, total);
  MCEDM_LAUNCH_CHECK("pack_conv_kernel");
  return MCEDM_OK;
}

__global__ void pack_bias_kernel(const float* __restrict__ b, float* __restrict__ dst, int Cout, int qkv_heads) {
  const int co = blockIdx.x * blockDim.x + threadIdx.x;
  if (co >= Cout) return;
  int cs = co;
  if (qkv_heads > 0) {
    const int per = Cout / qkv_heads, d = per / 3;
    const int h = co / per, rr = co % per, which = rr / d, c = rr % d;
    cs = h * per + c * 3 + which;
  }
  dst[co] = b[cs];
}

int launch_pack_bias(const float* b, float* dst, int Cout, int qkv_heads, hipStream_t stream) {
  hipLaunchKernelGGL(pack_bias_kernel, dim3(ceil_div(Cout, 256)), dim3(256), 0, stream, b, dst, Cout, qkv_heads);
  MCEDM_LAUNCH_CHECK("pack_bias_kernel");
  return MCEDM_OK;
}

template <class C>
static int launch_cfg(const ConvArgs& a, hipStream_t stream) {
  const int tiles_x = ceil_div(a.W, C::PW), tiles_y = ceil_div(a.H, C::PH);
  const int mtiles = ceil_div(a.Cout, C::MT);
  const int nchunks = ceil_div(a.Ca + a.Cb, C::KC);
  const long long blocks = (long long)a.B * tiles_x * tiles_y * mtiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) {
    set_error("conv grid out of range (%lld blocks)", blocks);
    return MCEDM_ERR_INVALID;
  }
  // algorithmic cost of this launch: 2*MAC flops; bytes = input read once + output written once + weights + residual
  static char name[96];
  if (prof_enabled())
    snprintf(name, sizeof(name), "conv_mfma_kernel<ConvCfg<%d, %d, %d, %d, %d, %d, %d>>", C::MT, C::PH, C::PW, C::WM,
             C::WN, C::TAPS, C::KC);
  const double px = (double)a.B * a.H * a.W;
  const double flops = 2.0 * px * a.Cout * (double)(a.Ca + a.Cb) * C::TAPS;
  const double bytes = 4.0 * ((double)a.B * (a.Ca + a.Cb) * a.Hs * a.Ws + px * a.Cout * (a.res ? 2 : 1) +
                              (double)a.Cout * (a.Ca + a.Cb) * C::TAPS);
  ProfScope ps(name, flops, bytes, stream);
  hipLaunchKernelGGL(conv_mfma_kernel<C>, dim3((unsigned)blocks), dim3(256), 0, stream, a, tiles_x, tiles_y, mtiles,
                     nchunks, cout_padded(a.Cout));
  MCEDM_LAUNCH_CHECK("conv_mfma_kernel");
  return MCEDM_OK;
}

static int g_force_mt = 0, g_force_ph = 0, g_force_pw = 0;   // test hook (mcedm_op_set_conv_tile); 0 = heuristic
void set_conv_tile_override(int mt, int ph, int pw) { g_force_mt = mt; g_force_ph = ph; g_force_pw = pw; }

template <int TAPS, int KC>
static int dispatch(const ConvArgs& a, hipStream_t stream) {
  const int coutp = cout_padded(a.Cout);
  if (g_force_mt) {
    const int id = g_force_mt * 10000 + g_force_ph * 100 + g_force_pw;
    MCEDM_REQUIRE(coutp % g_force_mt == 0, "conv: forced MT=%d does not divide padded Cout=%d", g_force_mt, coutp);
    switch (id) {
      case 1280832: return launch_cfg<ConvCfg<128, 8, 32, 1, 4, TAPS, KC>>(a, stream);
      case 640832: return launch_cfg<ConvCfg<64, 8, 32, 1, 4, TAPS, KC>>(a, stream);
      case 320832: return launch_cfg<ConvCfg<32, 8, 32, 1, 4, TAPS, KC>>(a, stream);
      case 1281616: return launch_cfg<ConvCfg<128, 16, 16, 1, 4, TAPS, KC>>(a, stream);
      case 641616: return launch_cfg<ConvCfg<64, 16, 16, 1, 4, TAPS, KC>>(a, stream);
      case 321616: return launch_cfg<ConvCfg<32, 16, 16, 1, 4, TAPS, KC>>(a, stream);
      case 640808: return launch_cfg<ConvCfg<64, 8, 8, 2, 2, TAPS, KC>>(a, stream);
      case 320808: return launch_cfg<ConvCfg<32, 8, 8, 1, 2, TAPS, KC>>(a, stream);
      default: set_error("conv: no such tile configuration (%d, %d, %d)", g_force_mt, g_force_ph, g_force_pw); return MCEDM_ERR_INVALID;
    }
  }
  // Pixel tile from the image width; channel tile as large as possible while the grid still
  // covers the 256 CUs about twice (each CU holds 2-3 of these workgroups).
  auto blocks_for = [&](int mt, int ph, int pw) {
    return (long long)a.B * ceil_div(a.H, ph) * ceil_div(a.W, pw) * ceil_div(a.Cout, mt);
  };
  const long long want = 512;
  if (a.W >= 24) {
    if (coutp % 128 == 0 && blocks_for(128, 8, 32) >= want) return launch_cfg<ConvCfg<128, 8, 32, 1, 4, TAPS, KC>>(a, stream);
    if (coutp % 64 == 0 && blocks_for(64, 8, 32) >= want) return launch_cfg<ConvCfg<64, 8, 32, 1, 4, TAPS, KC>>(a, stream);
    if (blocks_for(32, 8, 32) >= want) return launch_cfg<ConvCfg<32, 8, 32, 1, 4, TAPS, KC>>(a, stream);
  } else if (a.W >= 12) {
    if (coutp % 128 == 0 && blocks_for(128, 16, 16) >= want) return launch_cfg<ConvCfg<128, 16, 16, 1, 4, TAPS, KC>>(a, stream);
    if (coutp % 64 == 0 && blocks_for(64, 16, 16) >= want) return launch_cfg<ConvCfg<64, 16, 16, 1, 4, TAPS, KC>>(a, stream);
    if (blocks_for(32, 16, 16) >= want) return launch_cfg<ConvCfg<32, 16, 16, 1, 4, TAPS, KC>>(a, stream);
  }
  // small images or small grids: 8x8-pixel tiles
  if (coutp % 64 == 0) return launch_cfg<ConvCfg<64, 8, 8, 2, 2, TAPS, KC>>(a, stream);
  return launch_cfg<ConvCfg<32, 8, 8, 1, 2, TAPS, KC>>(a, stream);
}

int launch_conv(const ConvArgs& a, int taps, hipStream_t stream) {
  MCEDM_REQUIRE(taps == 9 || taps == 1, "conv: taps must be 9 or 1 (got %d)", taps);
  MCEDM_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.Cout > 0 && a.Ca + a.Cb > 0, "conv: empty shape");
  MCEDM_REQUIRE(a.out && a.wpk, "conv: null output / weights");
  if (a.resample == RS_UP) MCEDM_REQUIRE(a.Hs * 2 == a.H && a.Ws * 2 == a.W, "conv: up-resample needs H = 2*Hs");
  else if (a.resample == RS_DOWN) MCEDM_REQUIRE(a.Hs == a.H * 2 && a.Ws == a.W * 2, "conv: down-resample needs Hs = 2*H");
  else MCEDM_REQUIRE(a.Hs == a.H && a.Ws == a.W, "conv: source size mismatch");
  if (a.res && a.res_mode == RS_UP) MCEDM_REQUIRE(a.H % 2 == 0 && a.W % 2 == 0, "conv: up residual needs even size");
  return taps == 9 ? dispatch<9, 8>(a, stream) : dispatch<1, 16>(a, stream);
}

}  // namespace mcedm
