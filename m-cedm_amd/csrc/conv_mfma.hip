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
#include <cstdlib>

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

// SiLU with the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each); relative error ~1e-6, far inside the
// 1e-4 parity bar, and a third of the VALU work of expf() + IEEE division in the staging path.
// SiLU with the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each); relative error ~1e-6, far inside the
// 1e-4 parity bar, and a third of the VALU work of expf() + IEEE division in the staging path.
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ float apply_coef(float v, const Coef& c, int act) {
  float t = (v - c.mean) * c.scale + c.offset;
  return act ? silu_f(t) : t;
}

// ---- staging -----------------------------------------------------------------------------------------
// Split into an issue half (global loads -> registers) and a commit half (transform + LDS writes) so the
// loads of chunk c+1 are in flight while the MFMAs of chunk c run.  Everything that does not depend on the
// channel (tile coordinates, bounds, clamped source offsets) is computed once per workgroup; the loads are
// unconditional from clamped addresses (no exec-masked branches) and out-of-image / padded-channel elements
// are zeroed at commit.  The per-(sample, channel) transform rows are wave-uniform (scalar loads) and are
// prefetched together with the inputs.
template <class C, int RS>
struct TileGeom {
  static constexpr int NL = (RS == RS_DOWN) ? 4 : 1;
  static constexpr int SUB = (C::PLANE + 255) / 256;
  int soff[SUB][NL];   // clamped source offsets inside one channel plane
  bool inb[SUB];       // element lies inside the image (else it is conv zero padding)
  bool inp[SUB];       // element index lies inside the LDS plane
};

template <class C, int RS>
__device__ __forceinline__ void make_geom(const ConvArgs& p, TileGeom<C, RS>& G, int y0, int x0, int tid) {
#pragma unroll
  for (int sub = 0; sub < TileGeom<C, RS>::SUB; ++sub) {
    const int e = tid + sub * 256;
    const int r = e / C::PITCH;
    const int c = e - r * C::PITCH;
    const int y = y0 + r - C::HALO;
    const int x = x0 + c - C::HALO;
    G.inp[sub] = e < C::PLANE;
    G.inb[sub] = G.inp[sub] && ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
    const int yc = G.inb[sub] ? y : 0, xc = G.inb[sub] ? x : 0;
    if (RS == RS_NONE) {
      G.soff[sub][0] = yc * p.Ws + xc;
    } else if (RS == RS_UP) {
      G.soff[sub][0] = (yc >> 1) * p.Ws + (xc >> 1);
    } else {
      const int o = (2 * yc) * p.Ws + 2 * xc;
      G.soff[sub][0] = o; G.soff[sub][1] = o + 1; G.soff[sub][2] = o + p.Ws; G.soff[sub][3] = o + p.Ws + 1;
    }
  }
}

template <class C, int RS>
struct InputRegs {
  float raw[C::KC][TileGeom<C, RS>::SUB][TileGeom<C, RS>::NL];
  Coef cf[C::KC];
};

template <class C, int RS>
__device__ __forceinline__ void load_input(const ConvArgs& p, const TileGeom<C, RS>& G, InputRegs<C, RS>& R, int n,
                                           int c0) {
  const int Cin = p.Ca + p.Cb;
  const size_t src_plane = (size_t)p.Hs * p.Ws;
  const float* safe = p.xa ? p.xa : p.xb;     // any valid plane for padded channels (values are discarded)
#pragma unroll
  for (int cil = 0; cil < C::KC; ++cil) {
    const int ci = c0 + cil;
    const bool in_a = ci < p.Ca;
    const float* src = in_a ? p.xa : p.xb;
    const int cc = in_a ? ci : ci - p.Ca;
    const int CC = in_a ? p.Ca : p.Cb;
    const bool chan_ok = (ci < Cin) && (src != nullptr);
    const float* plane = chan_ok ? src + ((size_t)n * CC + cc) * src_plane : safe;
    R.cf[cil] = (chan_ok && p.coef) ? p.coef[(p.coef_batch ? (size_t)n * Cin : 0) + ci] : Coef{0.f, 1.f, 0.f, 0.f};
#pragma unroll
    for (int sub = 0; sub < TileGeom<C, RS>::SUB; ++sub)
#pragma unroll
      for (int q = 0; q < TileGeom<C, RS>::NL; ++q) R.raw[cil][sub][q] = plane[G.soff[sub][q]];
  }
}

template <class C, int RS>
__device__ __forceinline__ void store_input(const ConvArgs& p, const TileGeom<C, RS>& G, float* xl,
                                            const InputRegs<C, RS>& R, int c0, int tid) {
  const int Cin = p.Ca + p.Cb;
#pragma unroll
  for (int cil = 0; cil < C::KC; ++cil) {
    const int ci = c0 + cil;
    const bool chan_ok = (ci < Cin) && ((ci < p.Ca ? p.xa : p.xb) != nullptr);
#pragma unroll
    for (int sub = 0; sub < TileGeom<C, RS>::SUB; ++sub) {
      float v;
      if (RS == RS_DOWN) {
        // 2x2 box filter of the ACTIVATED source (adm_blocks.py:75-77 runs after silu(norm(x)))
        v = 0.25f * ((apply_coef(R.raw[cil][sub][0], R.cf[cil], p.act) + apply_coef(R.raw[cil][sub][1], R.cf[cil], p.act)) +
                     (apply_coef(R.raw[cil][sub][2], R.cf[cil], p.act) + apply_coef(R.raw[cil][sub][3], R.cf[cil], p.act)));
      } else {
        v = apply_coef(R.raw[cil][sub][0], R.cf[cil], p.act);
      }
      v = (chan_ok && G.inb[sub]) ? v : 0.f;
      if (G.inp[sub]) xl[cil * C::PLANE + tid + sub * 256] = v;
    }
  }
}

// weights: rows of MT floats out of the packed [chunk][tap][ci_local][CoutP] table
template <class C>
struct WeightRegs {
  static constexpr int NV4 = C::WL / 4;
  static constexpr int IT = (NV4 + 255) / 256;
  f32x4 v[IT];   // native vector type: HIP's float4 class defeats SROA here and lands in scratch
};

template <class C>
__device__ __forceinline__ void load_weights(const float* wpk, WeightRegs<C>& R, int ch, int m0, int coutp, int tid) {
  constexpr int V4_PER_ROW = C::MT / 4;
  const float* wbase = wpk + (size_t)ch * (C::TAPS * C::KC) * coutp + m0;
#pragma unroll
  for (int it = 0; it < WeightRegs<C>::IT; ++it) {
    int i = tid + it * 256;
    if (i >= WeightRegs<C>::NV4) i = WeightRegs<C>::NV4 - 1;     // clamp instead of branching; the store is guarded
    const int row = i / V4_PER_ROW;
    const int c4 = i - row * V4_PER_ROW;
    R.v[it] = *reinterpret_cast<const f32x4*>(wbase + (size_t)row * coutp + c4 * 4);
  }
}

template <class C>
__device__ __forceinline__ void store_weights(float* wl, const WeightRegs<C>& R, int tid) {
#pragma unroll
  for (int it = 0; it < WeightRegs<C>::IT; ++it) {
    const int i = tid + it * 256;
    if (i < WeightRegs<C>::NV4) reinterpret_cast<f32x4*>(wl)[i] = R.v[it];
  }
}

// FULL: every output channel of the tile exists (m0 + MT <= Cout); RM: -1 no residual, else its Resample mode.
template <class C, bool FULL, int RM, bool STATS>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[C::TM][C::TN], int n, int m0, int y0,
                                              int x0, int wm, int wn, int lane, float* red) {
  const size_t HW = (size_t)p.H * p.W;
#pragma unroll
  for (int i = 0; i < C::TM; ++i) {
    const int cbase = m0 + (wm * C::TM + i) * 32 + 4 * (lane >> 5);     // + (r&3) + 8*(r>>2)
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cbase + (r & 3) + 8 * (r >> 2);
      const int cc = FULL ? co : (co < p.Cout ? co : p.Cout - 1);
      bv[r] = p.bias ? p.bias[cc] : 0.f;
    }
    float gs1[4] = {0.f, 0.f, 0.f, 0.f}, gs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
      const int pix = (wn * C::TN + j) * 32 + (lane & 31);
      const int y = y0 + pix / C::PW;
      const int x = x0 + pix % C::PW;
      if (y < p.H && x < p.W) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cbase + (r & 3) + 8 * (r >> 2);
          const int cc = FULL ? co : (co < p.Cout ? co : p.Cout - 1);
          const size_t plane = (size_t)n * p.Cout + cc;
          if (RM == -1) {
            rv[r] = 0.f;
          } else if (RM == RS_NONE) {
            rv[r] = p.res[plane * HW + (size_t)y * p.W + x];
          } else if (RM == RS_UP) {
            rv[r] = p.res[plane * (HW >> 2) + (size_t)(y >> 1) * (p.W >> 1) + (x >> 1)];
          } else {
            const int Wr = p.W * 2;
            const float* q0 = p.res + plane * (HW * 4) + (size_t)(2 * y) * Wr + 2 * x;
            rv[r] = 0.25f * ((q0[0] + q0[1]) + (q0[Wr] + q0[Wr + 1]));
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cbase + (r & 3) + 8 * (r >> 2);
          float v = acc[i][j][r] + bv[r];
          if (RM != -1) v += rv[r];
          if (FULL || co < p.Cout) {
            p.out[((size_t)n * p.Cout + co) * HW + (size_t)y * p.W + x] = v;
            if (STATS) { gs1[r >> 2] += v; gs2[r >> 2] += v * v; }
          }
        }
      }
    }
    if (STATS) {
      // fused GroupNorm statistics of the output tile: lane -> 32-lane half (pixels) -> LDS slot of this wave.
      // 4-channel group inside the MT tile: gl = 8 (wm TM + i) + 2 (r>>2) + (lane>>5).
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = gs1[q], b = gs2[q];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
        if ((lane & 31) == 0) {
          const int gl = (wm * C::TM + i) * 8 + 2 * q + (lane >> 5);
          red[(wn * (C::MT / 4) + gl) * 2] = a;
          red[(wn * (C::MT / 4) + gl) * 2 + 1] = b;
        }
      }
    }
  }
}

template <class C, int RS>
__device__ __forceinline__ void conv_body(const ConvArgs& p, float* xl, float* wl, int tiles_x, int tiles_y,
                                          int mtiles, int nchunks, int coutp) {
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

  TileGeom<C, RS> geom;
  make_geom<C, RS>(p, geom, y0, x0, tid);
  InputRegs<C, RS> xin;
  WeightRegs<C> win;
  load_weights<C>(p.wpk, win, 0, m0, coutp, tid);
  load_input<C, RS>(p, geom, xin, n, 0);

  for (int ch = 0; ch < nchunks; ++ch) {
    store_weights<C>(wl, win, tid);
    store_input<C, RS>(p, geom, xl, xin, ch * C::KC, tid);
    __syncthreads();
    if (ch + 1 < nchunks) {   // next chunk's global loads fly while this chunk's MFMAs run
      load_weights<C>(p.wpk, win, ch + 1, m0, coutp, tid);
      load_input<C, RS>(p, geom, xin, n, (ch + 1) * C::KC);
    }
    if (wave < C::NWAVE) {
      // Register double-buffered operand fragments: the LDS reads of k-step s+1 are issued before the MFMAs
      // of k-step s.  The tap loop stays rolled (a fully unrolled chunk pushes the prefetch registers into
      // scratch); the fragment for the next tap's first k-step is fetched at the end of the current tap.
      float fa[2][C::TM], fb[2][C::TN];
#pragma unroll
      for (int i = 0; i < C::TM; ++i) fa[0][i] = wl[aoff + i * 32];
#pragma unroll
      for (int j = 0; j < C::TN; ++j) fb[0][j] = xl[boff[j]];
#pragma unroll 1
      for (int tap = 0; tap < C::TAPS; ++tap) {
        const int toff = (C::TAPS == 9) ? (tap / 3) * C::PITCH + (tap % 3) : 0;
        const int tn = (tap + 1 < C::TAPS) ? tap + 1 : tap;         // clamped: the last prefetch is discarded
        const int toff_n = (C::TAPS == 9) ? (tn / 3) * C::PITCH + (tn % 3) : 0;
        const float* wt = wl + aoff + tap * C::KC * C::MT;
        const float* wt_n = wl + aoff + tn * C::KC * C::MT;
#pragma unroll
        for (int kk = 0; kk < C::KC / 2; ++kk) {
          const int cur = kk & 1, nxt = cur ^ 1;
          if (kk + 1 < C::KC / 2) {
#pragma unroll
            for (int i = 0; i < C::TM; ++i) fa[nxt][i] = wt[2 * (kk + 1) * C::MT + i * 32];
#pragma unroll
            for (int j = 0; j < C::TN; ++j) fb[nxt][j] = xl[boff[j] + toff + 2 * (kk + 1) * C::PLANE];
          } else {
#pragma unroll
            for (int i = 0; i < C::TM; ++i) fa[nxt][i] = wt_n[i * 32];
#pragma unroll
            for (int j = 0; j < C::TN; ++j) fb[nxt][j] = xl[boff[j] + toff_n];
          }
#pragma unroll
          for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
          // pin the order: this step's LDS reads (next fragments) first, then this step's MFMAs, so the reads'
          // latency hides under TM*TN * 64 cycles of matrix work and the wait before the MFMAs is a counted one
          __builtin_amdgcn_sched_group_barrier(0x100, C::TM + C::TN, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, C::TM * C::TN, 0);
        }
      }
    }
    __syncthreads();
  }
  // ---- epilogue: + bias, + (resampled) residual, store NCHW.  Dispatch on wave-uniform conditions once, so
  // that inside a variant the 16 residual loads of an accumulator tile are issued back to back (one wait)
  // instead of one load -> wait -> store round trip per element.
  const bool full = (m0 + C::MT <= p.Cout);
  const int rm = p.res ? p.res_mode : -1;
  float* red = xl;            // the input tile is dead after the last chunk's closing barrier
  if (wave < C::NWAVE) {
    if (p.gsum) {
      if (full) {
        if (rm == -1) conv_epilogue<C, true, -1, true>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
        else if (rm == RS_NONE) conv_epilogue<C, true, RS_NONE, true>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
        else if (rm == RS_UP) conv_epilogue<C, true, RS_UP, true>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
        else conv_epilogue<C, true, RS_DOWN, true>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
      } else {
        if (rm == -1) conv_epilogue<C, false, -1, true>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
        else if (rm == RS_NONE) conv_epilogue<C, false, RS_NONE, true>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
        else if (rm == RS_UP) conv_epilogue<C, false, RS_UP, true>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
        else conv_epilogue<C, false, RS_DOWN, true>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
      }
    } else if (full) {
      if (rm == -1) conv_epilogue<C, true, -1, false>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
      else if (rm == RS_NONE) conv_epilogue<C, true, RS_NONE, false>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
      else if (rm == RS_UP) conv_epilogue<C, true, RS_UP, false>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
      else conv_epilogue<C, true, RS_DOWN, false>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
    } else {
      if (rm == -1) conv_epilogue<C, false, -1, false>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
      else if (rm == RS_NONE) conv_epilogue<C, false, RS_NONE, false>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
      else if (rm == RS_UP) conv_epilogue<C, false, RS_UP, false>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
      else conv_epilogue<C, false, RS_DOWN, false>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
    }
  }
  if (p.gsum) {               // wave-uniform: every wave of the workgroup reaches this barrier
    __syncthreads();
    constexpr int NG2 = C::MT / 4 * 2;            // (sum, sumsq) pairs of the tile's 4-channel groups
    if (tid < NG2) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < C::WN; ++w) t += red[w * NG2 + tid];        // fixed order: bitwise reproducible
      const int g = m0 / 4 + tid / 2;
      const int ngroups = (p.Cout + 3) / 4;
      const int ntiles = tiles_x * tiles_y;
      if (g < ngroups) p.gsum[(((size_t)n * ntiles + ty * tiles_x + tx) * ngroups + g) * 2 + (tid & 1)] = t;
    }
  }
}

// RESAMPLED = false is the hot instantiation (no up/down sampling): keeping it in a kernel of its own gives
// it its own register allocation (the 2x2-mean variant prefetches 4 source pixels per tile element).
template <class C, bool RESAMPLED>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs p, int tiles_x, int tiles_y, int mtiles,
                                                           int nchunks, int coutp) {
  __shared__ __attribute__((aligned(16))) float xl[C::XL];
  __shared__ __attribute__((aligned(16))) float wl[C::WL];
  if (!RESAMPLED) conv_body<C, RS_NONE>(p, xl, wl, tiles_x, tiles_y, mtiles, nchunks, coutp);
  else if (p.resample == RS_UP) conv_body<C, RS_UP>(p, xl, wl, tiles_x, tiles_y, mtiles, nchunks, coutp);
  else conv_body<C, RS_DOWN>(p, xl, wl, tiles_x, tiles_y, mtiles, nchunks, coutp);
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
static int launch_cfg(const ConvArgs& a_in, hipStream_t stream) {
  const ConvArgs& a = a_in;
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
    snprintf(name, sizeof(name), "conv_mfma_kernel<ConvCfg<%d, %d, %d, %d, %d, %d, %d>, %s>", C::MT, C::PH, C::PW,
             C::WM, C::WN, C::TAPS, C::KC, a.resample == RS_NONE ? "false" : "true");
  const double px = (double)a.B * a.H * a.W;
  const double flops = 2.0 * px * a.Cout * (double)(a.Ca + a.Cb) * C::TAPS;
  const double bytes = 4.0 * ((double)a.B * (a.Ca + a.Cb) * a.Hs * a.Ws + px * a.Cout * (a.res ? 2 : 1) +
                              (double)a.Cout * (a.Ca + a.Cb) * C::TAPS);
  ProfScope ps(name, flops, bytes, stream);
  if (a.resample == RS_NONE)
    hipLaunchKernelGGL((conv_mfma_kernel<C, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a, tiles_x, tiles_y,
                       mtiles, nchunks, cout_padded(a.Cout));
  else
    hipLaunchKernelGGL((conv_mfma_kernel<C, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a, tiles_x, tiles_y,
                       mtiles, nchunks, cout_padded(a.Cout));
  MCEDM_LAUNCH_CHECK("conv_mfma_kernel");
  if (a.gsum_tiles) *a.gsum_tiles = tiles_x * tiles_y;
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
  // The PIXEL tile is a function of the image size only (so that the fused GroupNorm partial sums, and with them
  // every output bit, do not depend on the batch size: exact batch shardability); the CHANNEL tile is the largest
  // one that still gives the 256 CUs about two workgroups each.
  auto blocks_for = [&](int mt, int ph, int pw) {
    return (long long)a.B * ceil_div(a.H, ph) * ceil_div(a.W, pw) * ceil_div(a.Cout, mt);
  };
  const long long want = 512;
  if ((long long)a.H * a.W <= 256 || a.W < 12) {          // <= 16x16 images: 8x8-pixel tiles
    if (coutp % 64 == 0) return launch_cfg<ConvCfg<64, 8, 8, 2, 2, TAPS, KC>>(a, stream);
    return launch_cfg<ConvCfg<32, 8, 8, 1, 2, TAPS, KC>>(a, stream);
  }
  if (a.W >= 24) {
    if (coutp % 128 == 0 && blocks_for(128, 8, 32) >= want) return launch_cfg<ConvCfg<128, 8, 32, 1, 4, TAPS, KC>>(a, stream);
    if (coutp % 64 == 0 && blocks_for(64, 8, 32) >= want) return launch_cfg<ConvCfg<64, 8, 32, 1, 4, TAPS, KC>>(a, stream);
    return launch_cfg<ConvCfg<32, 8, 32, 1, 4, TAPS, KC>>(a, stream);
  }
  if (coutp % 128 == 0 && blocks_for(128, 16, 16) >= want) return launch_cfg<ConvCfg<128, 16, 16, 1, 4, TAPS, KC>>(a, stream);
  if (coutp % 64 == 0 && blocks_for(64, 16, 16) >= want) return launch_cfg<ConvCfg<64, 16, 16, 1, 4, TAPS, KC>>(a, stream);
  return launch_cfg<ConvCfg<32, 16, 16, 1, 4, TAPS, KC>>(a, stream);
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
