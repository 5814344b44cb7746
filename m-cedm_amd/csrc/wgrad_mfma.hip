// wgrad_mfma.hip -- K10a: weight (and bias) gradient of the fused 3x3 / 1x1 convolution on fp32 MFMA.
//
//   dW[co][ci][tap] = sum_{n, pixel} dY[n][co][pixel] * X'[n][ci][pixel + tap]
//   X' = resample(act(coef(cat(xa, xb))))   -- the SAME fused input transform as the forward (conv_mfma.hip),
//                                              (the forward never stores X'; the backward rebuilds it into a scratch)
//
// Two steps: (1) X' is materialised once by an elementwise kernel at HBM speed; (2) the GEMM kernel stages plain rows.
// GEMM view per tap: D[co][ci] += A[co][pixel] * B[pixel][ci]; the contraction runs over pixels (2 per MFMA).
// A workgroup owns a 64 (co) x 64 (ci) block of dW for all taps (4 waves x one 32x32 tile x TAPS accumulators)
// and walks a contiguous run of the 64-pixel tiles of the batch (split-K).  NO ATOMICS: every split stores its partial
// block into its own slice of a [split][tap][CoP][CiP] fp32 scratch (ci contiguous: two 128-B segments per
// wave-instruction) and wgrad_reduce_kernel adds the slices in split order in fp64 and writes the reference
// [co][ci][kh][kw] layout, so the gradients (and with them a training run) are bitwise reproducible.  The block with
// ci-tile 0 also sums its dY tiles over pixels: that is the bias gradient, reduced the same way.
#include <cstdlib>

#include "common.hpp"
#include "prof.hpp"
#include "bwd.hpp"

namespace mcedm {

template <int PH_, int PW_, int TAPS_>
struct WgCfg {
  static constexpr int PH = PH_, PW = PW_, TAPS = TAPS_;
  static constexpr int CT = 64, IT = 64;
  static constexpr int HALO = (TAPS == 9) ? 1 : 0;
  static constexpr int PITCH = PW + 2 * HALO;
  static constexpr int ROWS = PH + 2 * HALO;
  static constexpr int PLANE = ROWS * PITCH;
  static constexpr int AP = (PLANE % 2 == 0) ? PLANE + 1 : PLANE;   // odd plane pitch: conflict-free B reads
  static constexpr int NPIX = PH * PW;
  static constexpr int DP = NPIX + 1;                                // odd row pitch: conflict-free A reads
  static constexpr int SUB = (PLANE + 255) / 256;
  static_assert(NPIX == 64, "64-pixel tiles");
};

bool wgrad_thin_applicable(const WgradArgs& a, int taps, int qkv_heads);
int launch_wgrad_thin(const WgradArgs& a, float* dw, float* db, hipStream_t s);
static bool wgrad_thin_enabled() {                       // MCEDM_WGRAD_THIN=0: the MFMA block kernel for these shapes too (A/B runs)
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_WGRAD_THIN"); env = e ? atoi(e) : 1; }
  return env != 0;
}

__device__ __forceinline__ float silu_w(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// ---- step 1: materialise the conv input  X' = resample(act(coef(cat(xa, xb))))  once, at HBM speed --------------
// (the forward never stores it; recomputing it inside the wgrad staging costs ~45 VALU ops per staged element and made
// that kernel VALU-bound.)  One thread per 4 consecutive output pixels of a row when W % 4 == 0.
// grid = (planes, chunks of a plane): blockIdx.x = (sample, channel) -- one transform row and one source plane per workgroup, 32-bit
// index arithmetic per quad (round 4's flat 64-bit index took four 64-bit divisions per quad: 3.7 TB/s read + write).
__global__ __launch_bounds__(256) void act_materialize_kernel(WgradArgs p, float* __restrict__ out, int vec) {
  const int Cin = p.Ca + p.Cb;
  const unsigned HW = (unsigned)p.H * p.W, HWs = (unsigned)p.Hs * p.Ws;
  const int n = blockIdx.x / Cin, ci = blockIdx.x - n * Cin;
  const bool in_a = ci < p.Ca;
  const float* src = in_a ? p.xa : p.xb;
  float* dst = out + (size_t)blockIdx.x * HW;
  const unsigned nq = HW / vec;
  if (!src) {
    for (unsigned q = blockIdx.y * 256u + threadIdx.x; q < nq; q += gridDim.y * 256u) {
      if (vec == 4) *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{0.f, 0.f, 0.f, 0.f};
      else dst[q] = 0.f;
    }
    return;
  }
  const float* plane = src + ((size_t)n * (in_a ? p.Ca : p.Cb) + (in_a ? ci : ci - p.Ca)) * HWs;
  Coef cf{0.f, 1.f, 0.f, 0.f};
  if (p.coef) cf = p.coef[(p.coef_batch ? (size_t)n * Cin : 0) + ci];
  const unsigned wq = (unsigned)p.W / vec;                       // quads per output row
  for (unsigned q = blockIdx.y * 256u + threadIdx.x; q < nq; q += gridDim.y * 256u) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.resample == RS_NONE && vec == 4) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(plane + 4 * q);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float tt = (s4[k] - cf.mean) * cf.scale + cf.offset;
        v[k] = p.act ? silu_w(tt) : tt;
      }
    } else {
      const unsigned y = q / wq, x0 = (q - y * wq) * vec;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < vec) {
          const unsigned x = x0 + k;
          if (p.resample == RS_NONE) {
            const float tt = (plane[y * p.Ws + x] - cf.mean) * cf.scale + cf.offset;
            v[k] = p.act ? silu_w(tt) : tt;
          } else if (p.resample == RS_UP) {
            const float tt = (plane[(y >> 1) * p.Ws + (x >> 1)] - cf.mean) * cf.scale + cf.offset;
            v[k] = p.act ? silu_w(tt) : tt;
          } else {
            const float* q0 = plane + (2 * y) * p.Ws + 2 * x;
            float s4[4] = {q0[0], q0[1], q0[p.Ws], q0[p.Ws + 1]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float tt = (s4[j] - cf.mean) * cf.scale + cf.offset;
              s4[j] = p.act ? silu_w(tt) : tt;
            }
            v[k] = 0.25f * ((s4[0] + s4[1]) + (s4[2] + s4[3]));
          }
        }
      }
    }
    if (vec == 4) *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{v[0], v[1], v[2], v[3]};
    else dst[q] = v[0];
  }
}

// ---- step 2: dW tile = dY tile x X' tile over 64-pixel tiles; staging is plain row copies -------------------------
// FAST (W % 4 == 0, 16-byte aligned tensors): both tiles are fetched with UNCONDITIONAL float4 loads from clamped
// addresses into registers while the previous tile's MFMAs run, and masked when they are written to LDS (a guarded
// load compiles to load + s_waitcnt vmcnt(0): one exposed memory round trip per row, 100 -> 1xx TFLOP/s).
template <class C, bool FAST>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs p, const float* __restrict__ xact, int tiles_x,
                                                       int tiles_y, int ctiles, int itiles, int nsplit, int ntiles,
                                                       int cop, int cip) {
  __shared__ float dyl[C::CT * C::DP];
  __shared__ float al[C::IT * C::AP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int mi = wave >> 1, ni = wave & 1;
  int bid = blockIdx.x;
  const int split = bid % nsplit; bid /= nsplit;
  const int it = bid % itiles;
  const int ct = bid / itiles;
  const int co0 = ct * C::CT, ci0 = it * C::IT;
  const int Cin = p.Ca + p.Cb;
  const int per = (ntiles + nsplit - 1) / nsplit;
  const int t_begin = split * per;
  const int t_end = (t_begin + per < ntiles) ? t_begin + per : ntiles;

  f32x16 acc[C::TAPS];
#pragma unroll
  for (int t = 0; t < C::TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  if (t_begin >= t_end) return;      // whole workgroup (depends on blockIdx only)

  const size_t HW = (size_t)p.H * p.W;
  // xact == nullptr: the operand is the un-transformed cat(xa, xb) itself, read in place (the decoder's skip projections: no 537 MB copy)
  auto plane_of = [&](int n, int ci) -> const float* {
    if (xact) return xact + ((size_t)n * Cin + ci) * HW;
    return ci < p.Ca ? p.xa + ((size_t)n * p.Ca + ci) * HW : p.xb + ((size_t)n * p.Cb + (ci - p.Ca)) * HW;
  };
  const bool vec_ok = (p.W % 4 == 0) && ((reinterpret_cast<size_t>(p.dy) & 15) == 0) &&
                      (((xact ? reinterpret_cast<size_t>(xact) : (reinterpret_cast<size_t>(p.xa) | reinterpret_cast<size_t>(p.xb))) & 15) == 0);
  const int tiles_img = tiles_x * tiles_y;
  constexpr int XROWS = C::IT * C::ROWS;          // rows of the input tile (channel x tile row)
  constexpr int DROWS4 = C::CT * C::NPIX / 4;     // float4 groups of the dY tile

  constexpr int NDY = DROWS4 / 256;                     // float4 of the dY tile per thread
  constexpr int NXR = (XROWS + 255) / 256;              // rows of the X' tile per thread
  f32x4 rdy[NDY], rx[NXR][C::PW / 4];
  float rh[NXR][2];
  unsigned ok_dy = 0, ok_x = 0;                         // validity bits, applied at commit
  auto fetch = [&](int t) {
    const int n = t / tiles_img;
    const int q = t - n * tiles_img;
    const int y0 = (q / tiles_x) * C::PH, x0 = (q % tiles_x) * C::PW;
    ok_dy = 0; ok_x = 0;
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int i4 = tid + 256 * j;
      const int cl = i4 / (C::NPIX / 4), pp = (i4 % (C::NPIX / 4)) * 4;
      const int y = y0 + pp / C::PW, x = x0 + pp % C::PW;
      const int co = co0 + cl;
      const bool ok = co < p.Cout && y < p.H && x < p.W;
      ok_dy |= (ok ? 1u : 0u) << j;
      const float* src = p.dy + ((size_t)n * p.Cout + (co < p.Cout ? co : p.Cout - 1)) * HW +
                         (size_t)(y < p.H ? y : p.H - 1) * p.W + (x < p.W ? x : p.W - 4);
      rdy[j] = *reinterpret_cast<const f32x4*>(src);
    }
#pragma unroll
    for (int k = 0; k < NXR; ++k) {
      int rr = tid + 256 * k;
      if (rr > XROWS - 1) rr = XROWS - 1;               // clamped duplicate; not committed
      const int cil = rr / C::ROWS, r = rr - cil * C::ROWS;
      const int ci = ci0 + cil;
      const int y = y0 + r - C::HALO;
      const bool row_ok = ci < Cin && (unsigned)y < (unsigned)p.H;
      const float* src = plane_of(n, ci < Cin ? ci : Cin - 1) +
                         (size_t)(y < 0 ? 0 : (y < p.H ? y : p.H - 1)) * p.W;
      unsigned bits = 0;
      if (C::HALO) {
        bits |= (row_ok && x0 > 0) ? 1u : 0u;
        bits |= (row_ok && x0 + C::PW < p.W) ? 2u : 0u;
        rh[k][0] = src[x0 > 0 ? x0 - 1 : 0];
        rh[k][1] = src[x0 + C::PW < p.W ? x0 + C::PW : p.W - 1];
      }
#pragma unroll
      for (int c4 = 0; c4 < C::PW / 4; ++c4) {
        const int x = x0 + 4 * c4;
        bits |= (row_ok && x < p.W) ? (4u << c4) : 0u;
        rx[k][c4] = *reinterpret_cast<const f32x4*>(src + (x < p.W ? x : p.W - 4));
      }
      ok_x |= bits << (k * 10);                          // PW / 4 + 2 <= 10 bits per row (PW <= 32)
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int i4 = tid + 256 * j;
      const int cl = i4 / (C::NPIX / 4), pp = (i4 % (C::NPIX / 4)) * 4;
      const bool ok = (ok_dy >> j) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) dyl[cl * C::DP + pp + e] = ok ? rdy[j][e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NXR; ++k) {
      const int rr = tid + 256 * k;
      if (rr < XROWS) {
        const int cil = rr / C::ROWS, r = rr - cil * C::ROWS;
        float* dst = al + cil * C::AP + r * C::PITCH;
        const unsigned bits = ok_x >> (k * 10);
        if (C::HALO) {
          dst[0] = (bits & 1u) ? rh[k][0] : 0.f;
          dst[C::PITCH - 1] = (bits & 2u) ? rh[k][1] : 0.f;
        }
#pragma unroll
        for (int c4 = 0; c4 < C::PW / 4; ++c4) {
          const bool ok = bits & (4u << c4);
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[C::HALO + 4 * c4 + e] = ok ? rx[k][c4][e] : 0.f;
        }
      }
    }
  };
  static_assert(!FAST || (NDY <= 16 && NXR <= 3 && C::PW <= 32), "validity bit budget");
  if (FAST) fetch(t_begin);

  for (int t = t_begin; t < t_end; ++t) {
    const int n = t / tiles_img;
    const int q = t - n * tiles_img;
    const int y0 = (q / tiles_x) * C::PH, x0 = (q % tiles_x) * C::PW;
    if (FAST) {
      commit();
      __syncthreads();
      fetch(t + 1 < t_end ? t + 1 : t);                 // unconditional (the last one is dropped): no phi copies
    } else {
    // ---- dY tile [CT][NPIX]: float4 per thread-iteration
#pragma unroll
    for (int j = 0; j < DROWS4 / 256; ++j) {
      const int i4 = tid + 256 * j;
      const int cl = i4 / (C::NPIX / 4), pp = (i4 % (C::NPIX / 4)) * 4;
      const int y = y0 + pp / C::PW, x = x0 + pp % C::PW;
      const int co = co0 + cl;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (co < p.Cout && y < p.H && x < p.W) {
        const float* src = p.dy + ((size_t)n * p.Cout + co) * HW + (size_t)y * p.W + x;
        if (vec_ok) v = *reinterpret_cast<const f32x4*>(src);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (x + e < p.W) ? src[e] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) dyl[cl * C::DP + pp + e] = v[e];
    }
    // ---- X' tile [IT][ROWS][PITCH]: one (channel, row) per thread-iteration; interior as float4, halo as scalars
    for (int rr = tid; rr < XROWS; rr += 256) {
      const int cil = rr / C::ROWS, r = rr - cil * C::ROWS;
      const int ci = ci0 + cil;
      const int y = y0 + r - C::HALO;
      float* dst = al + cil * C::AP + r * C::PITCH;
      const bool row_ok = ci < Cin && (unsigned)y < (unsigned)p.H;
      const float* src = plane_of(n, row_ok ? ci : 0) + (size_t)(row_ok ? y : 0) * p.W;
      if (C::HALO) {
        dst[0] = (row_ok && x0 > 0) ? src[x0 - 1] : 0.f;
        dst[C::PITCH - 1] = (row_ok && x0 + C::PW < p.W) ? src[x0 + C::PW] : 0.f;
      }
#pragma unroll
      for (int c4 = 0; c4 < C::PW / 4; ++c4) {
        const int x = x0 + 4 * c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row_ok && x < p.W) {
          if (vec_ok) v = *reinterpret_cast<const f32x4*>(src + x);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (x + e < p.W) ? src[x + e] : 0.f;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[C::HALO + 4 * c4 + e] = v[e];
      }
    }
    __syncthreads();
    }
    if (it == 0 && tid < C::CT) {     // bias gradient: row sums of the dY tile
      float s = 0.f;
#pragma unroll 8
      for (int pp = 0; pp < C::NPIX; ++pp) s += dyl[tid * C::DP + pp];
      bsum += s;
    }
    // ---- contraction over the tile's 64 pixels (32 MFMA steps of 2 pixels) for every tap; operand fragments
    // are double-buffered in registers and the LDS-read / MFMA order is pinned (reads of step s+1, then the MFMAs of s)
    const float* arow = dyl + (mi * 32 + l31) * C::DP + h;
    const float* brow = al + (ni * 32 + l31) * C::AP + h;
    float fa[2], fb[2][C::TAPS];
    fa[0] = arow[0];
#pragma unroll
    for (int tap = 0; tap < C::TAPS; ++tap) fb[0][tap] = brow[(C::TAPS == 9) ? (tap / 3) * C::PITCH + (tap % 3) : 0];
#pragma unroll
    for (int s2 = 0; s2 < C::NPIX / 2; ++s2) {
      const int cur = s2 & 1, nxt = cur ^ 1;
      const int pn = (s2 + 1 < C::NPIX / 2) ? 2 * (s2 + 1) : 2 * s2;      // clamped: the last prefetch is discarded
      const int boff = (pn / C::PW) * C::PITCH + (pn % C::PW);
      fa[nxt] = arow[pn];
#pragma unroll
      for (int tap = 0; tap < C::TAPS; ++tap)
        fb[nxt][tap] = brow[boff + ((C::TAPS == 9) ? (tap / 3) * C::PITCH + (tap % 3) : 0)];
#pragma unroll
      for (int tap = 0; tap < C::TAPS; ++tap)
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur], fb[cur][tap], acc[tap], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, C::TAPS + 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, C::TAPS, 0);
    }
    __syncthreads();
  }
  // ---- this split's partial block -> its own slice of the [split][tap][CoP][CiP] scratch (plain stores)
  const int ci = ci0 + ni * 32 + l31;
  float* slice = p.dwp + (size_t)split * C::TAPS * cop * cip;
#pragma unroll
  for (int tap = 0; tap < C::TAPS; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < p.Cout && ci < Cin) slice[((size_t)tap * cop + co) * cip + ci] = acc[tap][r];
    }
  if (it == 0 && tid < C::CT && co0 + tid < p.Cout && p.dbp) p.dbp[(size_t)split * cop + co0 + tid] = bsum;
}

// Fixed-order reduction of the split slices and conversion to the reference layout:
//   dW[co][ci][tap] = sum_s scratch[s][tap][perm(co)][ci]   (fp64);  db[co] = sum_s scratch_b[s][perm(co)].
// One workgroup per 64 consecutive scratch elements (ci fastest: a wave-instruction reads one 256-byte segment of a slice); its
// four waves take the slices s = w, w + 4, w + 8, ... -- four interleaved chains each, combined in a fixed order --, the four
// partial sums meet in LDS and wave 0 adds them in wave order.  (Round 4: one thread per element with the whole chain: 145
// workgroups and 0.56 TB/s on a 64 x 64-channel conv's 512 slices -- 3.5 ms of the reference network's 25.5 ms training step.)
// qkv_heads > 0: scratch rows are in packed (head, {q,k,v}, c) order, the parameter in (head, c, {q,k,v}) order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ dwp, const float* __restrict__ dbp,
                                                           float* __restrict__ dw, float* __restrict__ db, int Cout, int Cin, int taps,
                                                           int cop, int cip, int nact, int qkv_heads) {
  __shared__ double part[4][64];
  const size_t block = (size_t)taps * cop * cip;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + lane;
  const bool live = i < block + cop;
  const bool is_b = i >= block;
  const float* src = is_b ? dbp + (i - block) : dwp + i;
  const size_t stride = is_b ? (size_t)cop : block;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (live && (!is_b || dbp)) {
    int sp = wave;
    for (; sp + 12 < nact; sp += 16) {
      const float v0 = src[(size_t)sp * stride], v1 = src[(size_t)(sp + 4) * stride];
      const float v2 = src[(size_t)(sp + 8) * stride], v3 = src[(size_t)(sp + 12) * stride];
      s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
    }
    for (; sp < nact; sp += 4) s0 += (double)src[(size_t)sp * stride];
  }
  part[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave != 0 || !live) return;
  const double sum = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  const int cp = is_b ? (int)(i - block) : (int)((i / cip) % cop);       // packed row
  if (cp >= Cout) return;
  int co = cp;                                                            // reference row held by packed row cp
  if (qkv_heads > 0) {
    const int per = Cout / qkv_heads, d = per / 3;
    const int hh = cp / per, rr = cp % per, which = rr / d, c = rr % d;
    co = hh * per + c * 3 + which;
  }
  if (is_b) {
    if (db) db[co] = (float)sum;
  } else {
    const int ci = (int)(i % cip), tap = (int)(i / ((size_t)cop * cip));
    if (ci < Cin) dw[((size_t)co * Cin + ci) * taps + tap] = (float)sum;
  }
}

// split slices: at most 512 workgroups per launch, each with its own [taps][64][64] block (+ 64 bias partials)
static int wgrad_nsplit_max(int Cout, int Cin) {
  const int blocks = ceil_div(Cout, 64) * ceil_div(Cin, 64);
  return blocks >= 512 ? 1 : 512 / blocks;
}
size_t wgrad_scratch_floats(int Cout, int Cin, int taps) {
  const size_t cop = (Cout + 63) / 64 * 64, cip = (Cin + 63) / 64 * 64;
  const size_t direct = (size_t)wgrad_nsplit_max(Cout, Cin) * (taps * cop * cip + cop);
  const size_t wino = wgrad_wino_scratch_floats(Cout, Cin, taps);          // wgrad_wino.hip: the Winograd-domain partial blocks
  const size_t gemm1 = wgrad_gemm1_scratch_floats(Cout, Cin, taps);        // ... and the 1x1 GEMM form's slices
  const size_t m = direct > wino ? direct : wino;
  return m > gemm1 ? m : gemm1;
}

template <class C>
static int launch_wg(const WgradArgs& a, const float* xact, int* nact, hipStream_t s) {
  const int tiles_x = ceil_div(a.W, C::PW), tiles_y = ceil_div(a.H, C::PH);
  const int ctiles = ceil_div(a.Cout, C::CT), itiles = ceil_div(a.Ca + a.Cb, C::IT);
  const int ntiles = a.B * tiles_x * tiles_y;
  // one full round of resident workgroups (256 CUs x 2): a 1.5-round grid wastes a quarter of the machine.
  // The split factor depends on the conv's shape and the tile count only: same shapes -> same bits.
  int nsplit = wgrad_nsplit_max(a.Cout, a.Ca + a.Cb);
  // at least two 64-pixel tiles per split: a split's partial block ([taps][64][64] floats: 147 KB for a 3x3 conv) costs a store
  // here and a read in the reduction -- with one tile per split (3x3 convs at 16^2, B = 32: 128 splits) the kernel wrote and the
  // reduction re-read 75 MB for 2.4 GFLOP of matrix work
  if (nsplit > ntiles / 2) nsplit = ntiles / 2;
  if (nsplit < 1) nsplit = 1;
  const int cop = ctiles * 64, cip = itiles * 64;
  if (nact) *nact = ceil_div(ntiles, ceil_div(ntiles, nsplit));     // splits that own at least one tile (the others do not store)
  char name[64] = "";
  if (prof_enabled()) snprintf(name, sizeof(name), "wgrad_kernel<WgCfg<%d, %d, %d>>", C::PH, C::PW, C::TAPS);
  const double flops = 2.0 * a.B * a.H * (double)a.W * a.Cout * (a.Ca + a.Cb) * C::TAPS;
  ProfScope ps(name, flops, 4.0 * a.B * ((double)a.Cout * a.H * a.W + (double)(a.Ca + a.Cb) * a.H * a.W), s);
  const size_t xbits = xact ? reinterpret_cast<size_t>(xact) : (reinterpret_cast<size_t>(a.xa) | reinterpret_cast<size_t>(a.xb));
  const bool fast = (a.W % 4 == 0) && ((reinterpret_cast<size_t>(a.dy) & 15) == 0) && ((xbits & 15) == 0);
  if (fast)
    hipLaunchKernelGGL((wgrad_kernel<C, true>), dim3(ctiles * itiles * nsplit), dim3(256), 0, s, a, xact, tiles_x, tiles_y,
                       ctiles, itiles, nsplit, ntiles, cop, cip);
  else
    hipLaunchKernelGGL((wgrad_kernel<C, false>), dim3(ctiles * itiles * nsplit), dim3(256), 0, s, a, xact, tiles_x, tiles_y,
                       ctiles, itiles, nsplit, ntiles, cop, cip);
  MCEDM_LAUNCH_CHECK("wgrad_kernel");
  return MCEDM_OK;
}

// out: B * Cin * H * W floats = resample(act(coef(cat(xa, xb)))) at the conv resolution
int launch_act_materialize(const WgradArgs& a, float* out, hipStream_t s) {
  const int Cin = a.Ca + a.Cb;
  const size_t total = (size_t)a.B * Cin * a.H * a.W;
  const bool src16 = ((reinterpret_cast<size_t>(a.xa) | reinterpret_cast<size_t>(a.xb)) & 15) == 0 && ((size_t)a.Hs * a.Ws) % 4 == 0;
  const int vec = (a.W % 4 == 0 && (reinterpret_cast<size_t>(out) & 15) == 0 && ((size_t)a.H * a.W) % 4 == 0 &&
                   (a.resample != RS_NONE || src16)) ? 4 : 1;
  const unsigned nq = (unsigned)((size_t)a.H * a.W / vec);
  MCEDM_REQUIRE((size_t)a.B * Cin < (1ull << 31) && (size_t)a.H * a.W < (1ull << 31), "act_materialize: shape out of range");
  int bx = (int)((nq + 1023) / 1024);                         // four quads per thread and trip
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  ProfScope ps("act_materialize_kernel", 10.0 * total, 4.0 * ((double)a.B * Cin * a.Hs * a.Ws + (double)total), s);
  hipLaunchKernelGGL(act_materialize_kernel, dim3(a.B * Cin, bx), dim3(256), 0, s, a, out, vec);
  MCEDM_LAUNCH_CHECK("act_materialize_kernel");
  return MCEDM_OK;
}

// act_tmp: B * Cin * H * W floats (the materialised conv input)
int launch_wgrad(const WgradArgs& a, int taps, float* dw, float* db, int qkv_heads, float* act_tmp, hipStream_t s, bool have_act) {
  MCEDM_REQUIRE(taps == 9 || taps == 1, "wgrad: taps must be 9 or 1");
  MCEDM_REQUIRE(a.dy && a.dwp && dw && act_tmp, "wgrad: null pointer");
  const int Cin = a.Ca + a.Cb;
  const size_t cop = (a.Cout + 63) / 64 * 64, cip = (Cin + 63) / 64 * 64;
  WgradArgs b = a;
  b.dbp = a.dwp + (size_t)wgrad_nsplit_max(a.Cout, Cin) * taps * cop * cip;      // [split][CoP] bias partials behind the slices
  int rc, nact = 1;
  // an un-transformed single-source input (the attention projection's) IS the operand: no copy -- except for the Winograd
  // kernel, which reads the 4 bytes in front of its operand (and discards them): it only ever sees act_tmp, a scratch region
  // that is never the start of an allocation
  if (!have_act && wgrad_thin_applicable(a, taps, qkv_heads) && wgrad_thin_enabled()) return launch_wgrad_thin(a, dw, db, s);
  const bool wino = wgrad_wino_applicable(a, taps, qkv_heads);
  const bool plain = !wino && !have_act && !a.coef && !a.act && a.resample == RS_NONE && a.xa && (a.Cb == 0 || a.xb);
  const float* xact = plain ? (a.Cb == 0 ? a.xa : nullptr) : act_tmp;      // nullptr: the kernel reads cat(xa, xb) in place
  if (!have_act && !plain && (rc = launch_act_materialize(a, act_tmp, s))) return rc;
  if (wino) return launch_wgrad_wino(a, xact, dw, db, s);
  if (wgrad_gemm1_applicable(a, taps, xact, plain && a.Cb > 0)) {         // 1x1, 128-channel blocks: the GEMM on the Winograd kernel's stage machinery
    const size_t per_slice = (size_t)a.Cout * Cin + a.Cout;
    const size_t slices_max = wgrad_gemm1_scratch_floats(a.Cout, Cin, taps) / per_slice;
    float* dbp = a.dwp + slices_max * (size_t)a.Cout * Cin;
    int nslices = 0;
    if ((rc = launch_wgrad_gemm1(a, xact, dbp, &nslices, s))) return rc;
    const size_t total = (size_t)a.Cout * Cin + a.Cout;
    ProfScope ps("wgrad_reduce_kernel", (double)nslices * total, 4.0 * ((double)nslices + 1.0) * total, s);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, a.dwp, dbp, dw, db, a.Cout, Cin, 1, a.Cout, Cin,
                       nslices, qkv_heads);
    MCEDM_LAUNCH_CHECK("wgrad_reduce_kernel");
    return MCEDM_OK;
  }
  if (taps == 9) {
    if (a.W >= 24) rc = launch_wg<WgCfg<2, 32, 9>>(b, xact, &nact, s);
    else if (a.W >= 12) rc = launch_wg<WgCfg<4, 16, 9>>(b, xact, &nact, s);
    else rc = launch_wg<WgCfg<8, 8, 9>>(b, xact, &nact, s);
  } else {
    if (a.W >= 24) rc = launch_wg<WgCfg<2, 32, 1>>(b, xact, &nact, s);
    else if (a.W >= 12) rc = launch_wg<WgCfg<4, 16, 1>>(b, xact, &nact, s);
    else rc = launch_wg<WgCfg<8, 8, 1>>(b, xact, &nact, s);
  }
  if (rc) return rc;
  const size_t total = (size_t)taps * cop * cip + cop;
  ProfScope ps("wgrad_reduce_kernel", (double)nact * total, 4.0 * ((double)nact + 1.0) * total, s);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, a.dwp, b.dbp, dw, db, a.Cout, Cin, taps, (int)cop,
                     (int)cip, nact, qkv_heads);
  MCEDM_LAUNCH_CHECK("wgrad_reduce_kernel");
  return MCEDM_OK;
}

// =====================================================================================================================
// Thin shapes: the 3x3 weight gradient when ONE side has at most 4 channels -- conv_in (cat(cond, x): 4 input channels) and
// out_conv (2 output channels), adm_blocks.py:383-385, 403.  The 64 x 64-channel MFMA block of wgrad_kernel is 3-6 % full there
// (0.6 ms each at 128^2, B = 32: 5.9 TFLOP/s); the work is one pass over the wide tensor (268 MB) and 9 * S multiply-adds per
// element, i.e. HBM-bound.  One wave per (sample, wide channel, 64-column block), lanes along x, a sliding 3 x 3 window of the
// thin tensor's S channels in registers (3 S loads per row, L1 / L2 hits: every wave of the sample reads the same thin planes):
//
//   big = X' (S = Cout, out_conv):  R[ci][co][k] = sum X'[ci][y][x] * dY[co][y - 1 + k / 3][x - 1 + k % 3]  ->  dW[co][ci][8 - k]
//   big = dY (S = Cin,  conv_in):   R[co][ci][k] = sum dY[co][y][x] * X'[ci][y - 1 + k / 3][x - 1 + k % 3]  ->  dW[co][ci][k]
//
// X' = act(coef(cat(xa, xb))) is applied on the fly (no materialisation pass: the out_conv's 268 MB copy disappears as well); the
// zero padding is applied AFTER the transform.  Per-wave partial sums (fixed-order butterfly) go to [sample][block][wide][10 S];
// wgrad_thin_reduce_kernel adds them in that order in fp64: no atomics, bitwise reproducible.  The bias gradient rides along.
template <int S, bool BIG_IS_X>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(WgradArgs p, float* __restrict__ part, int ncol, int CB, int nwaves) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= nwaves) return;                                   // whole waves
  const int cbk = w % ncol, cb = (w / ncol) % CB, n = w / (ncol * CB);
  const int H = p.H, W = p.W, Cin = p.Ca + p.Cb;
  const size_t HW = (size_t)H * W;
  const int x = cbk * 64 + lane;
  const bool xin = x < W;
  const int xc = xin ? x : W - 1;
  const int xl = xc > 0 ? xc - 1 : 0, xr = xc + 1 < W ? xc + 1 : W - 1;
  const bool okl = xin && x > 0, okr = xin && x + 1 < W;
  // wide plane (and its transform row when it is X')
  const float* bigp;
  Coef bcf{0.f, 1.f, 0.f, 0.f};
  if (BIG_IS_X) {
    const bool in_a = cb < p.Ca;
    bigp = (in_a ? p.xa + ((size_t)n * p.Ca + cb) * HW : p.xb + ((size_t)n * p.Cb + (cb - p.Ca)) * HW);
    if (p.coef) bcf = p.coef[(p.coef_batch ? (size_t)n * Cin : 0) + cb];
  } else {
    bigp = p.dy + ((size_t)n * p.Cout + cb) * HW;
  }
  // thin planes (and their transform rows when they are X')
  const float* sp[S];
  Coef scf[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    scf[s] = Coef{0.f, 1.f, 0.f, 0.f};
    if (BIG_IS_X) {
      sp[s] = p.dy + ((size_t)n * p.Cout + s) * HW;
    } else {
      const bool in_a = s < p.Ca;
      sp[s] = in_a ? p.xa + ((size_t)n * p.Ca + s) * HW : p.xb + ((size_t)n * p.Cb + (s - p.Ca)) * HW;
      if (p.coef) scf[s] = p.coef[(p.coef_batch ? (size_t)n * Cin : 0) + s];
    }
  }
  auto xform = [&](float v, const Coef& cf) {
    const float t = (v - cf.mean) * cf.scale + cf.offset;
    return p.act ? silu_w(t) : t;
  };
  // row r of the thin tensor at columns x - 1, x, x + 1 (zeros outside the image; clamped addresses, values selected)
  auto load_row = [&](int r, float (&dst)[S][3]) {
    const bool rok = (unsigned)r < (unsigned)H;
    const size_t ro = (size_t)(rok ? r : 0) * W;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float a = sp[s][ro + xl], b = sp[s][ro + xc], c = sp[s][ro + xr];
      if (!BIG_IS_X) { a = xform(a, scf[s]); b = xform(b, scf[s]); c = xform(c, scf[s]); }
      dst[s][0] = (rok && okl) ? a : 0.f;
      dst[s][1] = (rok && xin) ? b : 0.f;
      dst[s][2] = (rok && okr) ? c : 0.f;
    }
  };
  float acc[S][9];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[s][k] = 0.f;
  float bsum[S];                                              // bias gradient: sum of dY (the wide operand: slot 0; the thin one: the
#pragma unroll                                                 // windows' centres, counted once, by the waves of wide channel 0)
  for (int s = 0; s < S; ++s) bsum[s] = 0.f;
  float win[3][S][3];                                         // rows y - 1, y, y + 1
  load_row(-1, win[0]);
  load_row(0, win[1]);
  for (int y = 0; y < H; ++y) {
    load_row(y + 1, win[2]);
    float bv = bigp[(size_t)y * W + xc];
    if (BIG_IS_X) bv = xform(bv, bcf);
    bv = xin ? bv : 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[s][k] = fmaf(bv, win[k / 3][s][k % 3], acc[s][k]);
    if (!BIG_IS_X) bsum[0] += bv;
    else if (cb == 0) {
#pragma unroll
      for (int s = 0; s < S; ++s) bsum[s] += win[1][s][1];
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int c = 0; c < 3; ++c) { win[0][s][c] = win[1][s][c]; win[1][s][c] = win[2][s][c]; }
  }
  // wave sums (fixed-order butterfly) -> this wave's record
  float* rec = part + ((size_t)(n * ncol + cbk) * CB + cb) * (10 * S);
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float v = acc[s][k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) rec[s * 9 + k] = v;
    }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float v = bsum[s];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) rec[9 * S + s] = v;
  }
}

// The same for power-of-two widths 16 .. 256 (every image the networks see): FOUR pixels per lane and 16-byte loads.  The general
// kernel above issues 3 S + 1 dword loads per pixel row and lane and is bound by the NUMBER of vector-memory instructions (a CU
// accepts one every ~25 cycles: 438 us for the out_conv at 128^2, B = 32, whatever the unrolling); here a lane owns the aligned
// quad x0 .. x0 + 3 of a row, a row is W / 4 lanes and a wave holds 64 / (W / 4) row streams that walk disjoint bands of rows:
// S + 1 16-byte loads per 4 pixels, the window's two outer columns come from the neighbouring lanes (ds_bpermute; the row's
// first / last lane sits on the image border: zero).  36 S multiply-adds per lane and trip.
template <int S, bool BIG_IS_X>
__global__ __launch_bounds__(256) void wgrad_thin4_kernel(WgradArgs p, float* __restrict__ part, int CB, int nwaves) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= nwaves) return;                                   // whole waves
  const int cb = w % CB, n = w / CB;
  const int H = p.H, W = p.W, Cin = p.Ca + p.Cb;
  const size_t HW = (size_t)H * W;
  const int nq = W >> 2;                                     // lanes per row (a power of two, 4 .. 64)
  const int xq = lane & (nq - 1), st = lane / nq, streams = 64 / nq;
  const int rows = (H + streams - 1) / streams;              // rows per stream
  const int y0 = st * rows, y1 = y0 + rows < H ? y0 + rows : H;
  const float* bigp;
  Coef bcf{0.f, 1.f, 0.f, 0.f};
  if (BIG_IS_X) {
    const bool in_a = cb < p.Ca;
    bigp = (in_a ? p.xa + ((size_t)n * p.Ca + cb) * HW : p.xb + ((size_t)n * p.Cb + (cb - p.Ca)) * HW);
    if (p.coef) bcf = p.coef[(p.coef_batch ? (size_t)n * Cin : 0) + cb];
  } else {
    bigp = p.dy + ((size_t)n * p.Cout + cb) * HW;
  }
  const float* sp[S];
  Coef scf[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    scf[s] = Coef{0.f, 1.f, 0.f, 0.f};
    if (BIG_IS_X) {
      sp[s] = p.dy + ((size_t)n * p.Cout + s) * HW;
    } else {
      const bool in_a = s < p.Ca;
      sp[s] = in_a ? p.xa + ((size_t)n * p.Ca + s) * HW : p.xb + ((size_t)n * p.Cb + (s - p.Ca)) * HW;
      if (p.coef) scf[s] = p.coef[(p.coef_batch ? (size_t)n * Cin : 0) + s];
    }
  }
  auto xform = [&](float v, const Coef& cf) {
    const float t = (v - cf.mean) * cf.scale + cf.offset;
    return p.act ? silu_w(t) : t;
  };
  // window row r of the thin tensor: columns x0 - 1 .. x0 + 4 of every channel (zeros outside the image, applied after the transform)
  auto load_row = [&](int r, float (&dst)[S][6]) {
    const bool rok = (unsigned)r < (unsigned)H;
    const size_t ro = (size_t)(rok ? r : 0) * W + 4 * xq;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      f32x4 q = *reinterpret_cast<const f32x4*>(sp[s] + ro);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = q[e];
        if (!BIG_IS_X) v = xform(v, scf[s]);
        dst[s][1 + e] = rok ? v : 0.f;
      }
      const float l = __shfl_up(dst[s][4], 1, 64), rr = __shfl_down(dst[s][1], 1, 64);
      dst[s][0] = xq > 0 ? l : 0.f;
      dst[s][5] = xq < nq - 1 ? rr : 0.f;
    }
  };
  float acc[S][9];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[s][k] = 0.f;
  float bsum[S];
#pragma unroll
  for (int s = 0; s < S; ++s) bsum[s] = 0.f;
  float win[3][S][6];
  load_row(y0 - 1, win[0]);
  load_row(y0, win[1]);
  for (int i = 0; i < rows; ++i) {                             // the same trip count in every lane (the shuffles need all of them)
    const int y = y0 + i;
    const bool live = y < y1;
    load_row(y + 1, win[2]);
    const int yc = y < H ? y : H - 1;
    const f32x4 bq = *reinterpret_cast<const f32x4*>(bigp + (size_t)yc * W + 4 * xq);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = bq[j];
      if (BIG_IS_X) v = xform(v, bcf);
      v = live ? v : 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[s][k] = fmaf(v, win[k / 3][s][j + k % 3], acc[s][k]);
      if (!BIG_IS_X) bsum[0] += v;
    }
    if (BIG_IS_X && cb == 0 && live) {
#pragma unroll
      for (int s = 0; s < S; ++s) bsum[s] += (win[1][s][1] + win[1][s][2]) + (win[1][s][3] + win[1][s][4]);
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int c = 0; c < 6; ++c) { win[0][s][c] = win[1][s][c]; win[1][s][c] = win[2][s][c]; }
  }
  float* rec = part + ((size_t)n * CB + cb) * (10 * S);
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float v = acc[s][k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) rec[s * 9 + k] = v;
    }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float v = bsum[s];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) rec[9 * S + s] = v;
  }
}

// dW / db from the records, in (sample, column block) order in fp64.  One thread per (wide channel, thin channel, tap); the
// bias gradient comes from the records' last S slots (see the kernel).
template <bool BIG_IS_X>
__global__ void wgrad_thin_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int S,
                                         int CB, int nrec, int Cin) {
  const int rl = 10 * S;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= CB * S * 9) return;
  const int k = i % 9, s = (i / 9) % S, cb = i / (9 * S);
  double sum = 0.0;
  for (int r = 0; r < nrec; ++r) sum += (double)part[((size_t)r * CB + cb) * rl + s * 9 + k];
  if (BIG_IS_X) dw[((size_t)s * Cin + cb) * 9 + (8 - k)] = (float)sum;       // s = co, cb = ci
  else dw[((size_t)cb * Cin + s) * 9 + k] = (float)sum;                       // cb = co, s = ci
  if (db && k == 0 && (BIG_IS_X ? cb == 0 : s == 0)) {
    double b = 0.0;
    const int slot = BIG_IS_X ? s : 0;
    for (int r = 0; r < nrec; ++r) b += (double)part[((size_t)r * CB + cb) * rl + 9 * S + slot];
    db[BIG_IS_X ? s : cb] = (float)b;
  }
}

bool wgrad_thin_applicable(const WgradArgs& a, int taps, int qkv_heads) {
  const int Cin = a.Ca + a.Cb;
  if (taps != 9 || qkv_heads != 0 || a.resample != RS_NONE || a.B < 1) return false;
  const bool thin_out = a.Cout >= 1 && a.Cout <= 4 && Cin >= 16, thin_in = Cin >= 1 && Cin <= 4 && a.Cout >= 16;
  if (!thin_out && !thin_in) return false;
  if (thin_in && a.xa == nullptr) return false;
  const int CB = thin_out ? Cin : a.Cout, S = thin_out ? a.Cout : Cin;
  const size_t rec = (size_t)a.B * ceil_div(a.W, 64) * CB * (10 * S);
  return rec <= wgrad_scratch_floats(a.Cout, Cin, 9);
}

template <int S, bool BIG_IS_X>
static int launch_thin_cfg(const WgradArgs& a, float* dw, float* db, hipStream_t s) {
  const int Cin = a.Ca + a.Cb;
  const int CB = BIG_IS_X ? Cin : a.Cout;
  const bool pow2 = a.W >= 16 && a.W <= 256 && (a.W & (a.W - 1)) == 0;
  const bool al16 = ((reinterpret_cast<size_t>(a.dy) | reinterpret_cast<size_t>(a.xa) | reinterpret_cast<size_t>(a.xb)) & 15) == 0;
  const bool quad = pow2 && al16;
  const int ncol = quad ? 1 : ceil_div(a.W, 64);
  const int nwaves = a.B * CB * ncol;
  const double px = (double)a.B * a.H * a.W;
  {
    ProfScope ps(quad ? "wgrad_thin4_kernel" : "wgrad_thin_kernel", 2.0 * px * a.Cout * Cin * 9, 4.0 * px * (a.Cout + Cin), s);
    if (quad) hipLaunchKernelGGL((wgrad_thin4_kernel<S, BIG_IS_X>), dim3(ceil_div(nwaves, 4)), dim3(256), 0, s, a, a.dwp, CB, nwaves);
    else hipLaunchKernelGGL((wgrad_thin_kernel<S, BIG_IS_X>), dim3(ceil_div(nwaves, 4)), dim3(256), 0, s, a, a.dwp, ncol, CB, nwaves);
    MCEDM_LAUNCH_CHECK("wgrad_thin_kernel");
  }
  const int nout = CB * S * 9;
  hipLaunchKernelGGL((wgrad_thin_reduce_kernel<BIG_IS_X>), dim3((nout + 255) / 256), dim3(256), 0, s, a.dwp, dw, db, S, CB, a.B * ncol, Cin);
  MCEDM_LAUNCH_CHECK("wgrad_thin_reduce_kernel");
  return MCEDM_OK;
}

int launch_wgrad_thin(const WgradArgs& a, float* dw, float* db, hipStream_t s) {
  MCEDM_REQUIRE(wgrad_thin_applicable(a, 9, 0), "wgrad_thin: shape not served");
  MCEDM_REQUIRE(a.dy && a.dwp && dw, "wgrad_thin: null pointer");
  const int Cin = a.Ca + a.Cb;
  if (a.Cout <= 4 && Cin >= 16) {
    switch (a.Cout) {
      case 1: return launch_thin_cfg<1, true>(a, dw, db, s);
      case 2: return launch_thin_cfg<2, true>(a, dw, db, s);
      case 3: return launch_thin_cfg<3, true>(a, dw, db, s);
      default: return launch_thin_cfg<4, true>(a, dw, db, s);
    }
  }
  switch (Cin) {
    case 1: return launch_thin_cfg<1, false>(a, dw, db, s);
    case 2: return launch_thin_cfg<2, false>(a, dw, db, s);
    case 3: return launch_thin_cfg<3, false>(a, dw, db, s);
    default: return launch_thin_cfg<4, false>(a, dw, db, s);
  }
}

}  // namespace mcedm
