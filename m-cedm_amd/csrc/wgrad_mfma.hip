// wgrad_mfma.hip -- K10a: weight (and bias) gradient of the fused 3x3 / 1x1 convolution on fp32 MFMA.
//
//   dW[co][ci][tap] = sum_{n, pixel} dY[n][co][pixel] * X'[n][ci][pixel + tap]
//   X' = resample(act(coef(cat(xa, xb))))   -- the SAME fused input transform as the forward (conv_mfma.hip),
//                                              recomputed while staging, so activations are never materialised.
//
// GEMM view per tap: D[co][ci] += A[co][pixel] * B[pixel][ci]; the contraction runs over pixels (2 per MFMA).
// A workgroup owns a 64 (co) x 64 (ci) block of dW for all taps (4 waves x one 32x32 tile x TAPS accumulators)
// and walks a strided subset of the 64-pixel tiles of the batch (split-K); partial sums are added into a
// [tap][CoP][CiP] fp32 scratch with float atomics (ci contiguous: two 128-B segments per wave-instruction), which
// wgrad_finish_kernel converts to the reference [co][ci][kh][kw] layout.  The block with ci-tile 0 also sums its
// dY tiles over pixels: that is the bias gradient.
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

__device__ __forceinline__ float silu_w(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// ---- shared tail: contraction of one staged tile, and the final atomic accumulation -----------------------
template <class C>
__device__ __forceinline__ void wg_contract(const float* dyl, const float* al, f32x16 (&acc)[C::TAPS], int mi, int ni,
                                            int l31, int h) {
  const float* arow = dyl + (mi * 32 + l31) * C::DP + h;
  const float* brow = al + (ni * 32 + l31) * C::AP + h;
#pragma unroll 4
  for (int s = 0; s < C::NPIX / 2; ++s) {
    const int pix = 2 * s;
    const float a = arow[pix];
    const int boff = (pix / C::PW) * C::PITCH + (pix % C::PW);
#pragma unroll
    for (int tap = 0; tap < C::TAPS; ++tap) {
      const int toff = (C::TAPS == 9) ? (tap / 3) * C::PITCH + (tap % 3) : 0;
      acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, brow[boff + toff], acc[tap], 0, 0, 0);
    }
  }
}

template <class C>
__device__ __forceinline__ void wg_flush(const WgradArgs& p, f32x16 (&acc)[C::TAPS], float bsum, int co0, int ci0, int it,
                                         int mi, int ni, int l31, int h, int tid, int cop, int cip) {
  const int Cin = p.Ca + p.Cb;
  const int ci = ci0 + ni * 32 + l31;
#pragma unroll
  for (int tap = 0; tap < C::TAPS; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < p.Cout && ci < Cin) atomicAdd(p.dwp + ((size_t)tap * cop + co) * cip + ci, acc[tap][r]);
    }
  if (it == 0 && tid < C::CT && co0 + tid < p.Cout && p.dbp) atomicAdd(p.dbp + co0 + tid, bsum);
}

// ---- hot variant (no resampling): flat, branch-free staging in batches (12 loads in flight, then 12 commits);
// the two workgroups resident on a CU overlap one's staging with the other's MFMAs.  Each workgroup walks a
// CONTIGUOUS range of tiles so the per-sample transform rows (kept in LDS) change rarely.
template <class C>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs p, int tiles_x, int tiles_y, int ctiles, int itiles,
                                                       int nsplit, int ntiles, int cop, int cip) {
  __shared__ float dyl[C::CT * C::DP];
  __shared__ float al[C::IT * C::AP];
  __shared__ Coef cfl[C::IT];
  constexpr int NDY = C::CT * C::NPIX / 4 / 256;               // float4 groups of dY per thread
  constexpr int NX = (C::IT * C::PLANE + 255) / 256;           // input elements per thread
  static_assert(C::CT * C::NPIX % 1024 == 0 && NX <= 64, "staging shape");
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
  if (t_begin >= t_end) return;      // (whole workgroup: t_begin depends on blockIdx only)

  const size_t HW = (size_t)p.H * p.W;
  const bool vec_ok = (p.W % 4 == 0) && ((reinterpret_cast<size_t>(p.dy) & 15) == 0);
  const float* safe = p.xa ? p.xa : p.xb;
  const int tiles_img = tiles_x * tiles_y;

  // stage one tile: global -> (transform) -> LDS in batches of XB elements (XB loads in flight, then XB commits)
  constexpr int XB = 12;
  auto stage_tile = [&](int t) {
    const int n = t / tiles_img;
    const int q = t - n * tiles_img;
    const int y0 = (q / tiles_x) * C::PH, x0 = (q % tiles_x) * C::PW;
    f32x4 dyv[NDY];
    bool dok[NDY];
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int i4 = tid + 256 * j;
      const int cl = i4 / (C::NPIX / 4), pp = (i4 % (C::NPIX / 4)) * 4;
      const int y = y0 + pp / C::PW, x = x0 + pp % C::PW;
      const int co = co0 + cl;
      dok[j] = co < p.Cout && y < p.H && x < p.W;
      const float* src = p.dy + ((size_t)n * p.Cout + (dok[j] ? co : 0)) * HW + (dok[j] ? (size_t)y * p.W + x : 0);
      if (vec_ok) {
        dyv[j] = *reinterpret_cast<const f32x4*>(src);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) dyv[j][e] = (dok[j] && x + e < p.W) ? src[e] : 0.f;
      }
    }
#pragma unroll 1
    for (int j0 = 0; j0 < NX; j0 += XB) {     // rolled: only one batch of registers is live at a time
      float xr[XB];
      bool xok[XB];
#pragma unroll
      for (int jj = 0; jj < XB; ++jj) {
        const int j = j0 + jj;
        if (j < NX) {
          const int idx = tid + 256 * j;
          const int cil = idx / C::PLANE, e = idx - cil * C::PLANE;
          const int r = e / C::PITCH, c = e - r * C::PITCH;
          const int y = y0 + r - C::HALO, x = x0 + c - C::HALO;
          const int ci = ci0 + cil;
          const bool in_a = ci < p.Ca;
          const float* base = in_a ? p.xa : p.xb;
          const int cc = in_a ? ci : ci - p.Ca, CC = in_a ? p.Ca : p.Cb;
          xok[jj] = idx < C::IT * C::PLANE && ci < Cin && base != nullptr && (unsigned)y < (unsigned)p.H &&
                    (unsigned)x < (unsigned)p.W;
          const float* ptr = xok[jj] ? base + ((size_t)n * CC + cc) * HW + (size_t)y * p.W + x : safe;
          xr[jj] = *ptr;
        }
      }
#pragma unroll
      for (int jj = 0; jj < XB; ++jj) {
        const int j = j0 + jj;
        if (j < NX) {
          const int idx = tid + 256 * j;
          if (idx < C::IT * C::PLANE) {
            const int cil = idx / C::PLANE, e = idx - cil * C::PLANE;
            const Coef cf = cfl[cil];
            const float tt = (xr[jj] - cf.mean) * cf.scale + cf.offset;
            const float v = p.act ? silu_w(tt) : tt;
            al[cil * C::AP + e] = xok[jj] ? v : 0.f;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int i4 = tid + 256 * j;
      const int cl = i4 / (C::NPIX / 4), pp = (i4 % (C::NPIX / 4)) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) dyl[cl * C::DP + pp + e] = dok[j] ? dyv[j][e] : 0.f;
    }
  };
  auto load_coefs = [&](int n) {
    if (tid < C::IT) {
      const int ci = ci0 + tid;
      Coef cf{0.f, 1.f, 0.f, 0.f};
      if (p.coef && ci < Cin) cf = p.coef[(p.coef_batch ? (size_t)n * Cin : 0) + ci];
      cfl[tid] = cf;
    }
  };

  int n_cur = -1;
  for (int t = t_begin; t < t_end; ++t) {
    const int n = t / tiles_img;
    if (n != n_cur) {          // transform rows of this sample -> LDS (rare: tiles of a workgroup are contiguous)
      load_coefs(n);
      n_cur = n;
      __syncthreads();
    }
    stage_tile(t);
    __syncthreads();
    if (it == 0 && tid < C::CT) {     // bias gradient: row sums of the dY tile
      float s = 0.f;
#pragma unroll 8
      for (int pp = 0; pp < C::NPIX; ++pp) s += dyl[tid * C::DP + pp];
      bsum += s;
    }
    wg_contract<C>(dyl, al, acc, mi, ni, l31, h);
    __syncthreads();
  }
  wg_flush<C>(p, acc, bsum, co0, ci0, it, mi, ni, l31, h, tid, cop, cip);
}

// ---- resampled variant (2x up / down between the activation and the conv): simple synchronous staging
template <class C>
__global__ __launch_bounds__(256, 2) void wgrad_resampled_kernel(WgradArgs p, int tiles_x, int tiles_y, int ctiles,
                                                                 int itiles, int nsplit, int ntiles, int cop, int cip) {
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

  f32x16 acc[C::TAPS];
#pragma unroll
  for (int t = 0; t < C::TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const size_t HW = (size_t)p.H * p.W;
  const size_t src_plane = (size_t)p.Hs * p.Ws;
  for (int t = split; t < ntiles; t += nsplit) {
    int q = t;
    const int tx = q % tiles_x; q /= tiles_x;
    const int ty = q % tiles_y;
    const int n = q / tiles_y;
    const int y0 = ty * C::PH, x0 = tx * C::PW;
#pragma unroll 4
    for (int idx = tid; idx < C::CT * C::NPIX; idx += 256) {
      const int cl = idx / C::NPIX, pp = idx % C::NPIX;
      const int y = y0 + pp / C::PW, x = x0 + pp % C::PW;
      const int co = co0 + cl;
      float v = 0.f;
      if (co < p.Cout && y < p.H && x < p.W) v = p.dy[((size_t)n * p.Cout + co) * HW + (size_t)y * p.W + x];
      dyl[cl * C::DP + pp] = v;
    }
    for (int cil = 0; cil < C::IT; ++cil) {
      const int ci = ci0 + cil;
      const bool in_a = ci < p.Ca;
      const float* src = in_a ? p.xa : p.xb;
      const int cc = in_a ? ci : ci - p.Ca;
      const int CC = in_a ? p.Ca : p.Cb;
      const bool chan_ok = (ci < Cin) && (src != nullptr);
      Coef cf{0.f, 1.f, 0.f, 0.f};
      if (chan_ok && p.coef) cf = p.coef[(p.coef_batch ? (size_t)n * Cin : 0) + ci];
      const float* plane = chan_ok ? src + ((size_t)n * CC + cc) * src_plane : nullptr;
#pragma unroll
      for (int sub = 0; sub < C::SUB; ++sub) {
        const int e = tid + sub * 256;
        if (e < C::PLANE) {
          const int r = e / C::PITCH, c = e - r * C::PITCH;
          const int y = y0 + r - C::HALO, x = x0 + c - C::HALO;
          float v = 0.f;
          if (chan_ok && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
            if (p.resample == RS_UP) {
              const float tt = (plane[(size_t)(y >> 1) * p.Ws + (x >> 1)] - cf.mean) * cf.scale + cf.offset;
              v = p.act ? silu_w(tt) : tt;
            } else if (p.resample == RS_DOWN) {
              const float* q0 = plane + (size_t)(2 * y) * p.Ws + 2 * x;
              float s4[4] = {q0[0], q0[1], q0[p.Ws], q0[p.Ws + 1]};
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float tt = (s4[k] - cf.mean) * cf.scale + cf.offset;
                s4[k] = p.act ? silu_w(tt) : tt;
              }
              v = 0.25f * ((s4[0] + s4[1]) + (s4[2] + s4[3]));
            } else {
              const float tt = (plane[(size_t)y * p.Ws + x] - cf.mean) * cf.scale + cf.offset;
              v = p.act ? silu_w(tt) : tt;
            }
          }
          al[cil * C::AP + e] = v;
        }
      }
    }
    __syncthreads();
    if (it == 0 && tid < C::CT) {
      float s = 0.f;
#pragma unroll 8
      for (int pp = 0; pp < C::NPIX; ++pp) s += dyl[tid * C::DP + pp];
      bsum += s;
    }
    wg_contract<C>(dyl, al, acc, mi, ni, l31, h);
    __syncthreads();
  }
  wg_flush<C>(p, acc, bsum, co0, ci0, it, mi, ni, l31, h, tid, cop, cip);
}

// grads in the reference layout: dW[co][ci][tap] = scratch[tap][perm(co)][ci]; db[co] = scratch_b[perm(co)].
// qkv_heads > 0: scratch rows are in packed (head, {q,k,v}, c) order, the parameter in (head, c, {q,k,v}) order.
__global__ void wgrad_finish_kernel(const float* __restrict__ dwp, const float* __restrict__ dbp, float* __restrict__ dw,
                                    float* __restrict__ db, int Cout, int Cin, int taps, int cop, int cip,
                                    int qkv_heads) {
  const size_t total = (size_t)Cout * Cin * taps;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total + Cout; i += (size_t)gridDim.x * blockDim.x) {
    const bool is_b = i >= total;
    const int co = is_b ? (int)(i - total) : (int)(i / ((size_t)Cin * taps));
    int cp = co;       // packed row holding reference row `co`
    if (qkv_heads > 0) {
      const int per = Cout / qkv_heads, d = per / 3;
      const int hh = co / per, rr = co % per, c = rr / 3, which = rr % 3;
      cp = hh * per + which * d + c;
    }
    if (is_b) {
      if (db) db[co] = dbp[cp];
    } else {
      const size_t rem = i % ((size_t)Cin * taps);
      const int ci = (int)(rem / taps), tap = (int)(rem % taps);
      dw[i] = dwp[((size_t)tap * cop + cp) * cip + ci];
    }
  }
}

size_t wgrad_scratch_floats(int Cout, int Cin, int taps) {
  const size_t cop = (Cout + 63) / 64 * 64, cip = (Cin + 63) / 64 * 64;
  return taps * cop * cip + cop;
}

template <class C>
static int launch_wg(const WgradArgs& a, hipStream_t s) {
  const int tiles_x = ceil_div(a.W, C::PW), tiles_y = ceil_div(a.H, C::PH);
  const int ctiles = ceil_div(a.Cout, C::CT), itiles = ceil_div(a.Ca + a.Cb, C::IT);
  const int ntiles = a.B * tiles_x * tiles_y;
  int nsplit = ceil_div(768, ctiles * itiles);
  if (nsplit > ntiles) nsplit = ntiles;
  if (nsplit < 1) nsplit = 1;
  const int cop = ctiles * 64, cip = itiles * 64;
  static char name[64];
  if (prof_enabled()) snprintf(name, sizeof(name), "wgrad_kernel<WgCfg<%d, %d, %d>>", C::PH, C::PW, C::TAPS);
  const double flops = 2.0 * a.B * a.H * (double)a.W * a.Cout * (a.Ca + a.Cb) * C::TAPS;
  ProfScope ps(name, flops, 4.0 * a.B * ((double)a.Cout * a.H * a.W + (double)(a.Ca + a.Cb) * a.Hs * a.Ws), s);
  if (a.resample == RS_NONE)
    hipLaunchKernelGGL(wgrad_kernel<C>, dim3(ctiles * itiles * nsplit), dim3(256), 0, s, a, tiles_x, tiles_y, ctiles,
                       itiles, nsplit, ntiles, cop, cip);
  else
    hipLaunchKernelGGL(wgrad_resampled_kernel<C>, dim3(ctiles * itiles * nsplit), dim3(256), 0, s, a, tiles_x, tiles_y,
                       ctiles, itiles, nsplit, ntiles, cop, cip);
  MCEDM_LAUNCH_CHECK("wgrad_kernel");
  return MCEDM_OK;
}

int launch_wgrad(const WgradArgs& a, int taps, float* dw, float* db, int qkv_heads, hipStream_t s) {
  MCEDM_REQUIRE(taps == 9 || taps == 1, "wgrad: taps must be 9 or 1");
  MCEDM_REQUIRE(a.dy && a.dwp && dw, "wgrad: null pointer");
  const int Cin = a.Ca + a.Cb;
  const size_t cop = (a.Cout + 63) / 64 * 64, cip = (Cin + 63) / 64 * 64;
  WgradArgs b = a;
  b.dbp = a.dwp + taps * cop * cip;
  MCEDM_HIP_TRY(hipMemsetAsync(a.dwp, 0, wgrad_scratch_floats(a.Cout, Cin, taps) * sizeof(float), s));
  int rc;
  if (taps == 9) {
    if (a.W >= 24) rc = launch_wg<WgCfg<2, 32, 9>>(b, s);
    else if (a.W >= 12) rc = launch_wg<WgCfg<4, 16, 9>>(b, s);
    else rc = launch_wg<WgCfg<8, 8, 9>>(b, s);
  } else {
    if (a.W >= 24) rc = launch_wg<WgCfg<2, 32, 1>>(b, s);
    else if (a.W >= 12) rc = launch_wg<WgCfg<4, 16, 1>>(b, s);
    else rc = launch_wg<WgCfg<8, 8, 1>>(b, s);
  }
  if (rc) return rc;
  const size_t total = (size_t)a.Cout * Cin * taps + a.Cout;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3(blocks), dim3(256), 0, s, a.dwp, b.dbp, dw, db, a.Cout, Cin, taps, (int)cop,
                     (int)cip, qkv_heads);
  MCEDM_LAUNCH_CHECK("wgrad_finish_kernel");
  return MCEDM_OK;
}

}  // namespace mcedm
