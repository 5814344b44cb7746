// bwd_kernels.hip -- K10b: backward of GroupNorm+FiLM+SiLU(+2x resampling), parameter-gradient reductions,
// the tiny dense products of the embedding MLP backward, and the attention backward (flash-style recompute).
#include <atomic>
#include <cstdlib>

#include "bwd.hpp"
#include "prof.hpp"

namespace mcedm {

// =========================================================================================================
// Backward through  u = resample(act(t)),  t = (x - mean) * rstd * g_c + o_c,  g_c = gamma_c (1 + s_nc)
// (adm_blocks.py:94-97,161,166 and the resampling of :72-77).  One workgroup per (sample, group), two passes
// over the group's data (the second mostly L2-resident): pass 1 reduces A_c = sum dt and B_c = sum dt*xhat per
// channel, pass 2 writes  dx = rstd * (g_c dt - mean_grp(g dt) - xhat * mean_grp(g dt xhat))  (+ extra gradient).
// The activation gradient dt is recomputed from x in both passes and never stored.
// =========================================================================================================
__device__ __forceinline__ float dsilu(float t) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-t));
  return sg * (1.0f + t * (1.0f - sg));
}

// gradient arriving at source pixel (ys, xs) through the forward resampling
__device__ __forceinline__ float fetch_resampled(const float* plane, int mode, int ys, int xs, int Ws) {
  if (mode == RS_NONE) return plane[(size_t)ys * Ws + xs];
  if (mode == RS_UP) {      // forward nearest-2x: each source pixel fed a 2x2 patch of the conv input
    const int Wc = 2 * Ws;
    const float* q = plane + (size_t)(2 * ys) * Wc + 2 * xs;
    return (q[0] + q[1]) + (q[Wc] + q[Wc + 1]);
  }
  const int Wc = Ws >> 1;   // forward 2x2 mean: each source pixel contributed a quarter
  return 0.25f * plane[(size_t)(ys >> 1) * Wc + (xs >> 1)];
}

constexpr int GN_MAX_CPG = 32;

template <int NT>
__global__ __launch_bounds__(NT) void gn_bwd_kernel(GnBwdArgs a) {
  const int C = a.Ca + a.Cb;
  const int cpg = C / a.groups;
  const int n = blockIdx.x / a.groups, g = blockIdx.x % a.groups;
  const int c0 = g * cpg;
  const int HWs = a.Hs * a.Ws;
  const int HWc = a.resample == RS_UP ? HWs * 4 : (a.resample == RS_DOWN ? HWs / 4 : HWs);
  const float mean = a.stats[((size_t)n * a.groups + g) * 2], rstd = a.stats[((size_t)n * a.groups + g) * 2 + 1];
  const int tid = threadIdx.x;
  const bool vec = a.resample == RS_NONE && (HWs & 3) == 0;      // all planes are then 16-byte aligned
  __shared__ double red[2][NT / 64];
  __shared__ float sA[GN_MAX_CPG], sB[GN_MAX_CPG], sM[2];

  for (int cl = 0; cl < cpg; ++cl) {
    const int c = c0 + cl;
    const float* x = (c < a.Ca) ? a.xa + ((size_t)n * a.Ca + c) * HWs : a.xb + ((size_t)n * a.Cb + (c - a.Ca)) * HWs;
    const float* dpl = a.dact + ((size_t)n * C + c) * HWc;
    const Coef cf = a.coef[(size_t)n * C + c];
    float pa = 0.f, pb = 0.f;
    if (vec) {            // no resampling, HW % 4 == 0: 16-byte loads
      const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
      const f32x4* d4 = reinterpret_cast<const f32x4*>(dpl);
      // four independent 16-byte load pairs in flight per thread (the loop is latency-bound otherwise: 2.4 TB/s)
      const int n4 = HWs / 4;
      int p = tid;
      for (; p + 3 * NT < n4; p += 4 * NT) {
        f32x4 xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { xv[u] = x4[p + NT * u]; dv[u] = d4[p + NT * u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float dt = dv[u][e];
            if (a.act) dt *= dsilu((xv[u][e] - cf.mean) * cf.scale + cf.offset);
            pa += dt;
            pb += dt * ((xv[u][e] - mean) * rstd);
          }
      }
      for (; p < n4; p += NT) {
        const f32x4 xv = x4[p], dv = d4[p];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float dt = dv[e];
          if (a.act) dt *= dsilu((xv[e] - cf.mean) * cf.scale + cf.offset);
          pa += dt;
          pb += dt * ((xv[e] - mean) * rstd);
        }
      }
    } else {
      for (int p = tid; p < HWs; p += NT) {
        const int ys = p / a.Ws, xs = p - ys * a.Ws;
        const float xv = x[p];
        float dt = fetch_resampled(dpl, a.resample, ys, xs, a.Ws);
        if (a.act) dt *= dsilu((xv - cf.mean) * cf.scale + cf.offset);
        pa += dt;
        pb += dt * ((xv - mean) * rstd);
      }
    }
    double da = pa, db = pb;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { da += __shfl_xor(da, off); db += __shfl_xor(db, off); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = da; red[1][tid >> 6] = db; }
    __syncthreads();
    if (tid == 0) {
      double ra = 0.0, rb = 0.0;
#pragma unroll
      for (int w = 0; w < NT / 64; w += 4) {
        ra += (red[0][w] + red[0][w + 1]) + (red[0][w + 2] + red[0][w + 3]);
        rb += (red[1][w] + red[1][w + 1]) + (red[1][w + 2] + red[1][w + 3]);
      }
      const float A = (float)ra, Bs = (float)rb;
      sA[cl] = A; sB[cl] = Bs;
      a.ab[((size_t)n * C + c) * 2] = A;
      a.ab[((size_t)n * C + c) * 2 + 1] = Bs;
    }
    __syncthreads();
  }
  if (tid == 0) {
    double m1 = 0, m2 = 0;
    for (int cl = 0; cl < cpg; ++cl) {
      const int c = c0 + cl;
      float gc = a.gamma[c];
      if (a.film) gc *= 1.0f + a.film[(size_t)(a.film_batch ? n : 0) * a.film_stride + c];
      m1 += (double)gc * sA[cl];
      m2 += (double)gc * sB[cl];
    }
    const double N = (double)cpg * HWs;
    sM[0] = (float)(m1 / N); sM[1] = (float)(m2 / N);
  }
  __syncthreads();
  const float m1 = sM[0], m2 = sM[1];
  for (int cl = 0; cl < cpg; ++cl) {
    const int c = c0 + cl;
    const bool in_a = c < a.Ca;
    const size_t xo = in_a ? ((size_t)n * a.Ca + c) * HWs : ((size_t)n * a.Cb + (c - a.Ca)) * HWs;
    const float* x = (in_a ? a.xa : a.xb) + xo;
    float* dx = (in_a ? a.dxa : a.dxb) + xo;
    const float* dpl = a.dact + ((size_t)n * C + c) * HWc;
    const float* addp = a.add ? a.add + ((size_t)n * C + c) * (a.add_mode == 2 ? HWc : HWs) : nullptr;
    float* xq = a.xact ? a.xact + ((size_t)n * C + c) * HWs : nullptr;      // act(coef(x)): the conv's weight-gradient operand
    const Coef cf = a.coef[(size_t)n * C + c];
    if (vec) {
      const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
      const f32x4* d4 = reinterpret_cast<const f32x4*>(dpl);
      const f32x4* a4 = reinterpret_cast<const f32x4*>(addp);
      f32x4* o4 = reinterpret_cast<f32x4*>(dx);
#pragma unroll 2
      for (int p = tid; p < HWs / 4; p += NT) {
        const f32x4 xv = x4[p], dv = d4[p];
        f32x4 o = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};
        if (addp) o = a4[p];
        if (a.accumulate) { const f32x4 old = o4[p]; o += old; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float dt = dv[e];
          const float tt = (xv[e] - cf.mean) * cf.scale + cf.offset;
          if (a.act) {
            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-tt));     // dsilu and silu from one sigmoid
            dt *= sg * (1.0f + tt * (1.0f - sg));
            u[e] = tt * sg;
          } else {
            u[e] = tt;
          }
          const float xh = (xv[e] - mean) * rstd;
          o[e] += cf.scale * dt - rstd * (m1 + xh * m2);
        }
        o4[p] = o;
        if (xq) reinterpret_cast<f32x4*>(xq)[p] = u;
      }
      continue;
    }
    for (int p = tid; p < HWs; p += NT) {
      const int ys = p / a.Ws, xs = p - ys * a.Ws;
      const float xv = x[p];
      float dt = fetch_resampled(dpl, a.resample, ys, xs, a.Ws);
      const float tt = (xv - cf.mean) * cf.scale + cf.offset;
      float uu = tt;
      if (a.act) {
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-tt));
        dt *= sg * (1.0f + tt * (1.0f - sg));
        uu = tt * sg;
      }
      const float xh = (xv - mean) * rstd;
      float v = cf.scale * dt - rstd * (m1 + xh * m2);      // cf.scale == rstd * g_c
      if (addp) v += (a.add_mode == 2) ? fetch_resampled(addp, a.resample, ys, xs, a.Ws) : addp[p];
      if (a.accumulate) v += dx[p];
      dx[p] = v;
      if (xq) xq[p] = uu;
    }
  }
}

// ---- the slab on chip between the two passes (round 5) ---------------------------------------------------------------------
// gn_bwd_kernel reads x and the incoming gradient twice (six to eight tensor passes for the algorithmic three).  Here a (sample,
// group) slab of >= 64 KB is cut into pieces of 4096 elements of one channel (32 KB of LDS for x and dact: four 256-thread
// workgroups per CU, so that one workgroup's loads overlap another's stores -- one 128 KB workgroup per CU was 60 % SLOWER than the
// two-pass kernel).  A workgroup keeps its piece in LDS, publishes the piece's (sum dt, sum dt xhat) in a table behind the counters,
// and waits for the k - 1 other pieces of the slab -- for a BOUNDED time (partners are adjacent workgroup ids: dispatched together).  A workgroup whose wait runs out computes the missing pieces' sums itself from global memory (the same
// function, the same bits), so no schedule can hang the kernel; the last workgroup to leave clears the exchange area (zero on
// entry, zero on exit).  Piece sums are added per channel in piece order in fp64: deterministic.  Un-resampled inputs only.
// four consecutive source pixels (one row, Ws % 4 == 0) of the gradient that arrives through the forward resampling: the vector form
// of fetch_resampled (same additions in the same order)
__device__ __forceinline__ f32x4 fetch_resampled_quad(const float* plane, int mode, int e0, int Ws) {
  if (mode == RS_NONE) return *reinterpret_cast<const f32x4*>(plane + e0);
  const int ys = e0 / Ws, xs = e0 - ys * Ws;
  if (mode == RS_UP) {
    const int Wc = 2 * Ws;
    const float* q = plane + (size_t)(2 * ys) * Wc + 2 * xs;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(q), a1 = *reinterpret_cast<const f32x4*>(q + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(q + Wc), b1 = *reinterpret_cast<const f32x4*>(q + Wc + 4);
    return f32x4{(a0[0] + a0[1]) + (b0[0] + b0[1]), (a0[2] + a0[3]) + (b0[2] + b0[3]), (a1[0] + a1[1]) + (b1[0] + b1[1]),
                 (a1[2] + a1[3]) + (b1[2] + b1[3])};
  }
  const int Wc = Ws >> 1;
  const f32x2 v = *reinterpret_cast<const f32x2*>(plane + (size_t)(ys >> 1) * Wc + (xs >> 1));
  return f32x4{0.25f * v[0], 0.25f * v[0], 0.25f * v[1], 0.25f * v[1]};
}

constexpr int GNL_PIECE = 4096;                 // elements per workgroup
constexpr int GNL_KMAX = 64;                    // workgroups per slab
constexpr int GNL_SYNC_WORDS = 2 + 2 * GNL_KMAX;   // per slab: a departure counter (+ 1 unused) and KMAX 64-bit slots
static_assert(GNL_SYNC_WORDS == MCEDM_GN_SYNC_WORDS, "mcedm_hip.h");

__global__ __launch_bounds__(256) void gn_bwd_lds_kernel(GnBwdArgs a, int ppc, unsigned spin_ticks) {
  constexpr int NT = 256, NQ = GNL_PIECE / 4;
  __shared__ f32x4 xs[NQ], ds[NQ];
  __shared__ double red[2][NT / 64];
  __shared__ float sP[GNL_KMAX][2], sM[2];
  __shared__ int s_have;
  const int C = a.Ca + a.Cb;
  const int cpg = C / a.groups, k = cpg * ppc;
  const int slab = blockIdx.x / k, j = blockIdx.x - slab * k;
  const int n = slab / a.groups, g = slab - n * a.groups;
  const int c0 = g * cpg;
  const int HWs = a.Hs * a.Ws;
  const int HWc = a.resample == RS_UP ? HWs * 4 : (a.resample == RS_DOWN ? HWs / 4 : HWs);
  const unsigned nslabs = gridDim.x / k;
  unsigned* const depart = a.sync + slab;                       // [nslabs] counters, [nslabs] unused, then [nslabs][KMAX] 64-bit slots
  unsigned long long* const slots = reinterpret_cast<unsigned long long*>(a.sync + 2 * nslabs) + (size_t)slab * GNL_KMAX;
  const float mean = a.stats[((size_t)n * a.groups + g) * 2], rstd = a.stats[((size_t)n * a.groups + g) * 2 + 1];
  const int tid = threadIdx.x;

  // piece jj = (channel c0 + jj / ppc, elements [jj % ppc * PIECE, + PIECE)): its two sums -> sP[jj]
  auto piece_sums = [&](int jj, bool own) {
    const int c = c0 + jj / ppc;
    const size_t off = (size_t)(jj % ppc) * GNL_PIECE;
    const float* x = ((c < a.Ca) ? a.xa + ((size_t)n * a.Ca + c) * HWs : a.xb + ((size_t)n * a.Cb + (c - a.Ca)) * HWs) + off;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const float* dpl = a.dact + ((size_t)n * C + c) * HWc;
    const Coef cf = a.coef[(size_t)n * C + c];
    f32x4 xv[NQ / NT], dv[NQ / NT];
#pragma unroll
    for (int u = 0; u < NQ / NT; ++u) {
      xv[u] = x4[tid + NT * u];
      dv[u] = fetch_resampled_quad(dpl, a.resample, (int)off + 4 * (tid + NT * u), a.Ws);
    }
    float pa = 0.f, pb = 0.f;
#pragma unroll
    for (int u = 0; u < NQ / NT; ++u) {
      if (own) { xs[tid + NT * u] = xv[u]; ds[tid + NT * u] = dv[u]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dt = dv[u][e];
        if (a.act) dt *= dsilu((xv[u][e] - cf.mean) * cf.scale + cf.offset);
        pa += dt;
        pb += dt * ((xv[u][e] - mean) * rstd);
      }
    }
    double da = pa, db = pb;
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) { da += __shfl_xor(da, off2); db += __shfl_xor(db, off2); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = da; red[1][tid >> 6] = db; }
    __syncthreads();
    if (tid == 0) {
      sP[jj][0] = (float)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
      sP[jj][1] = (float)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
    __syncthreads();
  };

  piece_sums(j, true);
  // exchange: every piece's two sums travel as ONE 64-bit word through a slot that is zero until published (a zero first sum is sent
  // as -0.0f), written and polled with agent-scope atomics: no fence anywhere -- a fence here is a write-back / invalidate of the
  // XCD's whole L2 per workgroup (the first version: 5x slower than the two-pass kernel).
  if (tid < 64) {                                               // wave 0: lane L watches slot L
    int have = 1;
    if (k > 1) {
      if (tid == 0) {
        float A = sP[j][0];
        if (A == 0.f) A = -0.f;
        const unsigned long long w = ((unsigned long long)__builtin_bit_cast(unsigned, sP[j][1]) << 32) | __builtin_bit_cast(unsigned, A);
        __hip_atomic_store(slots + j, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned long long v = 0;
      const bool mine = tid < k && tid != j;
      for (;;) {
        if (mine && v == 0) v = __hip_atomic_load(slots + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        have = __all(!mine || v != 0);
        if (have || __builtin_amdgcn_s_memrealtime() - t0 > spin_ticks) break;
        __builtin_amdgcn_s_sleep(8);
      }
      if (have && mine) {
        sP[tid][0] = __builtin_bit_cast(float, (unsigned)(v & 0xffffffffull));
        sP[tid][1] = __builtin_bit_cast(float, (unsigned)(v >> 32));
      }
    }
    if (tid == 0) s_have = have;
  }
  __syncthreads();
  if (!s_have) {                                               // the wait ran out: the other pieces' sums from global memory, same bits
    for (int jj = 0; jj < k; ++jj)
      if (jj != j) piece_sums(jj, false);
  }
  if (tid == 0) {
    double m1 = 0, m2 = 0;
    for (int cl = 0; cl < cpg; ++cl) {
      const int c = c0 + cl;
      double A = 0, Bs = 0;
      for (int pi = 0; pi < ppc; ++pi) { A += (double)sP[cl * ppc + pi][0]; Bs += (double)sP[cl * ppc + pi][1]; }
      const float Af = (float)A, Bf = (float)Bs;
      if (j == cl * ppc) { a.ab[((size_t)n * C + c) * 2] = Af; a.ab[((size_t)n * C + c) * 2 + 1] = Bf; }      // one writer per channel
      float gc = a.gamma[c];
      if (a.film) gc *= 1.0f + a.film[(size_t)(a.film_batch ? n : 0) * a.film_stride + c];
      m1 += (double)gc * Af;
      m2 += (double)gc * Bf;
    }
    const double N = (double)cpg * HWs;
    sM[0] = (float)(m1 / N); sM[1] = (float)(m2 / N);
  }
  __syncthreads();
  const float m1 = sM[0], m2 = sM[1];
  {
    const int c = c0 + j / ppc;
    const size_t off = (size_t)(j % ppc) * GNL_PIECE;
    const bool in_a = c < a.Ca;
    const size_t xo = (in_a ? ((size_t)n * a.Ca + c) * HWs : ((size_t)n * a.Cb + (c - a.Ca)) * HWs) + off;
    f32x4* o4 = reinterpret_cast<f32x4*>((in_a ? a.dxa : a.dxb) + xo);
    const float* apl = a.add ? a.add + ((size_t)n * C + c) * (a.add_mode == 2 ? HWc : HWs) : nullptr;
    const int amode = a.add_mode == 2 ? a.resample : RS_NONE;
    f32x4* q4 = a.xact ? reinterpret_cast<f32x4*>(a.xact + ((size_t)n * C + c) * HWs + off) : nullptr;
    const Coef cf = a.coef[(size_t)n * C + c];
#pragma unroll
    for (int u = 0; u < NQ / NT; ++u) {
      const int p = tid + NT * u;
      const f32x4 xv = xs[p], dv = ds[p];
      f32x4 o = {0.f, 0.f, 0.f, 0.f}, uu = {0.f, 0.f, 0.f, 0.f};
      if (apl) o = fetch_resampled_quad(apl, amode, (int)off + 4 * p, a.Ws);
      if (a.accumulate) { const f32x4 old = o4[p]; o += old; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dt = dv[e];
        const float tt = (xv[e] - cf.mean) * cf.scale + cf.offset;
        if (a.act) {
          const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-tt));
          dt *= sg * (1.0f + tt * (1.0f - sg));
          uu[e] = tt * sg;
        } else {
          uu[e] = tt;
        }
        const float xh = (xv[e] - mean) * rstd;
        o[e] += cf.scale * dt - rstd * (m1 + xh * m2);
      }
      o4[p] = o;
      if (q4) q4[p] = uu;
    }
  }
  if (k > 1 && tid == 0) {                                     // the last one out clears the slab's slots and its counter
    const unsigned old = atomicAdd(depart, 1u);
    if (old == (unsigned)k - 1) {
      for (int jj = 0; jj < k; ++jj) __hip_atomic_store(slots + jj, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static int gn_bwd_lds_env() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_GN_BWD_LDS"); env = e ? atoi(e) : 1; }
  return env;
}

// does the LDS-resident kernel serve this call (shape, switch, counters)?  -> pieces per channel
static bool gn_bwd_lds_plan(const GnBwdArgs& a, int* ppc_out) {
  const int C = a.Ca + a.Cb, cpg = C / a.groups;
  const size_t HW = (size_t)a.Hs * a.Ws;
  if (!gn_bwd_lds_env() || !a.sync || HW % GNL_PIECE != 0 || (size_t)cpg * HW < 16384) return false;
  if (a.resample != RS_NONE && (a.Ws % 4 != 0 || a.Hs % 2 != 0 || (a.resample != RS_UP && a.resample != RS_DOWN))) return false;
  const size_t k = (size_t)cpg * (HW / GNL_PIECE);
  if (k > GNL_KMAX) return false;
  if (((reinterpret_cast<size_t>(a.xa) | reinterpret_cast<size_t>(a.xb) | reinterpret_cast<size_t>(a.dact) | reinterpret_cast<size_t>(a.dxa) |
        reinterpret_cast<size_t>(a.dxb) | reinterpret_cast<size_t>(a.add) | reinterpret_cast<size_t>(a.xact)) & 15) != 0) return false;
  *ppc_out = (int)(HW / GNL_PIECE);
  return true;
}

static int launch_gn_bwd_lds(const GnBwdArgs& a, int ppc, hipStream_t s) {
  static int spin_us = -1;                                   // MCEDM_GN_BWD_SPIN_US: how long a workgroup waits for its partners (0: never)
  if (spin_us < 0) { const char* e = getenv("MCEDM_GN_BWD_SPIN_US"); spin_us = e ? atoi(e) : 100; }
  const unsigned ticks = (unsigned)spin_us * 100u;           // s_memrealtime: 100 MHz
  const int k = (a.Ca + a.Cb) / a.groups * ppc;
  hipLaunchKernelGGL(gn_bwd_lds_kernel, dim3((unsigned)a.B * a.groups * k), dim3(256), 0, s, a, ppc, ticks);
  MCEDM_LAUNCH_CHECK("gn_bwd_lds_kernel");
  return MCEDM_OK;
}

size_t gn_bwd_sync_words(int B, int groups) { return (size_t)B * groups * GNL_SYNC_WORDS; }

// ---- small slabs: everything in registers, one memory round trip (round 5) ---------------------------------------------------
// gn_bwd_kernel walks a slab channel by channel -- load, reduce behind two barriers, next channel -- and then again for pass 2:
// for a slab of 4 x 32^2 elements that is eight dependent memory round trips and 16 barriers for 32 KB of data (20 us per launch;
// the largest single item of config 2's training step).  Here a thread owns R quads of the slab (<= 8192 elements per slab), ALL
// loads (x, dact, the extra gradient, the accumulated dx) are issued up front, the per-channel sums come out of one segmented
// shuffle reduction + one pass over <= 128 partials in a fixed order, and pass 2 runs on the registers.  Un-resampled inputs.
template <int R>
__global__ __launch_bounds__(256) void gn_bwd_reg_kernel(GnBwdArgs a, int segshift) {
  const int C = a.Ca + a.Cb;
  const int cpg = C / a.groups;
  const int n = blockIdx.x / a.groups, g = blockIdx.x - n * a.groups;
  const int c0 = g * cpg;
  const int HWs = a.Hs * a.Ws, HW4 = HWs >> 2, Q = cpg * HW4;
  const float mean = a.stats[((size_t)n * a.groups + g) * 2], rstd = a.stats[((size_t)n * a.groups + g) * 2 + 1];
  const int tid = threadIdx.x;
  const int seg = 1 << segshift;                               // lanes that share a channel (<= one wave)
  __shared__ double red[R][16][2];
  __shared__ float sM[2];
  f32x4 xv[R], dv[R], av[R], ov[R];
  Coef cf[R];
  size_t xoff[R], coff[R]; int cc[R]; bool ok[R];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int q = tid + 256 * r;
    ok[r] = q < Q;
    const int cl = ok[r] ? q / HW4 : 0, p = ok[r] ? q - cl * HW4 : 0;
    const int c = c0 + cl;
    cc[r] = c;
    const bool in_a = c < a.Ca;
    xoff[r] = (in_a ? ((size_t)n * a.Ca + c) * HWs : ((size_t)n * a.Cb + (c - a.Ca)) * HWs) + 4 * (size_t)p;
    coff[r] = ((size_t)n * C + c) * HWs + 4 * (size_t)p;
    xv[r] = dv[r] = av[r] = ov[r] = zero4;
    cf[r] = a.coef[(size_t)n * C + c];
    if (ok[r]) {
      xv[r] = *reinterpret_cast<const f32x4*>((in_a ? a.xa : a.xb) + xoff[r]);
      dv[r] = *reinterpret_cast<const f32x4*>(a.dact + coff[r]);
      if (a.add) av[r] = *reinterpret_cast<const f32x4*>(a.add + coff[r]);
      if (a.accumulate) ov[r] = *reinterpret_cast<const f32x4*>((in_a ? a.dxa : a.dxb) + xoff[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float pa = 0.f, pb = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dt = dv[r][e];
      if (a.act) dt *= dsilu((xv[r][e] - cf[r].mean) * cf[r].scale + cf[r].offset);
      pa += dt;
      pb += dt * ((xv[r][e] - mean) * rstd);
    }
    double da = pa, db = pb;
    for (int off = seg >> 1; off > 0; off >>= 1) { da += __shfl_xor(da, off); db += __shfl_xor(db, off); }
    if ((tid & (seg - 1)) == 0) { red[r][tid >> segshift][0] = da; red[r][tid >> segshift][1] = db; }
  }
  __syncthreads();
  if (tid == 0) {
    double m1 = 0, m2 = 0;
    for (int cl = 0; cl < cpg; ++cl) {                         // the partials of channel cl: quads [cl HW4, (cl + 1) HW4), in quad order
      double A = 0, Bs = 0;
      for (int q0 = cl * HW4; q0 < (cl + 1) * HW4; q0 += seg) {
        const int r = q0 >> 8, sgi = (q0 & 255) >> segshift;
        A += red[r][sgi][0]; Bs += red[r][sgi][1];
      }
      const int c = c0 + cl;
      const float Af = (float)A, Bf = (float)Bs;
      a.ab[((size_t)n * C + c) * 2] = Af; a.ab[((size_t)n * C + c) * 2 + 1] = Bf;
      float gc = a.gamma[c];
      if (a.film) gc *= 1.0f + a.film[(size_t)(a.film_batch ? n : 0) * a.film_stride + c];
      m1 += (double)gc * Af;
      m2 += (double)gc * Bf;
    }
    const double N = (double)cpg * HWs;
    sM[0] = (float)(m1 / N); sM[1] = (float)(m2 / N);
  }
  __syncthreads();
  const float m1 = sM[0], m2 = sM[1];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!ok[r]) continue;
    f32x4 o = av[r] + ov[r], u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dt = dv[r][e];
      const float tt = (xv[r][e] - cf[r].mean) * cf[r].scale + cf[r].offset;
      if (a.act) {
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-tt));
        dt *= sg * (1.0f + tt * (1.0f - sg));
        u[e] = tt * sg;
      } else {
        u[e] = tt;
      }
      const float xh = (xv[r][e] - mean) * rstd;
      o[e] += cf[r].scale * dt - rstd * (m1 + xh * m2);
    }
    const bool in_a = cc[r] < a.Ca;
    *reinterpret_cast<f32x4*>((in_a ? a.dxa : a.dxb) + xoff[r]) = o;
    if (a.xact) *reinterpret_cast<f32x4*>(a.xact + coff[r]) = u;
  }
}

// quads per thread (1, 2, 4, 8) when gn_bwd_reg_kernel serves the call, else 0; *segshift: log2 of the lanes that share a channel
static int gn_bwd_reg_plan(const GnBwdArgs& a, int* segshift) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_GN_BWD_REG"); env = e ? atoi(e) : 1; }
  const int C = a.Ca + a.Cb, cpg = C / a.groups;
  const size_t HW = (size_t)a.Hs * a.Ws;
  if (!env || a.resample != RS_NONE || HW % 4 != 0 || (size_t)cpg * HW > 8192) return 0;
  const int HW4 = (int)(HW / 4);
  int sh = 0;
  if (HW4 >= 64) { if (HW4 % 64 != 0) return 0; sh = 6; }
  else { if (HW4 < 16 || (HW4 & (HW4 - 1)) != 0) return 0; while ((1 << sh) < HW4) ++sh; }
  if (((reinterpret_cast<size_t>(a.xa) | reinterpret_cast<size_t>(a.xb) | reinterpret_cast<size_t>(a.dact) | reinterpret_cast<size_t>(a.dxa) |
        reinterpret_cast<size_t>(a.dxb) | reinterpret_cast<size_t>(a.add) | reinterpret_cast<size_t>(a.xact)) & 15) != 0) return 0;
  *segshift = sh;
  const int Q = cpg * HW4;
  return Q <= 256 ? 1 : Q <= 512 ? 2 : Q <= 1024 ? 4 : 8;
}

int launch_gn_bwd(const GnBwdArgs& a, hipStream_t s) {
  const int C = a.Ca + a.Cb;
  MCEDM_REQUIRE(a.groups > 0 && C % a.groups == 0 && C / a.groups <= GN_MAX_CPG, "gn_bwd: bad groups (C=%d groups=%d)", C, a.groups);
  MCEDM_REQUIRE(a.dact && a.xa && a.coef && a.stats && a.gamma && a.dxa && a.ab && (a.Cb == 0 || (a.xb && a.dxb)), "gn_bwd: null pointer");
  MCEDM_REQUIRE(a.resample != RS_DOWN || (a.Hs % 2 == 0 && a.Ws % 2 == 0), "gn_bwd: odd source size for a 2x2 mean");
  MCEDM_REQUIRE(!a.xact || a.resample == RS_NONE, "gn_bwd: the activated input is emitted for un-resampled convs only");
  int ppc = 0;
  if (gn_bwd_lds_plan(a, &ppc)) {
    ProfScope ps("gn_bwd_lds_kernel", 20.0 * a.B * (double)C * a.Hs * a.Ws, 4.0 * 3 * a.B * (double)C * a.Hs * a.Ws, s);
    return launch_gn_bwd_lds(a, ppc, s);
  }
  int segshift = 0;
  if (const int R = gn_bwd_reg_plan(a, &segshift)) {
    ProfScope ps("gn_bwd_reg_kernel", 20.0 * a.B * (double)C * a.Hs * a.Ws, 4.0 * 3 * a.B * (double)C * a.Hs * a.Ws, s);
    const dim3 grid(a.B * a.groups);
    if (R == 1) hipLaunchKernelGGL(gn_bwd_reg_kernel<1>, grid, dim3(256), 0, s, a, segshift);
    else if (R == 2) hipLaunchKernelGGL(gn_bwd_reg_kernel<2>, grid, dim3(256), 0, s, a, segshift);
    else if (R == 4) hipLaunchKernelGGL(gn_bwd_reg_kernel<4>, grid, dim3(256), 0, s, a, segshift);
    else hipLaunchKernelGGL(gn_bwd_reg_kernel<8>, grid, dim3(256), 0, s, a, segshift);
    MCEDM_LAUNCH_CHECK("gn_bwd_reg_kernel");
    return MCEDM_OK;
  }
  ProfScope ps("gn_bwd_kernel", 20.0 * a.B * (double)C * a.Hs * a.Ws, 4.0 * 3 * a.B * (double)C * a.Hs * a.Ws, s);
  // slabs of >= 64 KB: 1024-thread workgroups (at most two per CU instead of eight, so that fewer slabs are between their two passes
  // at any time and more of pass 2 is served by the memory-side cache: 5.78 -> 5.43 ms per S128 step, 3.12 -> 2.77 ms on the ch = 64 network)
  static int nt = -1;
  if (nt < 0) { const char* e = getenv("MCEDM_GN_BWD_NT"); nt = e ? atoi(e) : 1024; }
  if (nt == 1024 && (size_t)(C / a.groups) * a.Hs * a.Ws >= 16384)
    hipLaunchKernelGGL(gn_bwd_kernel<1024>, dim3(a.B * a.groups), dim3(1024), 0, s, a);
  else
    hipLaunchKernelGGL(gn_bwd_kernel<256>, dim3(a.B * a.groups), dim3(256), 0, s, a);
  MCEDM_LAUNCH_CHECK("gn_bwd_kernel");
  return MCEDM_OK;
}

// dgamma_c = sum_n (1+s_nc) B_nc ; dbeta_c = sum_n (1+s_nc) A_nc ;
// d scale_nc = gamma_c B_nc + beta_c A_nc ; d shift_nc = A_nc          (t = xhat*gamma*(1+s) + beta*(1+s) + shift)
__global__ void gn_param_grads_kernel(const float* __restrict__ ab, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, const float* __restrict__ film, int film_batch,
                                      int film_stride, int B, int C, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta, float* __restrict__ dfilm, int dfilm_stride) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float gm = gamma[c], bt = beta[c];
  double dg = 0, db = 0;
  float ds0 = 0.f, dh0 = 0.f;
#pragma unroll 8                       // the loads of eight samples in flight (a chain of B dependent round trips otherwise: 14 us per launch)
  for (int n = 0; n < B; ++n) {
    const float A = ab[((size_t)n * C + c) * 2], Bs = ab[((size_t)n * C + c) * 2 + 1];
    float sc = 1.f;
    if (film) sc += film[(size_t)(film_batch ? n : 0) * film_stride + c];
    dg += (double)sc * Bs;
    db += (double)sc * A;
    if (dfilm) {
      const float dsc = gm * Bs + bt * A;
      if (film_batch) {
        dfilm[(size_t)n * dfilm_stride + c] = dsc;
        dfilm[(size_t)n * dfilm_stride + C + c] = A;
      } else {
        ds0 += dsc; dh0 += A;
      }
    }
  }
  dgamma[c] = (float)dg;
  dbeta[c] = (float)db;
  if (dfilm && !film_batch) { dfilm[c] = ds0; dfilm[C + c] = dh0; }
}

int launch_gn_param_grads(const float* ab, const float* gamma, const float* beta, const float* film, int film_batch,
                          int film_stride, int B, int C, float* dgamma, float* dbeta, float* dfilm, int dfilm_stride,
                          hipStream_t s) {
  hipLaunchKernelGGL(gn_param_grads_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, s, ab, gamma, beta, film, film_batch,
                     film_stride, B, C, dgamma, dbeta, dfilm, dfilm_stride);
  MCEDM_LAUNCH_CHECK("gn_param_grads_kernel");
  return MCEDM_OK;
}

// C[m][n] (+)= sum_k opA(m,k) * opB(k,n); one thread per output element (embedding-MLP sized problems only)
__global__ void small_gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ Cm, int M,
                                  int N, int K, int lda, int ldb, int ldc, int tA, int tB, int acc) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < K; ++k) {
    const float av = tA ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k];
    const float bv = tB ? Bm[(size_t)n * ldb + k] : Bm[(size_t)k * ldb + n];
    s = fmaf(av, bv, s);
  }
  float* o = Cm + (size_t)m * ldc + n;
  *o = acc ? *o + s : s;
}

// the same with the K loop spread over the 64 lanes of a wave (one wave per output element, fixed-order butterfly): for
// the long contractions of the embedding backward (K = all affine rows: 3840 at ch = 128), where one thread per output
// is a chain of K dependent loads on 4096 threads
__global__ __launch_bounds__(256) void small_gemm_wave_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                              float* __restrict__ Cm, int M, int N, int K, int lda, int ldb,
                                                              int ldc, int tA, int tB, int acc) {
  const size_t i = blockIdx.x * (size_t)4 + (threadIdx.x >> 6);
  if (i >= (size_t)M * N) return;
  const int lane = threadIdx.x & 63;
  const int m = (int)(i / N), n = (int)(i % N);
  float s = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float av = tA ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k];
    const float bv = tB ? Bm[(size_t)n * ldb + k] : Bm[(size_t)k * ldb + n];
    s = fmaf(av, bv, s);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) {
    float* o = Cm + (size_t)m * ldc + n;
    *o = acc ? *o + s : s;
  }
}

int launch_small_gemm(const float* A, const float* Bm, float* Cm, int M, int N, int K, int lda, int ldb, int ldc,
                      int transA, int transB, int accumulate, hipStream_t s) {
  const size_t total = (size_t)M * N;
  if (K >= 256) {
    hipLaunchKernelGGL(small_gemm_wave_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, A, Bm, Cm, M, N, K, lda, ldb,
                       ldc, transA, transB, accumulate);
    MCEDM_LAUNCH_CHECK("small_gemm_wave_kernel");
    return MCEDM_OK;
  }
  hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, A, Bm, Cm, M, N, K, lda,
                     ldb, ldc, transA, transB, accumulate);
  MCEDM_LAUNCH_CHECK("small_gemm_kernel");
  return MCEDM_OK;
}

// =========================================================================================================
// Attention backward (adm_blocks.py:111-118 + autograd of the einsum at :178), recomputing the probabilities
// from q, k and the per-query log-sum-exp instead of storing the [T x T] weights.
//   P = softmax_k(q.k/8); a = P v;  dv = P^T da;  dP = da^T v;  dS = P o (dP - delta), delta_q = sum_c da a;
//   dq = dS k / 8;  dk = dS^T q / 8
// Three single-wave kernels per 32-token tile, all on fp32 MFMA with operands read straight from global memory:
//   (1) lse_q, delta_q          (2) dq (query on the lanes)          (3) dk, dv (key on the lanes)
// =========================================================================================================
#define MCEDM_KEY_OF(r, h) ((r & 3) + 8 * (r >> 2) + 4 * (h))

// KW waves per workgroup take the key tiles w, w + KW, ... of the SAME 32-query tile and their (running max, sum) pairs meet in
// LDS, merged by wave 0 in wave order (round 5: one wave per tile was 512 waves per launch at T = 256, B = 32, two heads -- half a
// wave per SIMD with every K load and every exp chain exposed: 107 us for 0.5 GFLOP)
template <int KW>
__global__ __launch_bounds__(64 * KW) void attn_bwd_stats_kernel(const float* __restrict__ qkv, const float* __restrict__ a,
                                                                 const float* __restrict__ da, float* __restrict__ lse, int T) {
  __shared__ float ml[KW][2][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * 32;
  const size_t bh = blockIdx.y;
  const float* Q = qkv + (bh * 3 + 0) * 64 * (size_t)T;
  const float* K = qkv + (bh * 3 + 1) * 64 * (size_t)T;
  const int q = q0 + l31, qc = q < T ? q : T - 1;
  float qreg[32];
  float delta = 0.f;
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const size_t o = (size_t)(2 * s + h) * T + qc;
    qreg[s] = Q[o] * 0.125f;
    if (wave == 0) delta += da[bh * 64 * (size_t)T + o] * a[bh * 64 * (size_t)T + o];
  }
  delta += __shfl_xor(delta, 32);
  float m = -INFINITY, l = 0.f;
  for (int k0 = 32 * wave; k0 < T; k0 += 32 * KW) {
    const int kk = k0 + l31, kc = kk < T ? kk : T - 1;
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 32; ++st)
      sc = __builtin_amdgcn_mfma_f32_32x32x2f32(K[(size_t)(2 * st + h) * T + kc], qreg[st], sc, 0, 0, 0);
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (k0 + MCEDM_KEY_OF(r, h) >= T) sc[r] = -INFINITY;
      mx = fmaxf(mx, sc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) rs += expf(sc[r] - mn);
    rs += __shfl_xor(rs, 32);
    l = l * expf(m - mn) + rs;
    m = mn;
  }
  if (KW > 1) {
    if (h == 0) { ml[wave][0][l31] = m; ml[wave][1][l31] = l; }
    __syncthreads();
    if (wave > 0) return;
    m = -INFINITY; l = 0.f;
#pragma unroll
    for (int w = 0; w < KW; ++w) {                          // wave order: a fixed order (a wave without key tiles holds (-inf, 0))
      const float mw = ml[w][0][l31], lw = ml[w][1][l31];
      const float mn = fmaxf(m, mw);
      if (mn > -INFINITY) l = l * expf(m - mn) + lw * expf(mw - mn);
      m = mn;
    }
  }
  if (q < T && h == 0) {
    lse[(bh * T + q) * 2] = m + logf(l);
    lse[(bh * T + q) * 2 + 1] = delta;
  }
}

// KW waves per workgroup split the streamed axis (keys for dq, queries for dk / dv: tiles w, w + KW, ...) of the SAME output
// tile; the partial outputs are plain sums and meet in LDS, added by wave 0 in wave order (fixed order, no atomics).  One
// wave per tile means B * heads * T / 32 waves per launch (512 at T = 256, B = 32, two heads): half a wave per SIMD with
// every load latency exposed.
template <int NREG>
__device__ __forceinline__ bool merge_wave_partials(float (&v)[NREG], float* part, int wave, int lane, int KW) {
  if (KW == 1) return true;
  if (wave > 0) {
    float* dst = part + (size_t)(wave - 1) * NREG * 64 + lane;
#pragma unroll
    for (int r = 0; r < NREG; ++r) dst[r * 64] = v[r];
  }
  __syncthreads();
  if (wave > 0) return false;
  for (int w = 1; w < KW; ++w) {
    const float* src = part + (size_t)(w - 1) * NREG * 64 + lane;
#pragma unroll
    for (int r = 0; r < NREG; ++r) v[r] += src[r * 64];
  }
  return true;
}

template <int KW>
__global__ __launch_bounds__(64 * KW) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ da,
                                                              const float* __restrict__ lse, float* __restrict__ dqkv, int T) {
  __shared__ float part[KW > 1 ? (KW - 1) * 32 * 64 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * 32;
  const size_t bh = blockIdx.y;
  const float* Q = qkv + (bh * 3 + 0) * 64 * (size_t)T;
  const float* K = qkv + (bh * 3 + 1) * 64 * (size_t)T;
  const float* V = qkv + (bh * 3 + 2) * 64 * (size_t)T;
  const float* dA = da + bh * 64 * (size_t)T;
  const int q = q0 + l31, qc = q < T ? q : T - 1;
  float qreg[32], dareg[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    qreg[s] = Q[(size_t)(2 * s + h) * T + qc] * 0.125f;
    dareg[s] = dA[(size_t)(2 * s + h) * T + qc];
  }
  const float L = lse[(bh * T + qc) * 2], delta = lse[(bh * T + qc) * 2 + 1];
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  for (int k0 = 32 * wave; k0 < T; k0 += 32 * KW) {
    const int kk = k0 + l31, kc = kk < T ? kk : T - 1;
    f32x16 sc, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
    const unsigned koff = 4u * (unsigned)(h * T + kc);        // scalar row base + one per-lane offset (see attention_split_kernel)
#pragma unroll
    for (int st = 0; st < 32; ++st) {
      const float kv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(K + (size_t)(2 * st) * T) + koff);
      const float vv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(V + (size_t)(2 * st) * T) + koff);
      sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qreg[st], sc, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, dareg[st], dp, 0, 0, 0);
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = k0 + MCEDM_KEY_OF(r, h) < T;
      const float pv = ok ? expf(sc[r] - L) : 0.f;
      ds[r] = pv * (dp[r] - delta);
    }
    // dq[c][q] += sum_key K[c][key] * dS[key][q]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* krow = K + (size_t)(32 * i + l31) * T;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + MCEDM_KEY_OF(r, h);
        o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[key < T ? key : T - 1], ds[r], o[i], 0, 0, 0);
      }
    }
  }
  float flat[32];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) flat[16 * i + r] = o[i][r];
  if (!merge_wave_partials<32>(flat, part, wave, lane, KW)) return;
  if (q < T) {
    float* dQ = dqkv + (bh * 3 + 0) * 64 * (size_t)T;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dQ[(size_t)(32 * i + MCEDM_KEY_OF(r, h)) * T + q] = flat[16 * i + r] * 0.125f;
  }
}

template <int KW>
__global__ __launch_bounds__(64 * KW) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ da,
                                                               const float* __restrict__ lse, float* __restrict__ dqkv, int T) {
  __shared__ float part[KW > 1 ? (KW - 1) * 64 * 64 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const int k0 = blockIdx.x * 32;
  const size_t bh = blockIdx.y;
  const float* Q = qkv + (bh * 3 + 0) * 64 * (size_t)T;
  const float* K = qkv + (bh * 3 + 1) * 64 * (size_t)T;
  const float* V = qkv + (bh * 3 + 2) * 64 * (size_t)T;
  const float* dA = da + bh * 64 * (size_t)T;
  const int key = k0 + l31, kc = key < T ? key : T - 1;
  float kreg[32], vreg[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    kreg[s] = K[(size_t)(2 * s + h) * T + kc];
    vreg[s] = V[(size_t)(2 * s + h) * T + kc];
  }
  f32x16 ok[2], ov[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ok[i][r] = 0.f; ov[i][r] = 0.f; }
  for (int q0 = 32 * wave; q0 < T; q0 += 32 * KW) {
    const int qq = q0 + l31, qc = qq < T ? qq : T - 1;
    // S[q][k] and dP[q][k]: rows = queries (registers), column = this lane's key
    f32x16 sc, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 32; ++st) {
      sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Q[(size_t)(2 * st + h) * T + qc] * 0.125f, kreg[st], sc, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dA[(size_t)(2 * st + h) * T + qc], vreg[st], dp, 0, 0, 0);
    }
    float pr[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = q0 + MCEDM_KEY_OF(r, h);
      const int qj = qi < T ? qi : T - 1;
      const float L = lse[(bh * T + qj) * 2], delta = lse[(bh * T + qj) * 2 + 1];
      pr[r] = qi < T ? expf(sc[r] - L) : 0.f;
      ds[r] = pr[r] * (dp[r] - delta);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* darow = dA + (size_t)(32 * i + l31) * T;
      const float* qrow = Q + (size_t)(32 * i + l31) * T;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = q0 + MCEDM_KEY_OF(r, h);
        const int qj = qi < T ? qi : T - 1;
        ov[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(darow[qj], pr[r], ov[i], 0, 0, 0);
        ok[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[qj], ds[r], ok[i], 0, 0, 0);
      }
    }
  }
  float flat[64];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { flat[16 * i + r] = ok[i][r]; flat[32 + 16 * i + r] = ov[i][r]; }
  if (!merge_wave_partials<64>(flat, part, wave, lane, KW)) return;
  if (key < T) {
    float* dK = dqkv + (bh * 3 + 1) * 64 * (size_t)T;
    float* dV = dqkv + (bh * 3 + 2) * 64 * (size_t)T;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const size_t c = 32 * i + MCEDM_KEY_OF(r, h);
        dK[c * T + key] = flat[16 * i + r] * 0.125f;
        dV[c * T + key] = flat[32 + 16 * i + r];
      }
  }
}

// ---- T % 128 == 0: the streamed operand staged in LDS, shared by the four waves of a workgroup (round 5) ---------------------
// The kernels above read every MFMA operand of the streamed axis straight from global memory: one dword load per lane and MFMA
// (96 per 32-key tile), and a CU accepts a vector-memory instruction every ~25 cycles whatever its width -- 18.9 TFLOP/s at
// T = 1024 (the reference network's attention at 32^2: 4.5 ms of its 25 ms training step).  Here a workgroup = four 32-token
// tiles of the kept axis (one per wave) x the SAME 32-token chunks of the streamed axis: a chunk is fetched once per workgroup
// with two 16-byte loads per thread and tensor, written to LDS in the two layouts its two uses need ([channel][token] for the
// first GEMMs, [token][channel] at pitch 65 for the second: both conflict-free), double-buffered, one barrier per chunk; the
// MFMA operands are LDS dwords.  Same sums in the same order per output element as the kernels above except that a wave now
// walks ALL chunks in ascending order instead of every fourth (no cross-wave merge): deterministic, batch independent.
constexpr int ATP = 65;                    // pitch of the transposed chunk images
constexpr float ATT_LOG2E = 1.4426950408889634f;

// The dq kernel also produces the softmax statistics the dk / dv kernel needs (no attn_bwd_stats_kernel launch on this path: its
// S = K^T Q product is the one computed here anyway; 164 of 496 us at T = 1024): the scores are normalised ONLINE -- running
// maximum m and sum l per query as in the forward kernel, dq accumulated unnormalised and rescaled when m moves, divided by l at
// the end -- and (m + log l, delta = sum_c dA A) go to `lse` for the second kernel.
__global__ __launch_bounds__(256) void attn_bwd_dq_lds_kernel(const float* __restrict__ qkv, const float* __restrict__ a,
                                                              const float* __restrict__ da, float* __restrict__ lse,
                                                              float* __restrict__ dqkv, int T, int xmap) {
  __shared__ float Kc[2][64 * 32], Vc[2][64 * 32], Kt[2][32 * ATP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  int bx_, by_;
  xcd_group_map(xmap, bx_, by_);
  const size_t bh = by_;
  const float* Q = qkv + (bh * 3 + 0) * 64 * (size_t)T;
  const float* K = qkv + (bh * 3 + 1) * 64 * (size_t)T;
  const float* V = qkv + (bh * 3 + 2) * 64 * (size_t)T;
  const float* dA = da + bh * 64 * (size_t)T;
  const int q = bx_ * 128 + 32 * wave + l31;
  float qreg[32], dareg[32];
  float delta = 0.f;
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    qreg[s] = Q[(size_t)(2 * s + h) * T + q] * (0.125f * ATT_LOG2E);      // scores in units of log 2: v_exp_f32 is 2^x
    dareg[s] = dA[(size_t)(2 * s + h) * T + q];
    delta += dareg[s] * a[bh * 64 * (size_t)T + (size_t)(2 * s + h) * T + q];
  }
  delta += __shfl_xor(delta, 32);
  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  // staging: thread -> (channel c, key quad kq) of the chunk, two of them per tensor
  f32x4 rk[2], rv[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + 256 * j, c = idx >> 3, kq = idx & 7;
      rk[j] = *reinterpret_cast<const f32x4*>(K + (size_t)c * T + k0 + 4 * kq);
      rv[j] = *reinterpret_cast<const f32x4*>(V + (size_t)c * T + k0 + 4 * kq);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + 256 * j, c = idx >> 3, kq = idx & 7;
      *reinterpret_cast<f32x4*>(&Kc[buf][c * 32 + 4 * kq]) = rk[j];
      *reinterpret_cast<f32x4*>(&Vc[buf][c * 32 + 4 * kq]) = rv[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) Kt[buf][(4 * kq + e) * ATP + c] = rk[j][e];
    }
  };
  const int nch = T / 32;
  fetch(0);
  commit(0);
  __syncthreads();
  for (int n = 0; n < nch; ++n) {
    const int buf = n & 1;
    if (n + 1 < nch) fetch(32 * (n + 1));
    f32x16 sc, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 32; ++st) {
      sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kc[buf][(2 * st + h) * 32 + l31], qreg[st], sc, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[buf][(2 * st + h) * 32 + l31], dareg[st], dp, 0, 0, 0);
    }
    // a query's 32 scores of this chunk sit in the lanes l31 and l31 + 32
    float mx = sc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);   // 0 on the first chunk (m = -inf)
    float ds[16];
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = __builtin_amdgcn_exp2f(sc[r] - mn);
      rs += pr;
      ds[r] = pr * (dp[r] - delta);
    }
    rs += __shfl_xor(rs, 32);
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[buf][MCEDM_KEY_OF(r, h) * ATP + 32 * i + l31], ds[r], o[i], 0, 0, 0);
    }
    if (n + 1 < nch) commit(buf ^ 1);
    __syncthreads();
  }
  float* dQ = dqkv + (bh * 3 + 0) * 64 * (size_t)T;
  const float sc_out = 0.125f / l;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dQ[(size_t)(32 * i + MCEDM_KEY_OF(r, h)) * T + q] = o[i][r] * sc_out;
  if (h == 0) {
    lse[(bh * T + q) * 2] = m + __builtin_amdgcn_logf(l);      // base 2, as attn_bwd_dkv_lds_kernel expects it (v_log_f32 is log2)
    lse[(bh * T + q) * 2 + 1] = delta;
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_lds_kernel(const float* __restrict__ qkv, const float* __restrict__ da,
                                                               const float* __restrict__ lse, float* __restrict__ dqkv, int T, int xmap) {
  __shared__ float Qc[2][64 * 32], Ac[2][64 * 32], Qt[2][32 * ATP], At[2][32 * ATP], Ls[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  int bx_, by_;
  xcd_group_map(xmap, bx_, by_);
  const size_t bh = by_;
  const float* Q = qkv + (bh * 3 + 0) * 64 * (size_t)T;
  const float* K = qkv + (bh * 3 + 1) * 64 * (size_t)T;
  const float* V = qkv + (bh * 3 + 2) * 64 * (size_t)T;
  const float* dA = da + bh * 64 * (size_t)T;
  const int key = bx_ * 128 + 32 * wave + l31;
  float kreg[32], vreg[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    kreg[s] = K[(size_t)(2 * s + h) * T + key];
    vreg[s] = V[(size_t)(2 * s + h) * T + key];
  }
  f32x16 ok[2], ov[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ok[i][r] = 0.f; ov[i][r] = 0.f; }
  f32x4 rq[2], ra[2];
  float rl = 0.f;
  auto fetch = [&](int q0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + 256 * j, c = idx >> 3, qq = idx & 7;
      rq[j] = *reinterpret_cast<const f32x4*>(Q + (size_t)c * T + q0 + 4 * qq);
      ra[j] = *reinterpret_cast<const f32x4*>(dA + (size_t)c * T + q0 + 4 * qq);
    }
    if (tid < 64) rl = lse[(bh * T + q0) * 2 + tid];            // (lse, delta) pairs of the chunk's 32 queries
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + 256 * j, c = idx >> 3, qq = idx & 7;
      *reinterpret_cast<f32x4*>(&Qc[buf][c * 32 + 4 * qq]) = rq[j] * (0.125f * ATT_LOG2E);
      *reinterpret_cast<f32x4*>(&Ac[buf][c * 32 + 4 * qq]) = ra[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) { Qt[buf][(4 * qq + e) * ATP + c] = rq[j][e]; At[buf][(4 * qq + e) * ATP + c] = ra[j][e]; }
    }
    if (tid < 64) Ls[buf][tid] = rl;
  };
  const int nch = T / 32;
  fetch(0);
  commit(0);
  __syncthreads();
  for (int n = 0; n < nch; ++n) {
    const int buf = n & 1;
    if (n + 1 < nch) fetch(32 * (n + 1));
    f32x16 sc, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 32; ++st) {
      sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Qc[buf][(2 * st + h) * 32 + l31], kreg[st], sc, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[buf][(2 * st + h) * 32 + l31], vreg[st], dp, 0, 0, 0);
    }
    float pr[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = MCEDM_KEY_OF(r, h);
      pr[r] = __builtin_amdgcn_exp2f(sc[r] - Ls[buf][2 * qi]);
      ds[r] = pr[r] * (dp[r] - Ls[buf][2 * qi + 1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = MCEDM_KEY_OF(r, h);
        ov[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(At[buf][qi * ATP + 32 * i + l31], pr[r], ov[i], 0, 0, 0);
        ok[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qt[buf][qi * ATP + 32 * i + l31], ds[r], ok[i], 0, 0, 0);
      }
    if (n + 1 < nch) commit(buf ^ 1);
    __syncthreads();
  }
  float* dK = dqkv + (bh * 3 + 1) * 64 * (size_t)T;
  float* dV = dqkv + (bh * 3 + 2) * 64 * (size_t)T;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const size_t c = 32 * i + MCEDM_KEY_OF(r, h);
      dK[c * T + key] = ok[i][r] * 0.125f;
      dV[c * T + key] = ov[i][r];
    }
}

int launch_attention_bwd(const float* qkv, const float* a, const float* da, float* dqkv, float* lse, int B, int heads,
                         int T, hipStream_t s) {
  MCEDM_REQUIRE(qkv && a && da && dqkv && lse, "attention_bwd: null pointer");
  MCEDM_REQUIRE(B > 0 && heads > 0 && T > 0 && (long long)B * heads <= 65535, "attention_bwd: bad shape");
  const dim3 grid(ceil_div(T, 32), B * heads);
  ProfScope ps("attention_bwd", 10.0 * B * heads * (double)T * T * 64, 4.0 * 8 * B * heads * 64.0 * T, s);
  // split factor: a function of T only (never of the batch size)
  static int lds_env = -1;                                 // MCEDM_ATTN_BWD_LDS=0: the direct-from-global kernels at every T (A/B runs)
  if (lds_env < 0) { const char* e = getenv("MCEDM_ATTN_BWD_LDS"); lds_env = e ? atoi(e) : 1; }
  const bool lds_path = lds_env && T % 128 == 0 && ((reinterpret_cast<size_t>(qkv) | reinterpret_cast<size_t>(da)) & 15) == 0;
  if (!lds_path) {
    if (T >= 128) hipLaunchKernelGGL(attn_bwd_stats_kernel<4>, grid, dim3(256), 0, s, qkv, a, da, lse, T);
    else if (T >= 64) hipLaunchKernelGGL(attn_bwd_stats_kernel<2>, grid, dim3(128), 0, s, qkv, a, da, lse, T);
    else hipLaunchKernelGGL(attn_bwd_stats_kernel<1>, grid, dim3(64), 0, s, qkv, a, da, lse, T);
    MCEDM_LAUNCH_CHECK("attn_bwd_stats_kernel");
  }
  if (lds_path) {
    const dim3 g4(T / 128, B * heads);
    static int xmap = -1;
    if (xmap < 0) { const char* e = getenv("MCEDM_ATTN_XCD"); xmap = e ? atoi(e) : 1; }
    hipLaunchKernelGGL(attn_bwd_dq_lds_kernel, g4, dim3(256), 0, s, qkv, a, da, lse, dqkv, T, xmap);
    MCEDM_LAUNCH_CHECK("attn_bwd_dq_lds_kernel");
    hipLaunchKernelGGL(attn_bwd_dkv_lds_kernel, g4, dim3(256), 0, s, qkv, da, lse, dqkv, T, xmap);
  } else if (T >= 128) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<4>, grid, dim3(256), 0, s, qkv, da, lse, dqkv, T);
    MCEDM_LAUNCH_CHECK("attn_bwd_dq_kernel");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<4>, grid, dim3(256), 0, s, qkv, da, lse, dqkv, T);
  } else if (T >= 64) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<2>, grid, dim3(128), 0, s, qkv, da, lse, dqkv, T);
    MCEDM_LAUNCH_CHECK("attn_bwd_dq_kernel");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<2>, grid, dim3(128), 0, s, qkv, da, lse, dqkv, T);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<1>, grid, dim3(64), 0, s, qkv, da, lse, dqkv, T);
    MCEDM_LAUNCH_CHECK("attn_bwd_dq_kernel");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<1>, grid, dim3(64), 0, s, qkv, da, lse, dqkv, T);
  }
  MCEDM_LAUNCH_CHECK("attn_bwd_dkv_kernel");
  return MCEDM_OK;
}

}  // namespace mcedm
