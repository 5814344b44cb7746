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
//
// Kernels in this file: conv_mfma_kernel (the product path: one instantiation per tile configuration and
// resampling mode), conv8_mfma_kernel (experimental 8-wave variant of the largest tile, off by default,
// bit-identical), conv_small_cout_kernel (the ch -> out_channels output conv), pack_conv / pack_bias.
// What bounds them and what was tried is written up in DESIGN.md section 3; the short version: fp32 VALU work
// does not overlap fp32 MFMA work on gfx950, so every vector instruction here is paid in matrix time.
#include <atomic>
#include <cstdlib>

#include "common.hpp"
#include "conv_tile.hpp"
#include "pack.hpp"
#include "prof.hpp"

namespace mcedm {

__device__ Coef k_identity_coef = {0.f, 1.f, 0.f, 0.f};   // the table of a conv without an input transform


// Folded 1x1 projection (ConvArgs::sk_*): K chunks of 8 channels of cat(sk_xa, sk_xb) at the centre tap, accumulated onto
// the same tiles after the 3x3 loop.  The halo'd tile geometry of the 3x3 staging is reused as it is (the halo elements are
// loaded and never read: 1.3x the loads of a dedicated 1x1 tile, on a phase that is ~20 % of the launch).
template <class C>
__device__ __forceinline__ void conv_skip_phase(const ConvArgs& p, const TileGeom<C, RS_NONE>& G, float* xl, float* wl,
                                                f32x16 (&acc)[C::TM][C::TN], int n, int m0, int coutp, int aoff,
                                                const int (&boff)[C::TN], int tid, int wave) {
  static_assert(C::TAPS == 9, "the folded projection rides on a 3x3 conv");
  constexpr int SUB = TileGeom<C, RS_NONE>::SUB;
  constexpr int KS = 8;                                  // channels per iteration (xl holds KC = 8 planes)
  constexpr int V4 = C::MT / 4;                          // float4 per weight row
  constexpr int NV4 = KS * V4;                           // float4 of one iteration's weight rows: <= 256
  static_assert(NV4 <= C::NT && C::KC == KS, "one float4 of the skip slab per thread");
  const int Cin = p.sk_Ca + p.sk_Cb;
  const int iters = (Cin + KS - 1) / KS;
  const size_t plane = (size_t)p.H * p.W;
  const float* safe = p.sk_xa ? p.sk_xa : p.sk_xb;
  // the packed 1x1 table has ceil(Cin / 16) * 16 rows of coutp floats
  const __amdgpu_buffer_rsrc_t ws = make_rsrc(p.sk_wpk + m0, 4u * (unsigned)((size_t)((Cin + 15) / 16 * 16) * coutp - m0));
  const int wrow = tid / V4, wc4 = tid - wrow * V4;
  const unsigned wvoff = 4u * (unsigned)((tid < NV4 ? wrow : 0) * coutp + (tid < NV4 ? wc4 : 0) * 4);
  float raw[KS][SUB];
  f32x4 wv;
  auto issue = [&](int it) {
    const int c0 = it * KS;
#pragma unroll
    for (int cil = 0; cil < KS; ++cil) {
      const int ci = c0 + cil;
      const bool in_a = ci < p.sk_Ca;
      const float* src = in_a ? p.sk_xa : p.sk_xb;
      const int cc = in_a ? ci : ci - p.sk_Ca;
      const int CC = in_a ? p.sk_Ca : p.sk_Cb;
      const bool ok = (ci < Cin) && (src != nullptr);
      const float* pl = ok ? src + ((size_t)n * CC + cc) * plane : safe;
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub)
        raw[cil][sub] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pl) + G.boff[sub][0]);
    }
    wv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws, wvoff, 4u * (unsigned)(c0 * coutp), 0));
  };
  issue(0);
  for (int it = 0; it < iters; ++it) {
    const int c0 = it * KS;
#pragma unroll
    for (int cil = 0; cil < KS; ++cil) {
      const int ci = c0 + cil;
      const bool ok = (ci < Cin) && ((ci < p.sk_Ca ? p.sk_xa : p.sk_xb) != nullptr);
      const unsigned ckeep = ok ? 0xffffffffu : 0u;
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        const float v = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, raw[cil][sub]) & (G.keep[sub] & ckeep));
        if ((sub + 1) * C::NT <= C::PLANE || tid + sub * C::NT < C::PLANE) xl[cil * C::PLANE + tid + sub * C::NT] = v;
      }
    }
    if (tid < NV4) reinterpret_cast<f32x4*>(wl)[tid] = wv;
    __syncthreads();
    issue(it + 1 < iters ? it + 1 : it);                  // unconditional prefetch (the last one is dropped)
    if (wave < C::NWAVE) {
      constexpr int CENTRE = C::PITCH * C::HALO + C::HALO;
#pragma unroll
      for (int kk = 0; kk < KS / 2; ++kk) {
        float fa[C::TM], fb[C::TN];
#pragma unroll
        for (int i = 0; i < C::TM; ++i) fa[i] = wl[aoff + 2 * kk * C::MT + i * 32];
#pragma unroll
        for (int j = 0; j < C::TN; ++j) fb[j] = xl[boff[j] + CENTRE + 2 * kk * C::PLANE];
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
          for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
}

template <class C, int RS>
__device__ __forceinline__ void conv_body(const ConvArgs& p, float* xl, float* wl, Coef* cfl, int tiles_x, int tiles_y,
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

  int boff[C::TN];
#pragma unroll
  for (int j = 0; j < C::TN; ++j) {
    const int pix = (wn * C::TN + j) * 32 + (lane & 31);
    boff[j] = (lane >> 5) * (RS == RS_S2 ? 4 * C::PLANE : C::PLANE) + (pix / C::PW) * C::PITCH + (pix % C::PW);
  }
  const int aoff = (lane >> 5) * C::MT + wm * C::TM * 32 + (lane & 31);

  if (p.dbg && tid == 0) p.dbg[blockIdx.x * 16 + 0] = __builtin_amdgcn_s_memrealtime();
  TileGeom<C, RS> geom;
  make_geom<C, RS>(p, geom, y0, x0, tid);
  InputRegs<C, RS> xin;
  WeightRegs<C> win;
  WeightGeom<C> wgeom;
  make_wgeom<C>(p.wpk, wgeom, m0, coutp, (p.Ca + p.Cb + C::KC - 1) / C::KC, tid);
  load_weights<C>(wgeom, win, 0, coutp);
  // 1x1 convs on 16-byte addressable rows stage four pixels per load (conv_tile.hpp Vec4Geom); wave-uniform choice
  constexpr bool V4 = Vec4Geom<C>::OK && RS == RS_NONE;
  Vec4Geom<C> vgeom;
  bool v4 = false;
  if constexpr (V4) {
    v4 = (p.W & 3) == 0 && ((reinterpret_cast<size_t>(p.xa) | reinterpret_cast<size_t>(p.xb)) & 15) == 0;
    make_vec4_geom<C>(p, vgeom, y0, x0, tid);
  }
  auto load_in = [&](int c0) {
    if constexpr (V4) { if (v4) { load_input_v4<C, RS>(p, vgeom, xin, n, c0, wave); return; } }
    load_input<C, RS, false>(p, geom, xin, n, c0);
  };
  auto store_in = [&](int c0) {
    if constexpr (V4) { if (v4) { store_input_v4<C, RS>(p, vgeom, xl, xin, c0, tid, wave, cfl); return; } }
    store_input<C, RS>(p, geom, xl, xin, c0, tid, cfl);
  };
  load_in(0);
  stage_coef_rows<C::NT>(p, n, cfl, tid);
  // after the first chunk's loads are issued: the residual tile streams in behind them
  if (wave < C::NWAVE) {
    if (m0 + C::MT <= p.Cout) {
      if (!p.res) conv_init_acc<C, 0, true>(p, acc, n, m0, y0, x0, wm, wn, lane);
      else if (p.res_mode == RS_DOWN) conv_init_acc<C, 2, true>(p, acc, n, m0, y0, x0, wm, wn, lane);
      else conv_init_acc<C, 1, true>(p, acc, n, m0, y0, x0, wm, wn, lane);
    } else {
      if (!p.res) conv_init_acc<C, 0, false>(p, acc, n, m0, y0, x0, wm, wn, lane);
      else if (p.res_mode == RS_DOWN) conv_init_acc<C, 2, false>(p, acc, n, m0, y0, x0, wm, wn, lane);
      else conv_init_acc<C, 1, false>(p, acc, n, m0, y0, x0, wm, wn, lane);
    }
  }

  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memtime(); }
#ifdef MCEDM_CONV_TIMELINE   // per-phase cycle counters of the chunk loop (tools/conv_timeline.py); costs registers
  unsigned long long seg[5] = {0, 0, 0, 0, 0}, tprev = p.dbg ? __builtin_amdgcn_s_memtime() : 0;
#define MCEDM_STAMP(k) if (p.dbg) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); seg[k] += now_ - tprev; tprev = now_; }
#else
#define MCEDM_STAMP(k)
#endif
  __syncthreads();            // transform rows visible to every wave
  for (int ch = 0; ch < nchunks; ++ch) {
    store_weights<C>(wl, win, tid);
    store_in(ch * C::KCI);
    MCEDM_STAMP(0)
    __syncthreads();
    MCEDM_STAMP(1)
    {   // next chunk's global loads fly while this chunk's MFMAs run.  Unconditional (the last iteration re-reads
        // its own chunk and drops it): a guarded prefetch makes the loaded registers phis, which the compiler
        // resolves with a vmcnt(0) + register copies right here, in front of the MFMA loop.
      const int chn = ch + 1 < nchunks ? ch + 1 : ch;
      load_weights<C>(wgeom, win, chn, coutp);
      load_in(chn * C::KCI);
    }
    MCEDM_STAMP(2)
    if (wave < C::NWAVE) {
#pragma unroll
      for (int sc = 0; sc < C::CPI; ++sc)      // the packed chunks of this iteration, back to back
        mfma_chunk<C, true, RS == RS_S2>(xl + sc * C::KC * (RS == RS_S2 ? 4 * C::PLANE : C::PLANE), wl + sc * C::TAPS * C::KC * C::MT,
                                         acc, aoff, boff);
    }
    MCEDM_STAMP(3)
    __syncthreads();
    MCEDM_STAMP(4)
  }
#ifdef MCEDM_CONV_TIMELINE
  if (p.dbg && tid == 0) { for (int k = 0; k < 5; ++k) p.dbg[blockIdx.x * 16 + 8 + k] = seg[k]; }
#endif
  if constexpr (C::TAPS == 9 && RS == RS_NONE && C::CPI == 1) {
    if (p.sk_wpk) conv_skip_phase<C>(p, geom, xl, wl, acc, n, m0, coutp, aoff, boff, tid, wave);
  }
  // ---- epilogue: store NCHW (+ the fused GroupNorm partial sums); bias and residual are already in acc
  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 6] = __builtin_amdgcn_s_memtime(); }
  float* red = xl;            // the input tile is dead after the last chunk's closing barrier
  if (wave < C::NWAVE) conv_epilogue_any<C>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
  if (p.dbg && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p.dbg[blockIdx.x * 16 + 3] = __builtin_amdgcn_s_memrealtime();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.dbg[blockIdx.x * 16 + 4] = ((unsigned long long)xcc << 32) | hwid;
  }
  if (p.gsum) {               // wave-uniform: every wave of the workgroup reaches this barrier
    __syncthreads();
    conv_stats_store<C, C::WN>(p, red, n, m0, ty * tiles_x + tx, tiles_x * tiles_y, tid);
  }
}

// One kernel per resampling mode (0 none, 1 nearest 2x up, 2 2x2-mean down), each with its own register allocation:
// the down variant prefetches 4 source pixels per tile element and would otherwise set the budget (and the
// spills) of the other two.
template <class C, int RSK>
__global__ __launch_bounds__(256, RSK == RS_DOWN ? 2 : C::OCC) void conv_mfma_kernel(ConvArgs p, int tiles_x, int tiles_y,
                                                                                    int mtiles, int nchunks, int coutp) {
  __shared__ __attribute__((aligned(16))) float xl[C::XL];
  __shared__ __attribute__((aligned(16))) float wl[C::WL];
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];      // (Ca + Cb) transform rows
  conv_body<C, RSK>(p, xl, wl, reinterpret_cast<Coef*>(dyn_lds), tiles_x, tiles_y, mtiles, nchunks, coutp);
}

// Stride-2 3x3 conv (DDPM Downsample, models/ddim_blocks.py:85-104): the same body on a 4-phase input tile.
template <class C>
__global__ __launch_bounds__(256, 2) void conv_s2_mfma_kernel(ConvArgs p, int tiles_x, int tiles_y, int mtiles, int nchunks,
                                                             int coutp) {
  static_assert(C::TAPS == 9, "stride-2 conv is 3x3");
  __shared__ __attribute__((aligned(16))) float xl[4 * C::XL];
  __shared__ __attribute__((aligned(16))) float wl[C::WL];
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  conv_body<C, RS_S2>(p, xl, wl, reinterpret_cast<Coef*>(dyn_lds), tiles_x, tiles_y, mtiles, nchunks, coutp);
}


// ---- 8-wave variant for the large layers ------------------------------------------------------------
// One 512-thread workgroup per CU computes MT = 128 channels x (16 x 32) pixels: the work of two workgroups of the
// kernel above, sharing one weight slab.  The LDS slabs are double-buffered, so a chunk needs ONE barrier, and the two
// wave groups (waves 0-3 / 4-7, one of each per SIMD) run the chunk in opposite order:
//     group 0:  MFMA(chunk c)  ->  commit(chunk c+1) + issue loads(chunk c+2)
//     group 1:  commit(chunk c+1) + issue loads(chunk c+2)  ->  MFMA(chunk c)
// so on every SIMD one wave has matrix work while the other stages, by construction instead of by the luck of two
// independent workgroups drifting apart, and both groups meet at every barrier: no lag between them, no tail.
// Arithmetic (accumulation order per output element) and the GroupNorm partial sums (emitted per 8 x 32 sub-tile,
// reduced in the 4-wave order of the kernel above) are bit-identical to conv_mfma_kernel<128, 8, 32>, so choosing
// between the two by batch size does not break exact batch shardability.
template <class C>
__global__ __launch_bounds__(512, 1) void conv8_mfma_kernel(ConvArgs p, int tiles_x, int tiles_y, int mtiles, int nchunks,
                                                           int coutp) {
  static_assert(C::NT == 512 && C::WM == 1 && C::WN == 8 && C::PH == 16, "8-wave tile shape");
  extern __shared__ __attribute__((aligned(16))) float lds8[];
  constexpr int BUF = C::WL + C::XL;          // floats per buffer: weight slab, then input tile
  constexpr int NLOADS = C::KC * TileGeom<C, RS_NONE>::SUB;   // input loads a thread issues per chunk
  Coef* cfl = reinterpret_cast<Coef*>(lds8 + 2 * BUF);      // this sample's Ca + Cb transform rows
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;                        // WM == 1
  const int group = wave >> 2;

  int bid = blockIdx.x;
  const int mt = bid % mtiles; bid /= mtiles;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * C::PH, x0 = tx * C::PW;
  const int m0 = mt * C::MT;

  f32x16 acc[C::TM][C::TN];
  int boff[C::TN];
#pragma unroll
  for (int j = 0; j < C::TN; ++j) {
    const int pix = (wn * C::TN + j) * 32 + (lane & 31);
    boff[j] = (lane >> 5) * C::PLANE + (pix / C::PW) * C::PITCH + (pix % C::PW);
  }
  const int aoff = (lane >> 5) * C::MT + (lane & 31);

  if (p.dbg && tid == 0) p.dbg[blockIdx.x * 16 + 0] = __builtin_amdgcn_s_memrealtime();
#ifndef MCEDM_CONV_TIMELINE
  if (p.dbg && lane == 0) {     // which SIMD each wave landed on
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    p.dbg[blockIdx.x * 16 + 8 + wave] = hw;
  }
#endif
  TileGeom<C, RS_NONE> geom;
  make_geom<C, RS_NONE>(p, geom, y0, x0, tid);
  InputRegs<C, RS_NONE> xin;
  dma_weights<C>(p.wpk, lds8, 0, m0, coutp, tid);             // chunk 0 weights -> buffer 0
  load_input<C, RS_NONE, false>(p, geom, xin, n, 0);
  stage_coef_rows<C::NT>(p, n, cfl, tid);
  if (m0 + C::MT <= p.Cout) {
    if (!p.res) conv_init_acc<C, 0, true>(p, acc, n, m0, y0, x0, 0, wn, lane);
    else if (p.res_mode == RS_DOWN) conv_init_acc<C, 2, true>(p, acc, n, m0, y0, x0, 0, wn, lane);
    else conv_init_acc<C, 1, true>(p, acc, n, m0, y0, x0, 0, wn, lane);
  } else {
    if (!p.res) conv_init_acc<C, 0, false>(p, acc, n, m0, y0, x0, 0, wn, lane);
    else if (p.res_mode == RS_DOWN) conv_init_acc<C, 2, false>(p, acc, n, m0, y0, x0, 0, wn, lane);
    else conv_init_acc<C, 1, false>(p, acc, n, m0, y0, x0, 0, wn, lane);
  }
  __syncthreads();            // transform rows visible (also drains the chunk-0 weight DMA)
  // chunk 0 input -> buffer 0; chunk 1 input -> registers (clamped re-reads past the end are staged, never consumed)
  store_input<C, RS_NONE>(p, geom, lds8 + C::WL, xin, 0, tid, cfl);
  load_input<C, RS_NONE, false>(p, geom, xin, n, (1 < nchunks ? 1 : 0) * C::KC);
  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memtime(); }
  __syncthreads();

#ifdef MCEDM_CONV_TIMELINE   // per-phase cycle sums of wave 0 (group 0) and wave 4 (group 1): pre / MFMA / post / barrier
  unsigned long long seg8[4] = {0, 0, 0, 0}, tprev8 = __builtin_amdgcn_s_memtime();
#define MCEDM_STAMP8(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); seg8[k] += now_ - tprev8; tprev8 = now_; }
#else
#define MCEDM_STAMP8(k)
#endif
  for (int ch = 0; ch < nchunks; ++ch) {
    float* wl_c = lds8 + (ch & 1) * BUF;
    float* wl_n = lds8 + ((ch & 1) ^ 1) * BUF;
    const int c1 = ch + 1 < nchunks ? ch + 1 : ch;      // chunk held in the staging registers
    const int c2 = ch + 2 < nchunks ? ch + 2 : c1;      // chunk to fetch next
    // Per thread and iteration the VMEM order is: weight DMA (chunk c1 -> the other buffer), then NLOADS input loads
    // (chunk c2 -> registers).  The closing wait retires the DMA and leaves the input loads in flight.
    // (the MFMA code exists once; only the short staging block is duplicated around it: two full copies of the
    // iteration in an if/else made the register allocator spill the staging registers inside the loop)
    if (group == 0) {
      dma_weights<C>(p.wpk, wl_n, c1, m0, coutp, tid);
    } else {
      store_input<C, RS_NONE>(p, geom, wl_n + C::WL, xin, c1 * C::KC, tid, cfl);
      __builtin_amdgcn_sched_barrier(0);
      dma_weights<C>(p.wpk, wl_n, c1, m0, coutp, tid);
      __builtin_amdgcn_sched_barrier(0);
      load_input<C, RS_NONE, false>(p, geom, xin, n, c2 * C::KC);
    }
    __builtin_amdgcn_sched_barrier(0);
    MCEDM_STAMP8(0)
    mfma_chunk<C>(wl_c + C::WL, wl_c, acc, aoff, boff);
    __builtin_amdgcn_sched_barrier(0);
    MCEDM_STAMP8(1)
    if (group == 0) {
      store_input<C, RS_NONE>(p, geom, wl_n + C::WL, xin, c1 * C::KC, tid, cfl);
      load_input<C, RS_NONE, false>(p, geom, xin, n, c2 * C::KC);
    }
    __builtin_amdgcn_sched_barrier(0);
    MCEDM_STAMP8(2)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLOADS) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    MCEDM_STAMP8(3)
  }
#ifdef MCEDM_CONV_TIMELINE
  if (p.dbg && lane == 0 && (wave == 0 || wave == 4)) { for (int k = 0; k < 4; ++k) p.dbg[blockIdx.x * 16 + 8 + group * 4 + k] = seg8[k]; }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the dropped tail prefetch must not land in reused registers

  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 6] = __builtin_amdgcn_s_memtime(); }
  const bool full = (m0 + C::MT <= p.Cout);
  float* red = lds8;          // every slab is dead after the last chunk's barrier
  if (p.gsum) {                // (the launcher refuses pair records for this kernel)
    if (full) conv_epilogue<C, true, 4>(p, acc, n, m0, y0, x0, 0, wn, lane, red);
    else conv_epilogue<C, false, 4>(p, acc, n, m0, y0, x0, 0, wn, lane, red);
  } else {
    if (full) conv_epilogue<C, true, 0>(p, acc, n, m0, y0, x0, 0, wn, lane, red);
    else conv_epilogue<C, false, 0>(p, acc, n, m0, y0, x0, 0, wn, lane, red);
  }
  if (p.dbg && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p.dbg[blockIdx.x * 16 + 3] = __builtin_amdgcn_s_memrealtime();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.dbg[blockIdx.x * 16 + 4] = ((unsigned long long)xcc << 32) | hwid;
  }
  if (p.gsum) {
    __syncthreads();
    // two 8 x 32 sub-tile records (rows 0-7: waves 0-3, rows 8-15: waves 4-7), each the 4-wave sum of the 4-wave kernel
    constexpr int NG = C::MT / 4;
    if (tid < 2 * NG) {
      const int sub = tid / NG, e = tid - sub * NG;
      float sum, m2;
      conv_stats_combine<4>(red + (sub * 4 * NG + e) * 3, NG * 3, sum, m2);
      const int g = m0 / 4 + e;
      const int ngroups = (p.Cout + 3) / 4;
      const int tiles_y8 = (p.H + 7) / 8;
      const int ty8 = 2 * ty + sub;
      if (g < ngroups && ty8 < tiles_y8) {
        float* row = p.gsum + (((size_t)n * (tiles_x * tiles_y8) + ty8 * tiles_x + tx) * ngroups + g) * 2;
        row[0] = sum; row[1] = m2;
      }
    }
  }
}

// ---- output convolution (Cout <= 4) -------------------------------------------------------------------
// The network's last 3x3 conv maps ch -> out_channels (2): on the matrix path it would fill 2 of the 32 rows of an MFMA
// tile, so it is a direct fp32 kernel, and the one HBM-bound convolution of the network (268 MB in, 4 MB out at S128).
// Workgroup = 16 x 64 output pixels of one sample; each thread owns 4 consecutive pixels of a row and CO accumulators
// each, so one staged input value feeds up to 3 taps x 4 pixels from registers and a weight is read once per 4 pixels:
// 4.5x fewer LDS instructions than one pixel per thread (which ran at 1.6 TB/s, LDS-issue bound).  The transformed
// (GroupNorm + SiLU) halo tile is staged at a pitch of 72 floats with the tile's column -1 at LDS column 3, which puts
// every thread's middle four values on a 16-byte boundary: one ds_read_b128 + two ds_read_b32 per (channel, row),
// bank-conflict free.  fmaf chain over (channel, tap): deterministic, batch independent.
template <int CO>
__global__ __launch_bounds__(256) void conv_small_cout_kernel(ConvArgs p, const float* __restrict__ wpk,
                                                              const float* __restrict__ bias, int tiles_x, int tiles_y,
                                                              int nchunks, int coutp) {
  constexpr int KC = 8, TH = 16, TW = 64, ROWS = TH + 2, COLS = TW + 2, PITCH = 72, CPLANE = ROWS * PITCH;
  constexpr int NE = ROWS * COLS;                      // staged elements per channel
  constexpr int SUB = (NE + 255) / 256;                // ... per thread
  constexpr int NW = 9 * KC;
  __shared__ __attribute__((aligned(16))) float xl[KC * CPLANE];
  __shared__ __attribute__((aligned(16))) float wl[NW * CO];    // chunk weights [tap][ci_local][CO]: broadcast reads
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  Coef* cfl = reinterpret_cast<Coef*>(dyn_lds);
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int Cin = p.Ca + p.Cb;
  const size_t plane = (size_t)p.H * p.W;

  // per-thread staging slots, the same for every channel and chunk: element e = tid + k * 256 of the halo tile ->
  // clamped byte offset inside a channel plane, validity mask, LDS index inside a channel's tile
  unsigned soff[SUB], skeep[SUB];
  int slds[SUB];
#pragma unroll
  for (int k = 0; k < SUB; ++k) {
    const int e = tid + k * 256;
    const int r = e / COLS, c = e - r * COLS;
    const int y = y0 + r - 1, x = x0 + c - 1;
    const bool live = e < NE;
    const bool inb = live && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    soff[k] = inb ? 4u * (unsigned)(y * p.W + x) : 0u;
    skeep[k] = inb ? 0xffffffffu : 0u;
    slds[k] = live ? r * PITCH + c + 3 : -1;
  }
  const float* safe = p.xa ? p.xa : p.xb;
  float raw[KC][SUB];
  auto issue = [&](int ch) {
#pragma unroll
    for (int cil = 0; cil < KC; ++cil) {
      const int ci = ch * KC + cil;
      const bool in_a = ci < p.Ca;
      const float* src = in_a ? p.xa : p.xb;
      const int cc = in_a ? ci : ci - p.Ca;
      const int CC = in_a ? p.Ca : p.Cb;
      const bool ok = ci < Cin && src != nullptr;
      const float* pl = ok ? src + ((size_t)n * CC + cc) * plane : safe;
#pragma unroll
      for (int k = 0; k < SUB; ++k) raw[cil][k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pl) + soff[k]);
    }
  };
  issue(0);
  stage_coef_rows<256>(p, n, cfl, tid);
  const int wrow = tid < NW ? tid : NW - 1;        // threads 0..71 carry one (tap, channel) row of CO weights
  float wreg[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) wreg[co] = wpk[(size_t)wrow * coutp + co];
  const int py = tid >> 4, px4 = (tid & 15) * 4;
  float acc[CO][4];
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[co][q] = (bias && co < p.Cout) ? bias[co] : 0.f;
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll
    for (int cil = 0; cil < KC; ++cil) {
      const int ci = ch * KC + cil;
      const bool ok = ci < Cin && ((ci < p.Ca ? p.xa : p.xb) != nullptr);
      const Coef cf = cfl[ci < Cin ? ci : Cin - 1];
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        float v = apply_coef(raw[cil][k], cf, p.act);
        v = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & (ok ? skeep[k] : 0u));
        if (slds[k] >= 0) xl[cil * CPLANE + slds[k]] = v;
      }
    }
    if (tid < NW) {
#pragma unroll
      for (int co = 0; co < CO; ++co) wl[tid * CO + co] = wreg[co];
    }
    __syncthreads();
    const int chn = ch + 1 < nchunks ? ch + 1 : ch;
    issue(chn);
#pragma unroll
    for (int co = 0; co < CO; ++co) wreg[co] = wpk[((size_t)chn * NW + wrow) * coutp + co];
#pragma unroll
    for (int cil = 0; cil < KC; ++cil) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float* row = xl + cil * CPLANE + (py + r) * PITCH + px4 + 3;      // tile column px4 - 1
        float x6[6];
        x6[0] = row[0];
        const f32x4 mid = *reinterpret_cast<const f32x4*>(row + 1);             // 16-byte aligned
        x6[1] = mid[0]; x6[2] = mid[1]; x6[3] = mid[2]; x6[4] = mid[3];
        x6[5] = row[5];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          // output channels in pairs: one v_pk_fma_f32 per (pixel, channel pair)
          typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int c2 = 0; c2 < CO / 2; ++c2) {
            const f32x2 w2 = *reinterpret_cast<const f32x2*>(wl + ((r * 3 + b) * KC + cil) * CO + 2 * c2);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x2 a2 = {acc[2 * c2][q], acc[2 * c2 + 1][q]};
              const f32x2 xx = {x6[q + b], x6[q + b]};
              a2 = __builtin_elementwise_fma(xx, w2, a2);
              acc[2 * c2][q] = a2[0]; acc[2 * c2 + 1][q] = a2[1];
            }
          }
        }
      }
    }
    __syncthreads();
  }
  const int y = y0 + py;
  if (y < p.H) {
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      if (co >= p.Cout) continue;
      float* o = p.out + ((size_t)n * p.Cout + co) * plane + (size_t)y * p.W + x0 + px4;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (x0 + px4 + q < p.W) o[q] = acc[co][q];
    }
  }
}

// -------------------------------------------------------------------------------------------
// host side
int conv_kc_for(int taps) { return taps == 9 ? 8 : 16; }

size_t conv_packed_floats(int Cout, int Cin, int taps) {
  const int KC = conv_kc_for(taps);
  return (size_t)ceil_div(Cin, KC) * taps * KC * cout_padded(Cout);
}

// dst[((chunk*taps + tap)*KC + cil) * CoutP + co]; zero-padded in ci and co (pack.hpp pack_conv_value).
__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int taps,
                                 int KC, int coutp, int qkv_heads, int transpose_flip, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = pack_conv_value(w, i, Cout, Cin, taps, KC, coutp, qkv_heads, transpose_flip);
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

static unsigned long long* g_dbg = nullptr;
void set_conv_debug(unsigned long long* buf) { g_dbg = buf; }
unsigned long long* conv_debug_buffer() { return g_dbg; }

// A conv without an input transform reads one identity row with stride 0 (coef_rows = 0).
int conv_resolve_identity(ConvArgs& a) {
  a.coef_rows = 1;
  if (!a.coef) {
    static const Coef* ident[64] = {};     // per device
    int dev = 0;
    MCEDM_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device index %d out of range", dev); return MCEDM_ERR_INVALID; }
    if (!ident[dev]) MCEDM_HIP_TRY(hipGetSymbolAddress((void**)&ident[dev], HIP_SYMBOL(k_identity_coef)));
    a.coef = ident[dev]; a.coef_batch = 0; a.coef_rows = 0;
  }
  return MCEDM_OK;
}

template <class C>
static int launch_cfg(const ConvArgs& a_in, hipStream_t stream) {
  ConvArgs a = a_in;
  a.dbg = g_dbg;
  a.coef_rows = 1;
  if (!a.coef) {      // no input transform: one identity row, indexed with stride 0
    static const Coef* ident[64] = {};     // per device
    int dev = 0;
    MCEDM_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device index %d out of range", dev); return MCEDM_ERR_INVALID; }
    if (!ident[dev]) MCEDM_HIP_TRY(hipGetSymbolAddress((void**)&ident[dev], HIP_SYMBOL(k_identity_coef)));
    a.coef = ident[dev]; a.coef_batch = 0; a.coef_rows = 0;
  }
  const int tiles_x = ceil_div(a.W, C::PW), tiles_y = ceil_div(a.H, C::PH);
  const int mtiles = ceil_div(a.Cout, C::MT);
  const int nchunks = ceil_div(a.Ca + a.Cb, C::KCI);      // iterations of the K loop
  const long long blocks = (long long)a.B * tiles_x * tiles_y * mtiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) {
    set_error("conv grid out of range (%lld blocks)", blocks);
    return MCEDM_ERR_INVALID;
  }
  // algorithmic cost of this launch: 2*MAC flops; bytes = input read once + output written once + weights + residual
  char name[96] = "";
  if (prof_enabled())
    snprintf(name, sizeof(name), "conv_mfma_kernel<ConvCfg<%d, %d, %d, %d, %d, %d, %d, %d, %d>, %d>", C::MT, C::PH, C::PW,
             C::WM, C::WN, C::TAPS, C::KC, C::NT, C::CPI, (int)a.resample);   // = rocprofv3's name
  const double px = (double)a.B * a.H * a.W;
  const double skc = a.sk_wpk ? (double)(a.sk_Ca + a.sk_Cb) : 0.0;      // folded 1x1 projection: its own flops and input read
  const double flops = 2.0 * px * a.Cout * ((double)(a.Ca + a.Cb) * C::TAPS + skc);
  const double bytes = 4.0 * ((double)a.B * (a.Ca + a.Cb) * a.Hs * a.Ws + px * skc + px * a.Cout * (a.res ? 2 : 1) +
                              (double)a.Cout * ((a.Ca + a.Cb) * C::TAPS + skc));
  ProfScope ps(name, flops, bytes, stream);
  static int extra_lds = -1;     // diagnostics: MCEDM_CONV_EXTRA_LDS bytes of unused dynamic LDS lower the occupancy
  if (extra_lds < 0) { const char* e = getenv("MCEDM_CONV_EXTRA_LDS"); extra_lds = e ? atoi(e) : 0; }
  const unsigned rows_bytes = (unsigned)(a.Ca + a.Cb) * (unsigned)sizeof(Coef);      // transform rows in dynamic LDS
  MCEDM_REQUIRE((C::XL + C::WL) * sizeof(float) + rows_bytes <= 64 * 1024, "conv: %d input channels exceed the LDS row table",
                a.Ca + a.Cb);
  if (a.resample == RS_NONE)
    hipLaunchKernelGGL((conv_mfma_kernel<C, RS_NONE>), dim3((unsigned)blocks), dim3(256), rows_bytes + extra_lds, stream, a,
                       tiles_x, tiles_y, mtiles, nchunks, cout_padded(a.Cout));
  else if (a.resample == RS_UP)
    hipLaunchKernelGGL((conv_mfma_kernel<C, RS_UP>), dim3((unsigned)blocks), dim3(256), rows_bytes, stream, a, tiles_x, tiles_y,
                       mtiles, nchunks, cout_padded(a.Cout));
  else
    hipLaunchKernelGGL((conv_mfma_kernel<C, RS_DOWN>), dim3((unsigned)blocks), dim3(256), rows_bytes, stream, a, tiles_x, tiles_y,
                       mtiles, nchunks, cout_padded(a.Cout));
  MCEDM_LAUNCH_CHECK("conv_mfma_kernel");
  if (a.gsum_tiles) *a.gsum_tiles = SumTiles{tiles_x * tiles_y, tiles_x, C::PH, C::PW, a.gsum_rc == 2 ? 2 : 4};
  return MCEDM_OK;
}

template <class C>
static int launch_cfg_s2(const ConvArgs& a_in, hipStream_t stream) {
  ConvArgs a = a_in;
  a.dbg = nullptr;
  a.coef_rows = 1;
  if (!a.coef) {
    static const Coef* ident[64] = {};
    int dev = 0;
    MCEDM_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device index %d out of range", dev); return MCEDM_ERR_INVALID; }
    if (!ident[dev]) MCEDM_HIP_TRY(hipGetSymbolAddress((void**)&ident[dev], HIP_SYMBOL(k_identity_coef)));
    a.coef = ident[dev]; a.coef_batch = 0; a.coef_rows = 0;
  }
  const int tiles_x = ceil_div(a.W, C::PW), tiles_y = ceil_div(a.H, C::PH);
  const int mtiles = ceil_div(a.Cout, C::MT);
  const int nchunks = ceil_div(a.Ca + a.Cb, C::KCI);
  const long long blocks = (long long)a.B * tiles_x * tiles_y * mtiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) { set_error("conv grid out of range (%lld blocks)", blocks); return MCEDM_ERR_INVALID; }
  char name[96] = "";
  if (prof_enabled())
    snprintf(name, sizeof(name), "conv_s2_mfma_kernel<ConvCfg<%d, %d, %d, %d, %d, %d, %d, %d> >", C::MT, C::PH, C::PW, C::WM,
             C::WN, C::TAPS, C::KC, C::NT);
  const double px = (double)a.B * a.H * a.W;
  ProfScope ps(name, 2.0 * px * a.Cout * (double)(a.Ca + a.Cb) * 9,
               4.0 * ((double)a.B * (a.Ca + a.Cb) * a.Hs * a.Ws + px * a.Cout + (double)a.Cout * (a.Ca + a.Cb) * 9), stream);
  const unsigned rows_bytes = (unsigned)(a.Ca + a.Cb) * (unsigned)sizeof(Coef);
  MCEDM_REQUIRE((4 * C::XL + C::WL) * sizeof(float) + rows_bytes <= 64 * 1024, "conv: %d input channels exceed the LDS row table",
                a.Ca + a.Cb);
  hipLaunchKernelGGL((conv_s2_mfma_kernel<C>), dim3((unsigned)blocks), dim3(256), rows_bytes, stream, a, tiles_x, tiles_y,
                     mtiles, nchunks, cout_padded(a.Cout));
  MCEDM_LAUNCH_CHECK("conv_s2_mfma_kernel");
  if (a.gsum_tiles) *a.gsum_tiles = SumTiles{tiles_x * tiles_y, tiles_x, C::PH, C::PW, a.gsum_rc == 2 ? 2 : 4};
  return MCEDM_OK;
}

static int dispatch_s2(const ConvArgs& a, hipStream_t stream) {
  const int coutp = cout_padded(a.Cout);
  // pixel tile by image size only (as in dispatch: batch-independent bits); channel tile 64 when it divides
  if ((long long)a.H * a.W <= 256 || a.W < 12) {
    if (coutp % 64 == 0) return launch_cfg_s2<ConvCfg<64, 8, 8, 2, 2, 9, 8>>(a, stream);
    return launch_cfg_s2<ConvCfg<32, 8, 8, 1, 2, 9, 8>>(a, stream);
  }
  if (coutp % 64 != 0) return launch_cfg_s2<ConvCfg<32, 8, 8, 1, 2, 9, 8>>(a, stream);
  if (a.W >= 48) return launch_cfg_s2<ConvCfg<64, 8, 32, 1, 4, 9, 8>>(a, stream);
  return launch_cfg_s2<ConvCfg<64, 8, 16, 1, 4, 9, 8>>(a, stream);
}

typedef ConvCfg<128, 16, 32, 1, 8, 9, 8, 512> Conv8Cfg;
static int g_conv8 = -1;     // MCEDM_CONV8=1 selects the 8-wave kernel for the large layers (off: it only ties)

static int launch_conv8(const ConvArgs& a_in, hipStream_t stream) {
  typedef Conv8Cfg C;
  ConvArgs a = a_in;
  a.dbg = g_dbg;
  a.coef_rows = 1;
  if (!a.coef) {
    static const Coef* ident[64] = {};
    int dev = 0;
    MCEDM_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device index %d out of range", dev); return MCEDM_ERR_INVALID; }
    if (!ident[dev]) MCEDM_HIP_TRY(hipGetSymbolAddress((void**)&ident[dev], HIP_SYMBOL(k_identity_coef)));
    a.coef = ident[dev]; a.coef_batch = 0; a.coef_rows = 0;
  }
  const int tiles_x = ceil_div(a.W, C::PW), tiles_y = ceil_div(a.H, C::PH);
  const int mtiles = ceil_div(a.Cout, C::MT);
  const int nchunks = ceil_div(a.Ca + a.Cb, C::KC);
  const long long blocks = (long long)a.B * tiles_x * tiles_y * mtiles;
  constexpr int lds_max = 160 * 1024;
  const int lds_bytes = 2 * (C::WL + C::XL) * (int)sizeof(float) + (a.Ca + a.Cb) * (int)sizeof(Coef);
  static std::atomic<bool> attr_set[64];      // zero-initialised; a repeated set is benign, a data race is not
  {
    int dev = 0;
    MCEDM_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev].load(std::memory_order_acquire)) {
      MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)conv8_mfma_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      attr_set[dev].store(true, std::memory_order_release);
    }
  }
  char name[96] = "";
  if (prof_enabled()) snprintf(name, sizeof(name), "conv8_mfma_kernel<ConvCfg<128, 16, 32, 1, 8, 9, 8, 512> >");
  const double px = (double)a.B * a.H * a.W;
  const double flops = 2.0 * px * a.Cout * (double)(a.Ca + a.Cb) * C::TAPS;
  const double bytes = 4.0 * ((double)a.B * (a.Ca + a.Cb) * a.Hs * a.Ws + px * a.Cout * (a.res ? 2 : 1) +
                              (double)a.Cout * (a.Ca + a.Cb) * C::TAPS);
  ProfScope ps(name, flops, bytes, stream);
  hipLaunchKernelGGL((conv8_mfma_kernel<C>), dim3((unsigned)blocks), dim3(512), lds_bytes, stream, a, tiles_x, tiles_y, mtiles,
                     nchunks, cout_padded(a.Cout));
  MCEDM_LAUNCH_CHECK("conv8_mfma_kernel");
  if (a.gsum_tiles) *a.gsum_tiles = SumTiles{tiles_x * ceil_div(a.H, 8), tiles_x, 8, C::PW};     // records are per 8 x 32 sub-tile
  return MCEDM_OK;
}

void set_conv8(int enable) { g_conv8 = enable; }

static int launch_small_cout(const ConvArgs& a_in, hipStream_t stream) {
  struct C { enum { PH = 16, PW = 64, KC = 8 }; };      // the kernel's pixel tile and K chunk
  ConvArgs a = a_in;
  a.dbg = nullptr;
  a.coef_rows = 1;
  if (!a.coef) {
    static const Coef* ident[64] = {};
    int dev = 0;
    MCEDM_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device index %d out of range", dev); return MCEDM_ERR_INVALID; }
    if (!ident[dev]) MCEDM_HIP_TRY(hipGetSymbolAddress((void**)&ident[dev], HIP_SYMBOL(k_identity_coef)));
    a.coef = ident[dev]; a.coef_batch = 0; a.coef_rows = 0;
  }
  const int tiles_x = ceil_div(a.W, C::PW), tiles_y = ceil_div(a.H, C::PH);
  const int nchunks = ceil_div(a.Ca + a.Cb, C::KC);
  const long long blocks = (long long)a.B * tiles_x * tiles_y;
  if (blocks <= 0 || blocks > 0x7fffffffLL) { set_error("conv grid out of range (%lld blocks)", blocks); return MCEDM_ERR_INVALID; }
  const double px = (double)a.B * a.H * a.W;
  ProfScope ps("conv_small_cout_kernel", 2.0 * px * a.Cout * (double)(a.Ca + a.Cb) * 9,
               4.0 * ((double)a.B * (a.Ca + a.Cb) * a.Hs * a.Ws + px * a.Cout), stream);
  const unsigned rows_bytes = (unsigned)(a.Ca + a.Cb) * (unsigned)sizeof(Coef);
  MCEDM_REQUIRE(rows_bytes <= 20 * 1024, "conv: %d input channels exceed the LDS row table", a.Ca + a.Cb);
  if (a.Cout <= 2)
    hipLaunchKernelGGL(conv_small_cout_kernel<2>, dim3((unsigned)blocks), dim3(256), rows_bytes, stream, a, a.wpk, a.bias,
                       tiles_x, tiles_y, nchunks, cout_padded(a.Cout));
  else
    hipLaunchKernelGGL(conv_small_cout_kernel<4>, dim3((unsigned)blocks), dim3(256), rows_bytes, stream, a, a.wpk, a.bias,
                       tiles_x, tiles_y, nchunks, cout_padded(a.Cout));
  MCEDM_LAUNCH_CHECK("conv_small_cout_kernel");
  return MCEDM_OK;
}

static int g_force_mt = 0, g_force_ph = 0, g_force_pw = 0;   // test hook (mcedm_op_set_conv_tile); 0 = heuristic
void set_conv_tile_override(int mt, int ph, int pw) { g_force_mt = mt; g_force_ph = ph; g_force_pw = pw; }

template <int TAPS, int KC>
static int dispatch(const ConvArgs& a, hipStream_t stream) {
  const int coutp = cout_padded(a.Cout);
  // 8 x 8-pixel tiles (<= 16 x 16 images): 16 (3x3) / 64 (1x1) channels per K-loop iteration, see ConvCfg::CPI
  // measured (tools/conv_small_timeline.py): CPI = 2 / 4 made these launches SLOWER (8x8 3x3: 31 -> 38 us, 1x1: 20 -> 30 us,
  // 8x16: 67 -> 75 us): the K loop of the small tiles is not a chain of memory latencies but per-channel staging work
  // done by the few threads that own a tile element, so doubling the channels per iteration doubles that serial phase
  constexpr int SMALL_CPI = 1;
  constexpr int MID_CPI = 1;
#ifdef MCEDM_CONV_ONE   // compile-time probe builds: only the dominant configuration
  return launch_cfg<ConvCfg<128, 8, 32, 1, 4, 9, 8>>(a, stream);
#else
  if (g_force_mt) {
    const int id = g_force_mt * 10000 + g_force_ph * 100 + g_force_pw;
    MCEDM_REQUIRE(coutp % g_force_mt == 0, "conv: forced MT=%d does not divide padded Cout=%d", g_force_mt, coutp);
    switch (id) {
      case 1281632:
        MCEDM_REQUIRE(TAPS == 9 && a.resample == RS_NONE && a.Ca + a.Cb <= 2048 && !a.sk_wpk, "conv: the 8-wave kernel is 3x3, not resampled, without a folded projection");
        return launch_conv8(a, stream);
      case 1280832: return launch_cfg<ConvCfg<128, 8, 32, 1, 4, TAPS, KC>>(a, stream);
      case 1280816: return launch_cfg<ConvCfg<128, 8, 16, 1, 4, TAPS, KC>>(a, stream);
      case 640816: return launch_cfg<ConvCfg<64, 8, 16, 1, 4, TAPS, KC, 256, MID_CPI>>(a, stream);
      case 1280808: return launch_cfg<ConvCfg<128, 8, 8, 2, 2, TAPS, KC>>(a, stream);
      case 640832: return launch_cfg<ConvCfg<64, 8, 32, 1, 4, TAPS, KC>>(a, stream);
      case 320832: return launch_cfg<ConvCfg<32, 8, 32, 1, 4, TAPS, KC>>(a, stream);
      case 1281616: return launch_cfg<ConvCfg<128, 16, 16, 1, 4, TAPS, KC>>(a, stream);
      case 641616: return launch_cfg<ConvCfg<64, 16, 16, 1, 4, TAPS, KC>>(a, stream);
      case 321616: return launch_cfg<ConvCfg<32, 16, 16, 1, 4, TAPS, KC>>(a, stream);
      case 640808: return launch_cfg<ConvCfg<64, 8, 8, 2, 2, TAPS, KC, 256, SMALL_CPI>>(a, stream);
      case 320808: return launch_cfg<ConvCfg<32, 8, 8, 1, 2, TAPS, KC, 256, SMALL_CPI>>(a, stream);
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
    if (coutp % 64 == 0) return launch_cfg<ConvCfg<64, 8, 8, 2, 2, TAPS, KC, 256, SMALL_CPI>>(a, stream);
    return launch_cfg<ConvCfg<32, 8, 8, 1, 2, TAPS, KC, 256, SMALL_CPI>>(a, stream);
  }
  if (TAPS == 9 && a.W >= 24 && (long long)a.H * a.W <= 1024) {   // 3x3 on ~32x32 images: half-width tiles, 3 workgroups/CU
    if (coutp % 128 == 0 && blocks_for(128, 8, 16) >= want) return launch_cfg<ConvCfg<128, 8, 16, 1, 4, TAPS, KC>>(a, stream);
    if (coutp % 64 == 0) return launch_cfg<ConvCfg<64, 8, 16, 1, 4, TAPS, KC, 256, MID_CPI>>(a, stream);
    return launch_cfg<ConvCfg<32, 8, 32, 1, 4, TAPS, KC>>(a, stream);
  }
  if (a.W >= 24) {
    static int conv8_env = -1;
    if (conv8_env < 0) { const char* e = getenv("MCEDM_CONV8"); conv8_env = e ? atoi(e) : 0; }
    // experimental 8-wave kernel (needs about one workgroup per CU); results are bit-identical to <128, 8, 32>
    if (variant_choice(KV_CONV8, g_conv8, conv8_env) && !a.sk_wpk && a.gsum_rc != 2 && TAPS == 9 && a.resample == RS_NONE && coutp % 128 == 0 && a.Ca + a.Cb <= 2048 &&
        blocks_for(128, 16, 32) >= 224)
      return launch_conv8(a, stream);
    if (coutp % 128 == 0 && blocks_for(128, 8, 32) >= want) return launch_cfg<ConvCfg<128, 8, 32, 1, 4, TAPS, KC>>(a, stream);
    if (coutp % 64 == 0 && blocks_for(64, 8, 32) >= want) return launch_cfg<ConvCfg<64, 8, 32, 1, 4, TAPS, KC>>(a, stream);
    return launch_cfg<ConvCfg<32, 8, 32, 1, 4, TAPS, KC>>(a, stream);
  }
  if (coutp % 128 == 0 && blocks_for(128, 16, 16) >= want) return launch_cfg<ConvCfg<128, 16, 16, 1, 4, TAPS, KC>>(a, stream);
  if (coutp % 64 == 0 && blocks_for(64, 16, 16) >= want) return launch_cfg<ConvCfg<64, 16, 16, 1, 4, TAPS, KC>>(a, stream);
  return launch_cfg<ConvCfg<32, 16, 16, 1, 4, TAPS, KC>>(a, stream);
#endif
}

int launch_conv(const ConvArgs& a, int taps, hipStream_t stream) {
  MCEDM_REQUIRE(taps == 9 || taps == 1, "conv: taps must be 9 or 1 (got %d)", taps);
  MCEDM_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.Cout > 0 && a.Ca + a.Cb > 0, "conv: empty shape");
  MCEDM_REQUIRE(a.out && a.wpk, "conv: null output / weights");
  if (a.resample == RS_UP) MCEDM_REQUIRE(a.Hs * 2 == a.H && a.Ws * 2 == a.W, "conv: up-resample needs H = 2*Hs");
  else if (a.resample == RS_DOWN) MCEDM_REQUIRE(a.Hs == a.H * 2 && a.Ws == a.W * 2, "conv: down-resample needs Hs = 2*H");
  else if (a.resample == RS_S2) MCEDM_REQUIRE(taps == 9 && a.Hs == a.H * 2 && a.Ws == a.W * 2 && !a.res,
                                              "conv: the stride-2 conv is 3x3 on an even-sized source, without residual");
  else MCEDM_REQUIRE(a.Hs == a.H && a.Ws == a.W, "conv: source size mismatch");
  if (a.res && a.res_mode == RS_UP) MCEDM_REQUIRE(a.H % 2 == 0 && a.W % 2 == 0, "conv: up residual needs even size");
  if (a.sk_wpk) MCEDM_REQUIRE(taps == 9 && a.resample == RS_NONE && a.sk_Ca + a.sk_Cb > 0 && (a.sk_xa || a.sk_xb) && a.Cout > 4,
                              "conv: a folded 1x1 projection rides on an un-resampled 3x3 conv with more than 4 output channels");
  // 32-bit buffer offsets: one channel plane and the packed weight table must each stay below 4 GiB
  MCEDM_REQUIRE((unsigned long long)a.Hs * a.Ws * 4ull < (1ull << 32) &&
                (unsigned long long)cout_padded(a.Cout) * (a.Ca + a.Cb + 16) * taps * 4ull < (1ull << 32),
                "conv: plane or weight table exceeds the 4 GiB buffer range");
  if (a.res) {
    const unsigned long long rpix = a.res_mode == RS_DOWN ? 4ull * a.H * a.W : a.res_mode == RS_UP ? (a.H / 2ull) * (a.W / 2ull)
                                                                                                    : 1ull * a.H * a.W;
    MCEDM_REQUIRE(rpix * a.Cout * 4ull < (1ull << 32), "conv: one sample of the residual exceeds the 4 GiB buffer range");
  }
  // the output conv (ch -> 2): direct kernel, unless a tile is being forced by a test
  // (at <= 32 x 32 the padded matrix kernel is still faster: 43 vs 60 us at B = 64)
  if (taps == 9 && a.Cout <= 4 && a.resample == RS_NONE && !a.res && !a.gsum && !g_force_mt && (long long)a.H * a.W >= 4096)
    return launch_small_cout(a, stream);
  if (a.resample == RS_S2) return dispatch_s2(a, stream);
  // Winograd F(2x2, 3x3) kernel (conv_wino.hip) where the launch carries its weight table and the shape fits: chosen by shape
  // alone (never by the batch size), like every other tile choice here
  if (!g_force_mt && taps == 9 && a.wino && conv_wino_preferred(a) && conv_wino_applicable(a, taps)) return launch_conv_wino(a, stream);
  if (!g_force_mt && taps == 1) {     // un-transformed 1x1 convs at >= 32 x 32 (the decoder's skip projections): register-direct GEMM
    const int rc = try_launch_conv1x1_reg(a, taps, stream);
    if (rc != -1) return rc;
  }
  if (!g_force_mt) {     // small images: the input-resident kernel (conv_resident.hip), bit-identical per tile configuration
    const int rc = try_launch_conv_resident(a, taps, stream);
    if (rc != -1) return rc;
  }
  return taps == 9 ? dispatch<9, 8>(a, stream) : dispatch<1, 16>(a, stream);
}

}  // namespace mcedm
