// conv_resident.hip -- the implicit-GEMM convolution for SMALL images (<= 32 x 32): the whole K extent of the
// workgroup's input tile is staged in LDS once, then the K loop is matrix instructions and a weight stream only.
//
// Why a second kernel: at <= 32 x 32 the grids of conv_mfma_kernel are 64-512 workgroups, one or two per CU, and its K loop
// (per 8 channels: issue loads -> MFMA -> transform + LDS write, two barriers) is a chain of latencies with nothing on
// the CU to cover it: the in-kernel phase counters (tools/conv_small_timeline.py) showed 45 % matrix / 55 % staging per
// iteration on the 8 x 8 tile, with 100 of 256 threads owning a tile element.  Here
//   * phase 0 requests every byte the workgroup needs by LDS-DMA (global -> LDS without registers): the raw halo tile of
//     ALL input channels ([Cin][PH+2][PW+2] floats: 25-100 KB), the raw input of a folded 1x1 projection, the first
//     weight slabs: scalar base + one per-lane offset, tight issue loops, no vector arithmetic;
//   * phase 1 applies GroupNorm/FiLM/SiLU and the zero padding IN PLACE in LDS (each thread the elements its own wave
//     fetched), while the accumulators are initialised with bias + residual;
//   * phase 2 walks K: the weight slab of chunk c+2 streams global -> LDS by DMA into a ring of three slabs while the
//     MFMAs of chunk c run; ONE barrier per chunk, a counted vmcnt wait, nothing else in the loop.
// The MFMA chunk loop, the accumulator initialisation (bias + residual), the epilogue and the fused GroupNorm
// statistics are the ones of conv_mfma_kernel, and K is summed in the same order (chunk, tap, channel pair; then the
// folded projection's channel pairs), so for one tile configuration the two kernels are BIT-IDENTICAL (tested); which one
// runs is decided by the image size and channel counts only, never by the batch size.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "conv_tile.hpp"
#include "prof.hpp"

namespace mcedm {

static constexpr int SKC = 16;      // channels per K chunk of the folded projection (= the packed 1x1 table's chunk)

// LDS-DMA issued from inline assembly, NOT through __builtin_amdgcn_global_load_lds: with the builtin in a kernel, hipcc's
// wait-count insertion treats every later LDS read as possibly ordered against a pending "flat" operation and degrades
// the counted waits of the MFMA loop (ds_read x2 -> s_waitcnt lgkmcnt(2) -> MFMA) to lgkmcnt(0) after every pair of
// k-steps.  The compiler does not see these loads at all, so: every consumer is ordered by an explicit s_waitcnt vmcnt
// (+ barrier) below, and compiler-generated vmcnt waits around them are only ever more conservative (the hardware
// counter includes them).  Address = scalar base + 32-bit per-lane byte offset: no per-load vector arithmetic at all.
// M0 = LDS byte address of lane 0's element; lane i writes at M0 + i * size.  (M0 is not otherwise used by these kernels;
// it cannot be named in the clobber list.)
__device__ __forceinline__ unsigned lds_addr(const float* l) { return (unsigned)reinterpret_cast<size_t>((lds_ptr_t)const_cast<float*>(l)); }
__device__ __forceinline__ const float* uniform_ptr(const float* q) {      // the wave-uniform pointer, in SGPRs
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void lds_dma16(const float* sbase, unsigned voff, unsigned m0v) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               : : "s"(__builtin_amdgcn_readfirstlane(m0v)), "v"(voff), "s"(uniform_ptr(sbase)) : "memory");
}
__device__ __forceinline__ void lds_dma4(const float* sbase, unsigned voff, unsigned m0v) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
               : : "s"(__builtin_amdgcn_readfirstlane(m0v)), "v"(voff), "s"(uniform_ptr(sbase)) : "memory");
}

// A weight slab = a run of whole rows of the packed table ([tap][ci_local] rows of MT floats at column m0), global -> LDS by
// DMA: float4 number i = tid + it * NT of the slab goes to LDS float4 i.  A slab is a whole K chunk (TAPS * KC rows), or,
// in SPLIT mode, the rows of taps [0, TSPLIT) / [TSPLIT, TAPS) of a chunk (half the LDS: two workgroups per CU).
// Every wave issues exactly ITU instructions per slab (a wave past the end of the slab repeats its previous step: same
// data to the same place), so the K loop can wait with a counted vmcnt for the slab it needs and leave the younger slab's
// DMA in flight.
template <class C, bool SPLIT>
struct SlabGeom {
  static constexpr int TSPLIT = 5;
  static constexpr int V4 = C::MT / 4;                              // float4 per row
  static constexpr int ROWS_IT = C::NT / V4;                        // slab rows per DMA step of the workgroup
  static constexpr int ROWS_MAX = (SPLIT ? TSPLIT : C::TAPS) * C::KC;
  static constexpr int SL = ROWS_MAX * C::MT;                       // floats of a ring slot
  static constexpr int ITU = (ROWS_MAX * V4 + C::NT - 1) / C::NT;   // DMA instructions per wave and slab
  static constexpr int NU = SPLIT ? 2 : 1;                          // slabs (= K-loop units) per chunk
  static_assert(C::NT % V4 == 0 && (ROWS_MAX * V4) % 64 == 0 && ROWS_MAX * V4 >= C::NT, "whole rows per step, whole waves");
  static_assert(!SPLIT || (C::TAPS == 9 && ((C::TAPS - TSPLIT) * C::KC * V4) % 64 == 0), "split slabs are for 3x3");
  static_assert(ITU <= (SPLIT ? C::TAPS - TSPLIT : C::TAPS), "one DMA step per tap of a unit");
};

template <class C, int RS, bool SPLIT>
__global__ __launch_bounds__(256, 2) void conv_resident_kernel(ConvArgs p, int tiles_x, int tiles_y, int mtiles, int nchunks,
                                                              int nsk, int coutp, int nslab) {
  static_assert(RS == RS_NONE || RS == RS_UP, "resampling modes of the resident kernel");
  static_assert(C::NT == 256 && C::NWAVE == 4 && C::CPI == 1, "four compute waves");
  static_assert(SKC * C::MT / 4 == C::NT, "the projection's weight slab is one float4 per thread (one DMA step)");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CPS = C::NT / C::PLANE;             // input channels staged per step of the workgroup
  constexpr int CPD = C::NT / C::NPIX;              // projection channels per step (a wave = 64 pixels of one channel)
  static_assert(CPS >= 1, "tile plane larger than the workgroup");
  typedef SlabGeom<C, SPLIT> SG;
  float* wl = lds;                                  // [nslab][SL] ring of weight slabs (nslab = 3, or 2 when LDS is short)
  float* xl = lds + nslab * SG::SL;                 // [nchunks * KC][PLANE] input tile: raw by DMA, then transformed in place
  float* sk = xl + nchunks * C::KC * C::PLANE;      // [nsk * SKC][NPIX] raw input of the folded projection
  Coef* cfl = reinterpret_cast<Coef*>(sk + nsk * SKC * C::NPIX);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN;
  const int wn = wave % C::WN;
  int bid = blockIdx.x;
  const int mt = bid % mtiles; bid /= mtiles;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * C::PH, x0 = tx * C::PW;
  const int m0 = mt * C::MT;
  const int Ca = p.Ca, Cin = p.Ca + p.Cb;
  const int nwu = SG::NU * nchunks;                 // K-loop units that stream 3x3 / 1x1 weights; then nsk projection units
  const int nunits = nwu + nsk;
  const int dist = nslab - 1;                       // slabs in flight ahead of the one being consumed

  if (p.dbg && tid == 0) p.dbg[blockIdx.x * 16 + 0] = __builtin_amdgcn_s_memrealtime();
#ifdef MCEDM_CONV_TIMELINE
  unsigned long long pseg[4] = {0, 0, 0, 0}, pprev = __builtin_amdgcn_s_memtime();
#define MCEDM_PSTAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pseg[k] += now_ - pprev; pprev = now_; }
#else
#define MCEDM_PSTAMP(k)
#endif
  // ---- phase 0: every byte this workgroup reads from HBM / L2 is requested here, by DMA, in a handful of tight loops
  const unsigned wvoff = 4u * (unsigned)((tid / SG::V4) * coutp + (tid % SG::V4) * 4);   // this lane's float4 inside a DMA step
  const unsigned wl_addr = lds_addr(wl);
  const float* wbase = p.wpk + m0;                              // + chunk * TAPS * KC * coutp
  const float* skbase = p.sk_wpk + m0;                          // + chunk * SKC * coutp (null + m0 is never dereferenced)
  const size_t wchunk = (size_t)C::TAPS * C::KC * coutp;
  // unit v of the K loop -> where its slab comes from; kind 2: conv weights, 1: projection weights, 0: past the end
  auto unit_src = [&](int v, const float*& base, int& nv4) -> int {
    if (v < nwu) {
      const int ch = SPLIT ? v >> 1 : v, part = SPLIT ? v & 1 : 0;
      base = wbase + (size_t)ch * wchunk + (part ? (size_t)SG::TSPLIT * C::KC * coutp : 0);
      nv4 = (SPLIT ? (part ? C::TAPS - SG::TSPLIT : SG::TSPLIT) : C::TAPS) * C::KC * SG::V4;
      return 2;
    }
    if (v < nunits) { base = skbase + (size_t)(v - nwu) * SKC * coutp; nv4 = C::NT; return 1; }
    return 0;
  };
  auto dma_step = [&](int it, const float* base, int nv4, int slab) {
    int st = it;
    if (it * C::NT + wave * 64 >= nv4) st = it - 1;                   // wave-uniform
    lds_dma16(base + (size_t)st * SG::ROWS_IT * coutp, wvoff, wl_addr + 4u * (unsigned)(slab * SG::SL) + 16u * (unsigned)(st * C::NT + wave * 64));
  };
  auto dma_unit = [&](int v, int slab) -> int {     // all steps at once; returns the DMA instructions per wave
    const float* base; int nv4;
    const int kind = unit_src(v, base, nv4);
    if (kind == 2) {
#pragma unroll
      for (int it = 0; it < SG::ITU; ++it) dma_step(it, base, nv4, slab);
      return SG::ITU;
    }
    if (kind == 1) { dma_step(0, base, nv4, slab); return 1; }
    return 0;
  };
  for (int j = 0; j < dist; ++j) dma_unit(j, j);

  // input tile: thread (chsub, pos) owns element pos of channel s * CPS + chsub in step s
  const bool active = tid < CPS * C::PLANE;
  const int chsub = active ? tid / C::PLANE : 0;
  const int pos = active ? tid - chsub * C::PLANE : 0;
  const size_t src_plane = (size_t)p.Hs * p.Ws;
  unsigned keep;
  {
    const int r = pos / C::PITCH, c = pos - r * C::PITCH;
    const int y = y0 + r - C::HALO, x = x0 + c - C::HALO;
    const bool inb = active && ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
    keep = inb ? 0xffffffffu : 0u;
    const int yc = inb ? y : 0, xc = inb ? x : 0;     // padding reads a clamped address and is masked in the transform pass
    const unsigned o = (RS == RS_UP) ? (unsigned)((yc >> 1) * p.Ws + (xc >> 1)) : (unsigned)(yc * p.Ws + xc);
    const unsigned boff = 4u * ((unsigned)chsub * (unsigned)src_plane + o);
    const int nsteps = Cin / CPS;                   // the launcher checks Ca % CPS == 0 and Cin % CPS == 0
    const unsigned xl_addr = lds_addr(xl) + 4u * (unsigned)(wave * 64);
    if (wave * 64 < CPS * C::PLANE) {               // waves that own no tile element issue nothing
      const float* pa = p.xa + (size_t)n * Ca * src_plane;
      const float* pb = p.xb + (size_t)n * p.Cb * src_plane - (size_t)Ca * src_plane;   // indexed by the concat channel
      for (int s = 0; s < nsteps; ++s) {
        const int c0 = s * CPS;
        const float* plane = (c0 < Ca ? pa : pb) + (size_t)c0 * src_plane;
        if (active) lds_dma4(plane, boff, xl_addr + 4u * (unsigned)(s * (CPS * C::PLANE)));
      }
    }
  }
  if (nsk) {                                        // raw input of the folded projection (interior pixels only)
    const int pix = tid % C::NPIX, csub = tid / C::NPIX;
    const int y = min(y0 + pix / C::PW, p.H - 1), x = min(x0 + pix % C::PW, p.W - 1);   // clamped: such pixels are never stored
    const size_t plane = (size_t)p.H * p.W;
    const unsigned poff = 4u * (unsigned)((size_t)csub * plane + (size_t)y * p.W + x);
    const unsigned sk_addr = lds_addr(sk) + 4u * (unsigned)(wave * 64);
    const float* pa = p.sk_xa + (size_t)n * p.sk_Ca * plane;
    const float* pb = p.sk_xb + (size_t)n * p.sk_Cb * plane - (size_t)p.sk_Ca * plane;
    const int steps = nsk * SKC / CPD;
    for (int s = 0; s < steps; ++s) {
      const int c0 = s * CPD;
      lds_dma4((c0 < p.sk_Ca ? pa : pb) + (size_t)c0 * plane, poff, sk_addr + 4u * (unsigned)(s * C::NT));
    }
  }
  MCEDM_PSTAMP(0)

  // ---- phase 1: accumulators = bias (+ residual); transform rows; then the in-place transform of the staged tile
  f32x16 acc[C::TM][C::TN];
  if (m0 + C::MT <= p.Cout) {
    if (!p.res) conv_init_acc<C, 0, true>(p, acc, n, m0, y0, x0, wm, wn, lane);
    else if (p.res_mode == RS_DOWN) conv_init_acc<C, 2, true>(p, acc, n, m0, y0, x0, wm, wn, lane);
    else conv_init_acc<C, 1, true>(p, acc, n, m0, y0, x0, wm, wn, lane);
  } else {
    if (!p.res) conv_init_acc<C, 0, false>(p, acc, n, m0, y0, x0, wm, wn, lane);
    else if (p.res_mode == RS_DOWN) conv_init_acc<C, 2, false>(p, acc, n, m0, y0, x0, wm, wn, lane);
    else conv_init_acc<C, 1, false>(p, acc, n, m0, y0, x0, wm, wn, lane);
  }
  stage_coef_rows<C::NT>(p, n, cfl, tid);
  // channels that pad the last chunk: zeros (their packed weights are zero too, but LDS garbage may be NaN)
  for (int e = Cin * C::PLANE + tid; e < nchunks * C::KC * C::PLANE; e += C::NT) xl[e] = 0.f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // every DMA of phase 0 has landed
  __syncthreads();                                  // transform rows visible to every wave
  MCEDM_PSTAMP(1)
  if (active) {
    // each thread transforms the elements its own wave's DMA wrote: read -> (x - mean) * scale + offset -> SiLU -> mask -> write
    const int nsteps = Cin / CPS;
    float* xt = xl + tid;
    const Coef* ct = cfl + chsub;
    auto pass = [&](auto act_tag) {
      constexpr bool ACT = decltype(act_tag)::value;
#pragma unroll 8
      for (int s = 0; s < nsteps; ++s) {
        const Coef cf = ct[s * CPS];
        float v = (xt[s * (CPS * C::PLANE)] - cf.mean) * cf.scale + cf.offset;
        if (ACT) v = silu_f(v);
        xt[s * (CPS * C::PLANE)] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & keep);
      }
    };
    if (p.act) pass(std::true_type{}); else pass(std::false_type{});
  }
  MCEDM_PSTAMP(2)

  int boffm[C::TN];
#pragma unroll
  for (int j = 0; j < C::TN; ++j) {
    const int pix = (wn * C::TN + j) * 32 + (lane & 31);
    boffm[j] = (lane >> 5) * C::PLANE + (pix / C::PW) * C::PITCH + (pix % C::PW);
  }
  const int aoff = (lane >> 5) * C::MT + wm * C::TM * 32 + (lane & 31);
  __syncthreads();
  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memtime(); }
  MCEDM_PSTAMP(3)
#ifdef MCEDM_CONV_TIMELINE
  if (p.dbg && tid == 0) { for (int k = 0; k < 3; ++k) p.dbg[blockIdx.x * 16 + 13 + k] = pseg[k]; p.dbg[blockIdx.x * 16 + 7] = pseg[3]; }
#endif

  // ---- phase 2: K loop.  Iteration i consumes slab i % nslab; the DMA of slab i + dist is issued before its MFMAs and
  // the wait at the end only covers slab i + 1 (issued one iteration earlier when dist == 2).
  int sc = 0, sn = dist % nslab;                    // ring positions of iteration i and i + dist
#ifdef MCEDM_CONV_TIMELINE   // per-phase cycle sums of wave 0 (tools/conv_small_timeline.py): DMA issue / MFMA / wait / barrier
  unsigned long long seg[4] = {0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define MCEDM_RSTAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); seg[k] += now_ - tprev; tprev = now_; }
#else
#define MCEDM_RSTAMP(k)
#endif
  int u = 0;                                        // unit being consumed
  auto finish_unit = [&](int issued) {
    __builtin_amdgcn_sched_barrier(0);
    MCEDM_RSTAMP(1)
    if (dist >= 2 && issued == SG::ITU) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(SG::ITU) : "memory");
    else if (dist >= 2 && issued == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    MCEDM_RSTAMP(2)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    MCEDM_RSTAMP(3)
    sc = sc + 1 == nslab ? 0 : sc + 1;
    sn = sn + 1 == nslab ? 0 : sn + 1;
    ++u;
  };
  // one weight unit: taps [T0, T1) of chunk c out of slab sc
  auto weight_unit = [&](int c, auto t0_tag, auto t1_tag) {
    constexpr int T0 = decltype(t0_tag)::value, T1 = decltype(t1_tag)::value;
    const float* wc = wl + sc * SG::SL;
    const float* nb; int nv4;
    const int kind = unit_src(u + dist, nb, nv4);
    if (kind == 2) {
      // the common case: the slab dist units ahead is a weight slab and its ITU DMA instructions go out one per tap, in
      // the shadow of the MFMAs (issued in a block in front of them they cost ~90 cycles each: the memory pipeline
      // accepts them slowly)
      const int slab = sn;
      MCEDM_RSTAMP(0)
      mfma_chunk<C, true, false, T0, T1>(xl + c * C::KC * C::PLANE, wc, acc, aoff, boffm, [&](int t) {
        if (t < SG::ITU) dma_step(t, nb, nv4, slab);
      });
      finish_unit(SG::ITU);
    } else {
      const int issued = kind == 1 ? (dma_step(0, nb, nv4, sn), 1) : 0;
      __builtin_amdgcn_sched_barrier(0);
      MCEDM_RSTAMP(0)
      mfma_chunk<C, true, false, T0, T1>(xl + c * C::KC * C::PLANE, wc, acc, aoff, boffm);
      finish_unit(issued);
    }
  };
  for (int c = 0; c < nchunks; ++c) {
    if constexpr (SPLIT) {
      weight_unit(c, std::integral_constant<int, 0>{}, std::integral_constant<int, SG::TSPLIT>{});
      weight_unit(c, std::integral_constant<int, SG::TSPLIT>{}, std::integral_constant<int, C::TAPS>{});
    } else {
      weight_unit(c, std::integral_constant<int, 0>{}, std::integral_constant<int, C::TAPS>{});
    }
  }
  for (int j = 0; j < nsk; ++j) {
    // one chunk of the folded projection: SKC channels at the centre tap out of the raw tile; fragments double-buffered
    const float* wc = wl + sc * SG::SL;
    const int issued = dma_unit(u + dist, sn);
    __builtin_amdgcn_sched_barrier(0);
    MCEDM_RSTAMP(0)
    const float* skc = sk + (size_t)j * SKC * C::NPIX + (lane >> 5) * C::NPIX + (lane & 31);
    float fa[2][C::TM], fb[2][C::TN];
#pragma unroll
    for (int ii = 0; ii < C::TM; ++ii) fa[0][ii] = wc[aoff + ii * 32];
#pragma unroll
    for (int jj = 0; jj < C::TN; ++jj) fb[0][jj] = skc[(wn * C::TN + jj) * 32];
#pragma unroll
    for (int kk = 0; kk < SKC / 2; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1, kn = kk + 1 < SKC / 2 ? kk + 1 : kk;
#pragma unroll
      for (int ii = 0; ii < C::TM; ++ii) fa[nxt][ii] = wc[aoff + 2 * kn * C::MT + ii * 32];
#pragma unroll
      for (int jj = 0; jj < C::TN; ++jj) fb[nxt][jj] = skc[2 * kn * C::NPIX + (wn * C::TN + jj) * 32];
#pragma unroll
      for (int ii = 0; ii < C::TM; ++ii)
#pragma unroll
        for (int jj = 0; jj < C::TN; ++jj) acc[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][ii], fb[cur][jj], acc[ii][jj], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, C::TM + C::TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, C::TM * C::TN, 0);
    }
    finish_unit(issued);
  }

#ifdef MCEDM_CONV_TIMELINE
  if (p.dbg && tid == 0) { for (int k = 0; k < 4; ++k) p.dbg[blockIdx.x * 16 + 8 + k] = seg[k]; p.dbg[blockIdx.x * 16 + 12] = 0; }
#endif
  // ---- epilogue (as conv_mfma_kernel): store + fused GroupNorm statistics
  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 6] = __builtin_amdgcn_s_memtime(); }
  const bool full = (m0 + C::MT <= p.Cout);
  float* red = wl;                                  // the slabs are dead after the last barrier
  if (p.gsum) {
    if (full) conv_epilogue<C, true, true>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
    else conv_epilogue<C, false, true>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
  } else {
    if (full) conv_epilogue<C, true, false>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
    else conv_epilogue<C, false, false>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
  }
  if (p.dbg && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p.dbg[blockIdx.x * 16 + 3] = __builtin_amdgcn_s_memrealtime();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.dbg[blockIdx.x * 16 + 4] = ((unsigned long long)xcc << 32) | hwid;
  }
  if (p.gsum) {
    __syncthreads();
    constexpr int NG = C::MT / 4;
    if (tid < NG) {
      float sum, m2;
      conv_stats_combine<C::WN>(red + tid * 3, NG * 3, sum, m2);
      const int g = m0 / 4 + tid;
      const int ngroups = (p.Cout + 3) / 4;
      const int ntiles = tiles_x * tiles_y;
      if (g < ngroups) {
        float* row = p.gsum + (((size_t)n * ntiles + ty * tiles_x + tx) * ngroups + g) * 2;
        row[0] = sum; row[1] = m2;
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// host side
static int g_resident = -1;      // -1: default (MCEDM_CONV_RESIDENT, else on); 0 / 1: forced by mcedm_op_set_conv_resident
void set_conv_resident(int enable) { g_resident = enable; }
static int resident_level() {     // 0: off, 1: the 8 x 8-pixel tile (<= 16 x 16 images), 2: also the 8 x 16 tile (~32 x 32 images)
  if (g_resident >= 0) return g_resident;
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_CONV_RESIDENT"); env = e ? atoi(e) : 1; }
  return env;
}

static constexpr int LDS_MAX = 160 * 1024;

template <class C, bool SPLIT>
static size_t resident_lds_bytes(const ConvArgs& a, int nslab) {
  const int Cin = a.Ca + a.Cb;
  const int nchunks = ceil_div(Cin, C::KC);
  const int nsk = a.sk_wpk ? ceil_div(a.sk_Ca + a.sk_Cb, SKC) : 0;
  return sizeof(float) * ((size_t)nslab * SlabGeom<C, SPLIT>::SL + (size_t)nchunks * C::KC * C::PLANE + (size_t)nsk * SKC * C::NPIX) + sizeof(Coef) * (size_t)Cin;
}

// 3 weight slabs (two in flight) when they fit into `budget` bytes of LDS, else 2; 0: this conv does not fit at all
template <class C, bool SPLIT>
static int resident_slabs(const ConvArgs& a, size_t budget = (size_t)LDS_MAX - 1024) {
  constexpr int CPS = C::NT / C::PLANE, CPD = C::NT / C::NPIX;
  const int Cin = a.Ca + a.Cb;
  if (!a.xa || a.Ca <= 0 || (a.Cb > 0 && !a.xb) || a.Ca % CPS || Cin % CPS) return 0;
  if (a.sk_wpk) {
    const int Csk = a.sk_Ca + a.sk_Cb;
    if (!a.sk_xa || a.sk_Ca <= 0 || (a.sk_Cb > 0 && !a.sk_xb) || Csk % SKC || a.sk_Ca % CPD) return 0;
  }
  for (int nslab = 3; nslab >= 2; --nslab)
    if (resident_lds_bytes<C, SPLIT>(a, nslab) <= budget) return nslab;
  return 0;
}

template <class C, int RS, bool SPLIT = false>
static int launch_resident(const ConvArgs& a_in, int nslab, hipStream_t stream) {
  ConvArgs a = a_in;
  a.dbg = conv_debug_buffer();
  { const int rc = conv_resolve_identity(a); if (rc != MCEDM_OK) return rc; }
  const int tiles_x = ceil_div(a.W, C::PW), tiles_y = ceil_div(a.H, C::PH);
  const int mtiles = ceil_div(a.Cout, C::MT);
  const int nchunks = ceil_div(a.Ca + a.Cb, C::KC);
  const int nsk = a.sk_wpk ? ceil_div(a.sk_Ca + a.sk_Cb, SKC) : 0;
  const long long blocks = (long long)a.B * tiles_x * tiles_y * mtiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) { set_error("conv grid out of range (%lld blocks)", blocks); return MCEDM_ERR_INVALID; }
  static bool attr_set[64] = {};
  int dev = 0;
  MCEDM_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { set_error("device index %d out of range", dev); return MCEDM_ERR_INVALID; }
  if (!attr_set[dev]) {
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)conv_resident_kernel<C, RS, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
    attr_set[dev] = true;
  }
  char name[96] = "";
  if (prof_enabled())
    snprintf(name, sizeof(name), "conv_resident_kernel<ConvCfg<%d, %d, %d, %d, %d, %d, %d, %d, %d>, %d, %s>", C::MT, C::PH, C::PW,
             C::WM, C::WN, C::TAPS, C::KC, C::NT, C::CPI, RS, SPLIT ? "true" : "false");
  const double px = (double)a.B * a.H * a.W;
  const double skc = a.sk_wpk ? (double)(a.sk_Ca + a.sk_Cb) : 0.0;
  const double flops = 2.0 * px * a.Cout * ((double)(a.Ca + a.Cb) * C::TAPS + skc);
  const double bytes = 4.0 * ((double)a.B * (a.Ca + a.Cb) * a.Hs * a.Ws + px * skc + px * a.Cout * (a.res ? 2 : 1) +
                              (double)a.Cout * ((a.Ca + a.Cb) * C::TAPS + skc));
  ProfScope ps(name, flops, bytes, stream);
  const unsigned lds_bytes = (unsigned)resident_lds_bytes<C, SPLIT>(a, nslab);
  hipLaunchKernelGGL((conv_resident_kernel<C, RS, SPLIT>), dim3((unsigned)blocks), dim3(256), lds_bytes, stream,
                     a, tiles_x, tiles_y, mtiles, nchunks, nsk, cout_padded(a.Cout), nslab);
  MCEDM_LAUNCH_CHECK("conv_resident_kernel");
  if (a.gsum_tiles) *a.gsum_tiles = SumTiles{tiles_x * tiles_y, tiles_x, C::PH, C::PW};
  return MCEDM_OK;
}

// Returns MCEDM_OK after launching, or -1 when this shape is not served here (the caller falls through to
// conv_mfma_kernel).  The choice depends on the image size, channel counts and resampling mode only.
int try_launch_conv_resident(const ConvArgs& a, int taps, hipStream_t stream) {
  if (resident_level() <= 0) return -1;
  if (a.resample != RS_NONE && !(a.resample == RS_UP && taps == 9)) return -1;
  if (cout_padded(a.Cout) % 64 != 0) return -1;
  const bool small = (long long)a.H * a.W <= 256 || a.W < 12;          // the 8 x 8-pixel tile of dispatch()
  if (taps == 9) {
    typedef ConvCfg<64, 8, 8, 2, 2, 9, 8> S;
    typedef ConvCfg<64, 8, 16, 1, 4, 9, 8> M;
    if (small) {
      const int ns = resident_slabs<S, false>(a);
      if (!ns) return -1;
      return a.resample == RS_UP ? launch_resident<S, RS_UP>(a, ns, stream) : launch_resident<S, RS_NONE>(a, ns, stream);
    }
    if (a.W >= 24 && (long long)a.H * a.W <= 1024 && cout_padded(a.Cout) % 128 != 0) {
      // ~32 x 32 images, 512 workgroups at B = 64.  Half-chunk weight slabs when that lets TWO workgroups share a CU
      // (<= 80 KB each: Cin <= 64 without a folded projection); otherwise one workgroup per CU in two rounds, which is
      // slower than conv_mfma_kernel today (70 vs 59 us), so opt-in (level 2)
      const int ns2 = resident_slabs<M, true>(a, 80 * 1024 - 512);
      if (ns2 == 3)
        return a.resample == RS_UP ? launch_resident<M, RS_UP, true>(a, ns2, stream) : launch_resident<M, RS_NONE, true>(a, ns2, stream);
      if (resident_level() < 2) return -1;
      const int ns = resident_slabs<M, false>(a);
      if (!ns) return -1;
      return a.resample == RS_UP ? launch_resident<M, RS_UP>(a, ns, stream) : launch_resident<M, RS_NONE>(a, ns, stream);
    }
    return -1;
  }
  typedef ConvCfg<64, 8, 8, 2, 2, 1, 16> P;
  if (small && !a.sk_wpk) {
    const int ns = resident_slabs<P, false>(a);
    if (ns) return launch_resident<P, RS_NONE>(a, ns, stream);
  }
  return -1;
}

}  // namespace mcedm
