// conv_resident.hip -- the implicit-GEMM convolution for SMALL images (<= 32 x 32): the whole K extent of the
// workgroup's input tile is staged in LDS once, then the K loop is matrix instructions and a weight stream only.
//
// Why a second kernel: at <= 32 x 32 the grids of conv_mfma_kernel are 64-512 workgroups, one or two per CU, and its K loop
// (per 8 channels: issue loads -> MFMA -> transform + LDS write, two barriers) is a chain of latencies with nothing on
// the CU to cover it: the in-kernel phase counters (tools/conv_small_timeline.py) showed 45 % matrix / 55 % staging per
// iteration on the 8 x 8 tile, with 100 of 256 threads owning a tile element.  Here
//   * phase 0 requests every byte the workgroup needs by LDS-DMA (global -> LDS without registers): the raw halo tile of
//     the resident input channels, laid out as whole 16-byte segments of the image rows ([channel][PH+2][PW+8] floats, one
//     lane-linear instruction per channel tile; 2x up-sampled / unaligned sources: one element per lane), and the first
//     weight slabs: scalar base + one per-lane offset, tight issue loops, no vector arithmetic;
//   * phase 1 applies GroupNorm/FiLM/SiLU and the zero padding IN PLACE in LDS (pairs of adjacent elements per thread),
//     while the accumulators are initialised with bias + residual and the transform rows are derived;
//   * phase 2 walks K in units (a chunk; half a chunk in SPLIT mode; KS chunks in K-split mode): the weight slab two units
//     ahead streams global -> LDS by DMA into a ring of three, one instruction per tap under the MFMAs; ONE barrier and a
//     counted vmcnt wait per unit, nothing else in the loop.  A folded 1x1 projection streams its raw input through the
//     same tile buffer afterwards.
//   * passes: when the input does not fit the LDS budget, pass_c channels are resident at a time.  At ~32 x 32 the budget
//     is 80 KB so that two workgroups share a CU (one computes while the other re-stages); at <= 8 x 8 the tile is
//     32 channels x 4 x 8 pixels with the K loop split over the four waves (4x the workgroups).
// The MFMA chunk loop, the accumulator initialisation (bias + residual), the epilogue and the fused GroupNorm
// statistics are the ones of conv_mfma_kernel, and K is summed in the same order (chunk, tap, channel pair; then the
// folded projection's channel pairs), so for one tile configuration the two kernels are BIT-IDENTICAL (tested; the K-split
// tile groups the sum per wave and is only close); which one runs is decided by the image size, channel counts and
// alignment only, never by the batch size.  DESIGN.md section 3 has the measurements and what did not work.
#include <atomic>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "conv_tile.hpp"
#include "prof.hpp"

namespace mcedm {

// the value, hidden from the optimiser: what is computed from it cannot be hoisted out of the enclosing loop (and kept
// in registers across the K loop)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

static constexpr int SKC = 16;      // channels per K chunk of the folded projection (= the packed 1x1 table's chunk)

// Tile configuration of the resident kernel: ConvCfg's members (the shared device code is written against them) with an
// LDS tile whose rows are whole 16-byte segments of the image rows: row r holds image columns x0 - 4 ... x0 + PW + 3
// (3x3; tile column -1 at layout column XL0 = 3), so a row is fetched as PITCH / 4 aligned float4 and the tile of one
// channel ([ROWS][PITCH] floats) is ONE lane-linear DMA instruction of ROWS * PITCH / 4 lanes.
// KS > 1 (K split): the workgroup's KS * WM * WN = 4 waves form KS groups that own the SAME output tile and take every
// KS-th K chunk each; the partial accumulators meet in LDS before the epilogue.  It makes the tile small enough
// (32 channels x 4 x 8 pixels) for 8 x 8 images to spread over 256 workgroups instead of 64, with a quarter of the
// dependent MFMA chain per wave.
// DB: the pass tile is double-buffered (two tiles of pass_c channels: the raw tile of pass p + 1 arrives by DMA under the
// MFMAs of pass p) and the passes are kept short, so the K loop starts after the first 16 channels have landed instead of
// after all of them; the K sum runs in the same order.
template <int MT_, int PH_, int PW_, int WM_, int WN_, int TAPS_, int KC_, int KS_ = 1, int DB_ = 0>
struct ResCfg {
  static constexpr int MT = MT_, PH = PH_, PW = PW_, WM = WM_, WN = WN_, TAPS = TAPS_, KC = KC_, KS = KS_;
  static constexpr bool DB = DB_ != 0;
  static constexpr int CPI = 1, KCI = KC_, NT = 256;
  static constexpr int HALO = (TAPS == 9) ? 1 : 0;
  static constexpr int XL0 = HALO ? 3 : 0;                  // layout column of tile column -HALO
  static constexpr int XM = HALO ? 4 : 0;                   // image columns to the left of x0 that a row holds
  static constexpr int PITCH = HALO ? PW + 8 : PW;
  static constexpr int ROWS = PH + 2 * HALO;
  static constexpr int PLANE = ROWS * PITCH;
  static constexpr int NPIX = PH * PW;
  static constexpr int TM = MT / WM / 32, TN = NPIX / WN / 32;
  static constexpr int XL = KCI * PLANE, WL = TAPS * KCI * MT;
  static constexpr int NWAVE = WM * WN * KS;
  static constexpr int OCC = 2;
  // wide staging: lanes (float4 row segments) per channel tile; channels per wave instruction, or (BIG tiles, used for
  // >= 64 x 64 images: the channel tile is larger than a wave) wave instructions per channel
  static constexpr int RSEG = PITCH / 4, LPC = ROWS * RSEG;
  static constexpr bool BIG = LPC > 64;
  static constexpr int CPW = BIG ? 1 : 64 / LPC, NIC = (LPC + 63) / 64;
  // transform pass on pairs of adjacent elements: pairs per row / channel; channels per workgroup step, or (BIG) pair
  // positions per thread
  static constexpr int PPR = HALO ? (PW + 4) / 2 : PW / 2, PC0 = HALO ? 2 : 0, PPC = ROWS * PPR;
  static constexpr int CPT = BIG ? 1 : NT / PPC, NPP = (PPC + NT - 1) / NT;
  static_assert(NWAVE == 4 && TM >= 1 && TN >= 1 && PITCH % 4 == 0 && CPW >= 1 && CPT >= 1 && (BIG || NPP == 1), "tile shape");
};

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
  static constexpr int CROWS = C::TAPS * C::KC;                     // table rows of one K chunk
  static constexpr int ROWS_MAX = SPLIT ? TSPLIT * C::KC : C::KS * CROWS;   // a slab: half a chunk, or the KS chunks of a unit
  static constexpr int SL = ROWS_MAX * C::MT;                       // floats of a ring slot
  static constexpr int ITU = (ROWS_MAX * V4 + C::NT - 1) / C::NT;   // DMA instructions per wave and slab
  static constexpr int NU = SPLIT ? 2 : 1;                          // slabs (= K-loop units) per chunk
  static_assert(C::NT % V4 == 0 && (ROWS_MAX * V4) % 64 == 0 && ROWS_MAX * V4 >= C::NT, "whole rows per step, whole waves");
  static_assert(!SPLIT || (C::KS == 1 && C::TAPS == 9 && ((C::TAPS - TSPLIT) * C::KC * V4) % 64 == 0), "split slabs are for 3x3");
  static_assert((CROWS * V4) % 64 == 0, "a chunk is whole waves of float4");
  // one DMA step per tap of a unit; a unit with fewer taps than steps (the 128-channel tile's second half: 4 taps, 5 steps)
  // issues the rest behind its last tap
  static_assert(ITU <= (SPLIT ? C::TAPS - TSPLIT : C::TAPS) + 1, "DMA steps per unit");
};

template <class C, int RS, bool SPLIT>
__global__ __launch_bounds__(256, 2) void conv_resident_kernel(ConvArgs p, int tiles_x, int tiles_y, int mtiles, int pass_c,
                                                              int coutp, int nslab, int wide, int stagger) {
  static_assert(RS == RS_NONE || RS == RS_UP, "resampling modes of the resident kernel");
  static_assert(C::NT == 256 && C::NWAVE == 4 && C::CPI == 1, "four compute waves");
  // (the 128-channel tile does not take folded projections: the launcher sends those convs to conv_mfma_kernel)
  static_assert(C::MT > 64 || (SKC * C::MT / 4 <= C::NT && (SKC * C::MT / 4) % 64 == 0), "the projection's weight slab is one DMA step of whole waves");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CPS = C::BIG ? 1 : C::NT / C::PLANE;   // narrow staging: input channels per step of the workgroup
  constexpr int CPD = C::BIG ? 1 : C::NT / C::NPIX;    // narrow staging of the projection's input: channels per step
  static_assert(C::BIG || (C::NT / C::PLANE >= 1), "tile plane larger than the workgroup");
  static_assert(!C::BIG || RS == RS_NONE, "big tiles stage by row segments only (un-resampled, aligned sources)");
  typedef SlabGeom<C, SPLIT> SG;
  float* wl = lds;                                  // [nslab][SL] ring of weight slabs (nslab = 3, or 2 when LDS is short)
  float* xl = lds + nslab * SG::SL;                 // [pass_c][PLANE] input tile of one pass: raw by DMA, then transformed in
                                                    // place; the projection's passes reuse it as [pass_c][NPIX], raw
  // BIG tiles hold TWO pass tiles: the raw tile of pass p + 1 streams in by DMA under the MFMAs of pass p (a big-image
  // launch is rounds of two workgroups per CU that run in lockstep, so nothing else would cover that latency)
  constexpr int NBUF = (C::BIG || C::DB) ? 2 : 1;
  Coef* cfl = reinterpret_cast<Coef*>(xl + NBUF * pass_c * C::PLANE);      // this sample's Ca + Cb transform rows
  int xoff = 0;                                     // float offset of the pass tile in use (0 / pass_c * PLANE)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave / (C::WM * C::WN);            // K group of this wave (0 when KS == 1)
  const int wm = (wave / C::WN) % C::WM;
  const int wn = wave % C::WN;
  int bid = blockIdx.x;
  const int mt = bid % mtiles; bid /= mtiles;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * C::PH, x0 = tx * C::PW;
  const int m0 = mt * C::MT;
  const int Ca = p.Ca, Cin = p.Ca + p.Cb;
  const int Csk = p.sk_wpk ? p.sk_Ca + p.sk_Cb : 0;
  const int nchunks = (Cin + C::KC - 1) / C::KC;    // K chunks of the conv over all passes
  const int nsk = Csk / SKC;                        // K chunks of the folded projection (the launcher checks Csk % SKC == 0)
  const int nwu = SG::NU * ((nchunks + C::KS - 1) / C::KS);   // K-loop units that stream conv weights (KS chunks each, or half a
                                                              // chunk in SPLIT mode); then nsk projection units
  const int nunits = nwu + nsk;
  const int dist = nslab - 1;                       // slabs in flight ahead of the one being consumed

  if (p.dbg && tid == 0) p.dbg[blockIdx.x * 16 + 0] = __builtin_amdgcn_s_memrealtime();
  // ---- weight stream: units of the K loop, numbered over all passes
  const unsigned wvoff = 4u * (unsigned)((tid / SG::V4) * coutp + (tid % SG::V4) * 4);   // this lane's float4 inside a DMA step
  const unsigned wl_addr = lds_addr(wl);
  const float* wbase = p.wpk + m0;                              // + chunk * TAPS * KC * coutp
  const float* skbase = p.sk_wpk + m0;                          // + chunk * SKC * coutp (null + m0 is never dereferenced)
  const size_t wchunk = (size_t)C::TAPS * C::KC * coutp;
  // unit v of the K loop -> where its slab comes from; kind 2: conv weights, 1: projection weights, 0: past the end
  auto unit_src = [&](int v, const float*& base, int& nv4) -> int {
    if (v < nwu) {
      if (SPLIT) {
        const int ch = v >> 1, part = v & 1;
        base = wbase + (size_t)ch * wchunk + (part ? (size_t)SG::TSPLIT * C::KC * coutp : 0);
        nv4 = (part ? C::TAPS - SG::TSPLIT : SG::TSPLIT) * C::KC * SG::V4;
      } else {
        base = wbase + (size_t)v * C::KS * wchunk;
        nv4 = min(C::KS, nchunks - v * C::KS) * SG::CROWS * SG::V4;   // the last unit may hold fewer chunks
      }
      return 2;
    }
    if (v < nunits) { base = skbase + (size_t)(v - nwu) * SKC * coutp; nv4 = SKC * SG::V4; return 1; }
    return 0;
  };
  // step `it` of a slab's DMA; a wave whose float4s of that step lie past the slab repeats its last valid step
  auto dma_step = [&](int it, const float* base, int nv4, int slab) {
    int st = it;
    while (st > 0 && st * C::NT + wave * 64 >= nv4) --st;             // wave-uniform
    lds_dma16(base + (size_t)st * SG::ROWS_IT * coutp, wvoff, wl_addr + 4u * (unsigned)(slab * SG::SL) + 16u * (unsigned)(st * C::NT + wave * 64));
  };
  auto dma_unit = [&](int v, int slab) -> int {     // all steps at once; returns the DMA instructions this wave issued
    const float* base; int nv4;
    const int kind = unit_src(v, base, nv4);
    if (kind == 2) {
#pragma unroll
      for (int it = 0; it < SG::ITU; ++it) dma_step(it, base, nv4, slab);
      return SG::ITU;
    }
    if (kind == 1 && wave * 64 < nv4) { dma_step(0, base, nv4, slab); return 1; }
    return 0;
  };
  for (int j = 0; j < dist; ++j) dma_unit(j, j);

  // ---- input tile of a pass: raw values by DMA into the [channel][ROWS][PITCH] layout.
  const size_t src_plane = (size_t)p.Hs * p.Ws;
  const float* pa = p.xa + (size_t)n * Ca * src_plane;
  const float* pb = p.xb + (size_t)n * p.Cb * src_plane - (size_t)Ca * src_plane;       // indexed by the concat channel
  const unsigned xl_base = lds_addr(xl);
  // wide (un-resampled source, rows 16-byte aligned): lane (csub, r, q) of a wave fetches float4 q of tile row r of
  // channel c + csub: CPW whole channel tiles per instruction, LDS image lane-linear
  unsigned wboff;
  bool wlane;
  {
    const int csub = lane / C::LPC, rem = lane - csub * C::LPC, r = rem / C::RSEG, q = rem - r * C::RSEG;
    const int y = y0 - C::HALO + r, xs = x0 - C::XM + 4 * q;
    wlane = lane < C::CPW * C::LPC;
    const bool inb = wlane && ((unsigned)y < (unsigned)p.H) && ((unsigned)xs < (unsigned)p.W);   // W % 4 == 0: all four or none
    wboff = 4u * ((unsigned)csub * (unsigned)src_plane + (inb ? (unsigned)(y * p.Ws + xs) : 0u));   // padding: clamped, masked later
  }
  // BIG tiles: part k of a channel tile = its float4 segments [64 k, 64 k + 64): lane -> (row, segment), computed where it
  // is used (once per pass: a few VALU instructions) instead of living in registers across the K loop, whose eight
  // accumulator blocks leave no room for them
  auto big_seg = [&](int k, unsigned& off) -> bool {
    const int sg = k * 64 + opaque(lane), r = sg / C::RSEG, q = sg - r * C::RSEG;
    const int y = y0 - C::HALO + r, xs = x0 - C::XM + 4 * q;
    const bool in_tile = sg < C::LPC;
    const bool inb = in_tile && ((unsigned)y < (unsigned)p.H) && ((unsigned)xs < (unsigned)p.W);
    off = 4u * (inb ? (unsigned)(y * p.Ws + xs) : 0u);
    return in_tile;
  };
  // narrow (2x up-sampled or unaligned source): thread pos of the workgroup fetches element pos of channel c (+ chsub)
  const bool nactive = !C::BIG && tid < CPS * C::PLANE;
  unsigned nboff = 0;
  if constexpr (!C::BIG) {
    const int chsub = nactive ? tid / C::PLANE : 0, pos = nactive ? tid - chsub * C::PLANE : 0;
    const int r = pos / C::PITCH, c = pos - r * C::PITCH;
    const int y = y0 - C::HALO + r, x = x0 - C::XM + c;
    const bool inb = nactive && ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
    const int yc = inb ? y : 0, xc = inb ? x : 0;
    const unsigned o = (RS == RS_UP) ? (unsigned)((yc >> 1) * p.Ws + (xc >> 1)) : (unsigned)(yc * p.Ws + xc);
    nboff = 4u * ((unsigned)chsub * (unsigned)src_plane + o);
  }
  // requests channels [cb, cb + pc) of the (concatenated, possibly up-sampled) input
  auto request_main = [&](int cb, int pc, int dst) {           // dst: float offset of the destination tile inside xl
    const int cend = min(cb + pc, Cin);
    const unsigned xl_dst = xl_base + 4u * (unsigned)dst;
    if constexpr (C::BIG) {                         // (channel, part) pairs round-robin over the waves; always wide
      for (int idx = wave; idx < (cend - cb) * C::NIC; idx += 4) {
        const int cl = idx / C::NIC, k = idx - cl * C::NIC, c0 = cb + cl;          // wave-uniform
        const float* plane = (c0 < Ca ? pa : pb) + (size_t)c0 * src_plane;
        unsigned off;
        if (big_seg(k, off)) lds_dma16(plane, off, xl_dst + 4u * (unsigned)(cl * C::PLANE + k * 256));
      }
    } else if (wide) {                              // the launcher checks Ca % CPW == 0 and Cin % CPW == 0
      for (int c0 = cb + wave * C::CPW; c0 < cend; c0 += 4 * C::CPW) {
        const float* plane = (c0 < Ca ? pa : pb) + (size_t)c0 * src_plane;
        if (wlane) lds_dma16(plane, wboff, xl_dst + 4u * (unsigned)((c0 - cb) * C::PLANE));
      }
    } else if (wave * 64 < CPS * C::PLANE) {        // Ca % CPS == 0 and Cin % CPS == 0; waves without an element issue nothing
      for (int c0 = cb; c0 < cend; c0 += CPS) {
        const float* plane = (c0 < Ca ? pa : pb) + (size_t)c0 * src_plane;
        if (nactive) lds_dma4(plane, nboff, xl_dst + 4u * (unsigned)((c0 - cb) * C::PLANE + wave * 64));
      }
    }
    // channels that pad the last chunk: zeros (their packed weights are zero too, but LDS garbage may be NaN)
    for (int e = (cend - cb) * C::PLANE + tid; e < pc * C::PLANE; e += C::NT) xl[dst + e] = 0.f;
  };
  // GroupNorm / FiLM / SiLU and the zero padding, in place, on PAIRS of horizontally adjacent elements (one 8-byte LDS
  // read and write per pair; the per-element arithmetic is apply_coef's).  Thread (tsub, r, pj)
  // owns layout columns PC0 + 2 pj, + 1 of row r of channel c + tsub; after the barrier that follows the DMA wait.
  const bool tactive = !C::BIG && tid < C::CPT * C::PPC;
  int tsub = 0, tpos = 0;
  unsigned keep0 = 0, keep1 = 0;
  if constexpr (!C::BIG) {
    tsub = tactive ? tid / C::PPC : 0;
    const int prem = tactive ? tid - tsub * C::PPC : 0, r = prem / C::PPR, col = C::PC0 + 2 * (prem - r * C::PPR);
    const int y = y0 - C::HALO + r, x = x0 - C::XM + col;
    const bool iny = (unsigned)y < (unsigned)p.H;
    keep0 = (iny && (unsigned)x < (unsigned)p.W) ? 0xffffffffu : 0u;
    keep1 = (iny && (unsigned)(x + 1) < (unsigned)p.W) ? 0xffffffffu : 0u;
    tpos = r * C::PITCH + col;
  }
  auto transform_main = [&](int cb, int pc) {
    const int cend = min(cb + pc, Cin);
    if constexpr (C::BIG) {
      // every channel has more pairs than the workgroup has threads: this thread owns pair positions tid + k NT (recomputed
      // per pass, see big_seg)
      int bpos[C::NPP];
      unsigned bkeep0[C::NPP], bkeep1[C::NPP];
      bool bact[C::NPP];
#pragma unroll
      for (int k = 0; k < C::NPP; ++k) {
        const int e = opaque(tid) + k * C::NT;
        bact[k] = e < C::PPC;
        const int prem = bact[k] ? e : 0, r = prem / C::PPR, col = C::PC0 + 2 * (prem - r * C::PPR);
        const int y = y0 - C::HALO + r, x = x0 - C::XM + col;
        const bool iny = (unsigned)y < (unsigned)p.H;
        bkeep0[k] = (iny && (unsigned)x < (unsigned)p.W) ? 0xffffffffu : 0u;
        bkeep1[k] = (iny && (unsigned)(x + 1) < (unsigned)p.W) ? 0xffffffffu : 0u;
        bpos[k] = r * C::PITCH + col;
      }
      auto pass = [&](auto act_tag) {
        constexpr bool ACT = decltype(act_tag)::value;
        for (int c = cb; c < cend; ++c) {
          const Coef cf = cfl[c];
#pragma unroll
          for (int k = 0; k < C::NPP; ++k) {
            if (bact[k]) {
              float* px = xl + xoff + (c - cb) * C::PLANE + bpos[k];
              float v0 = (px[0] - cf.mean) * cf.scale + cf.offset;
              float v1 = (px[1] - cf.mean) * cf.scale + cf.offset;
              if (ACT) { v0 = silu_f(v0); v1 = silu_f(v1); }
              px[0] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & bkeep0[k]);
              px[1] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v1) & bkeep1[k]);
            }
          }
        }
      };
      if (p.act) pass(std::true_type{}); else pass(std::false_type{});
      return;
    }
    if (tactive) {
      auto pass = [&](auto act_tag) {
        constexpr bool ACT = decltype(act_tag)::value;
#pragma unroll 4
        for (int c = cb + tsub; c < cend; c += C::CPT) {
          const Coef cf = cfl[c];
          float* px = xl + xoff + (c - cb) * C::PLANE + tpos;   // 8-byte aligned: one ds_read_b64 / ds_write_b64
          float v0 = (px[0] - cf.mean) * cf.scale + cf.offset;
          float v1 = (px[1] - cf.mean) * cf.scale + cf.offset;
          if (ACT) { v0 = silu_f(v0); v1 = silu_f(v1); }
          px[0] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & keep0);
          px[1] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v1) & keep1);
        }
      };
      if (p.act) pass(std::true_type{}); else pass(std::false_type{});
    }
  };

  int boffm[C::TN];
#pragma unroll
  for (int j = 0; j < C::TN; ++j) {
    const int pix = (wn * C::TN + j) * 32 + (lane & 31);
    boffm[j] = (lane >> 5) * C::PLANE + (pix / C::PW) * C::PITCH + (pix % C::PW) + C::XL0;
  }
  const int aoff = (lane >> 5) * C::MT + wm * C::TM * 32 + (lane & 31);

  // ---- K loop machinery.  Unit u consumes slab sc; the DMA of unit u + dist goes into slab sn, one instruction per tap
  // under the MFMAs, and the wait that ends a unit only covers slab u + 1 (issued one unit earlier when dist == 2).
  f32x16 acc[C::TM][C::TN];
  int u = 0, sc = 0, sn = dist % nslab;
  auto finish_unit = [&](int issued) {
    __builtin_amdgcn_sched_barrier(0);
    if (dist >= 2 && issued == SG::ITU) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(SG::ITU) : "memory");
    else if (dist >= 2 && issued == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    sc = sc + 1 == nslab ? 0 : sc + 1;
    sn = sn + 1 == nslab ? 0 : sn + 1;
    ++u;
  };
  // one weight unit: taps [T0, T1) of the pass's chunk(s) KS * ul + ks out of slab sc; cmax = chunks in this pass
  auto weight_unit = [&](int ul, int cmax, auto t0_tag, auto t1_tag) {
    constexpr int T0 = decltype(t0_tag)::value, T1 = decltype(t1_tag)::value;
    const int cl = C::KS * ul + ks;
    const float* wc = wl + sc * SG::SL + ks * (SG::CROWS * C::MT);
    const float* xc = xl + xoff + cl * C::KC * C::PLANE;
    const float* nb; int nv4;
    const int kind = unit_src(u + dist, nb, nv4);
    if constexpr (C::KS == 1) {
      // ONE copy of the MFMA chunk whatever the slab `dist` units ahead is: with the chunk duplicated in the arms of an
      // if / else the accumulators get a second set of registers for the phi at the join (D != C in the first MFMA of
      // every unit), which the 8-block tiles cannot afford (spills inside the K loop).  The tap hook branches instead
      // (wave-uniform, one scalar compare per tap).
      const bool stream = kind == 2;
      const int slab = sn;
      int issued = 0;
      if (kind == 1 && wave * 64 < nv4) { dma_step(0, nb, nv4, slab); issued = 1; }
      __builtin_amdgcn_sched_barrier(0);
      mfma_chunk<C, true, false, T0, T1>(xc, wc, acc, aoff, boffm, [&](int t) {
        if (stream && t < SG::ITU) dma_step(t, nb, nv4, slab);
        if constexpr (SG::ITU > T1 - T0) {
          if (stream && t == T1 - T0 - 1) dma_step(T1 - T0, nb, nv4, slab);
        }
      });
      finish_unit(stream ? SG::ITU : issued);
      return;
    }
    if (kind == 2) {
      // the common case: the slab dist units ahead is a weight slab and its ITU DMA instructions go out one per tap, in
      // the shadow of the MFMAs (issued in a block in front of them they cost ~90 cycles each: the memory pipeline
      // accepts them slowly)
      const int slab = sn;
      if (C::KS == 1 || cl < cmax) {
        mfma_chunk<C, true, false, T0, T1>(xc, wc, acc, aoff, boffm, [&](int t) {
          if (t < SG::ITU) dma_step(t, nb, nv4, slab);
        });
      } else {
#pragma unroll
        for (int t = 0; t < SG::ITU; ++t) dma_step(t, nb, nv4, slab);
      }
      finish_unit(SG::ITU);
    } else {
      const int issued = (kind == 1 && wave * 64 < nv4) ? (dma_step(0, nb, nv4, sn), 1) : 0;
      __builtin_amdgcn_sched_barrier(0);
      if (C::KS == 1 || cl < cmax) mfma_chunk<C, true, false, T0, T1>(xc, wc, acc, aoff, boffm);
      finish_unit(issued);
    }
  };

  // ---- passes over the conv's input channels: pass_c channels of the tile are resident at a time (all of them when
  // they fit).  While this workgroup re-stages, the other workgroup of the CU has the matrix pipe.
  const int cin_pad = nchunks * C::KC;
  // Stagger: when the conv needs more than one pass, every other "slot" of workgroups (blockIdx / stagger: the launcher
  // passes the CU count, so the two workgroups that share a CU differ) takes a short FIRST pass.  Its matrix phase then
  // starts while its CU partner is still staging, and from there on one of them computes while the other re-stages.
  // The K sum visits the chunks in the same order either way.
  const int first_c = (NBUF == 1 && stagger > 0 && cin_pad > pass_c && ((blockIdx.x / stagger) & 1)) ? 16 : pass_c;
  request_main(0, min(first_c, cin_pad), 0);
  // accumulators = bias (+ residual), transform rows: under the first pass's DMA
  if (ks == 0) {
    if (m0 + C::MT <= p.Cout) {
      if (!p.res) conv_init_acc<C, 0, true>(p, acc, n, m0, y0, x0, wm, wn, lane);
      else if (p.res_mode == RS_DOWN) conv_init_acc<C, 2, true>(p, acc, n, m0, y0, x0, wm, wn, lane);
      else conv_init_acc<C, 1, true>(p, acc, n, m0, y0, x0, wm, wn, lane);
    } else {
      if (!p.res) conv_init_acc<C, 0, false>(p, acc, n, m0, y0, x0, wm, wn, lane);
      else if (p.res_mode == RS_DOWN) conv_init_acc<C, 2, false>(p, acc, n, m0, y0, x0, wm, wn, lane);
      else conv_init_acc<C, 1, false>(p, acc, n, m0, y0, x0, wm, wn, lane);
    }
  } else {
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int j = 0; j < C::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  stage_coef_rows<C::NT>(p, n, cfl, tid);
  for (int cb = 0, pc = 0; cb < cin_pad; cb += pc) {
    pc = min(cb == 0 ? first_c : pass_c, cin_pad - cb);
    if (NBUF == 1 && cb > 0) request_main(cb, pc, 0);   // (the barrier that ended the previous unit freed xl)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's share of the tile (and every older DMA) has landed
    __syncthreads();                                // ... and everybody else's; first pass: transform rows visible
    transform_main(cb, pc);
    __syncthreads();
    if (NBUF == 2 && cb + pc < cin_pad)             // the next pass's raw tile -> the other buffer, under this pass's MFMAs
      request_main(cb + pc, min(pass_c, cin_pad - cb - pc), xoff ? 0 : pass_c * C::PLANE);
    if (cb == 0 && p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memtime(); }
    const int cmax = pc / C::KC;                    // chunks of this pass (the launcher makes passes whole units)
    for (int ul = 0; ul * C::KS < cmax; ++ul) {
      if constexpr (SPLIT) {
        weight_unit(ul, cmax, std::integral_constant<int, 0>{}, std::integral_constant<int, SG::TSPLIT>{});
        weight_unit(ul, cmax, std::integral_constant<int, SG::TSPLIT>{}, std::integral_constant<int, C::TAPS>{});
      } else {
        weight_unit(ul, cmax, std::integral_constant<int, 0>{}, std::integral_constant<int, C::TAPS>{});
      }
    }
    if (NBUF == 2) xoff = xoff ? 0 : pass_c * C::PLANE;
  }
  // ---- passes over the folded projection's input: raw interior pixels, [channel][NPIX]
  if (Csk) {
    const size_t plane = (size_t)p.H * p.W;
    const float* qa = p.sk_xa + (size_t)n * p.sk_Ca * plane;
    const float* qb = p.sk_xb + (size_t)n * p.sk_Cb * plane - (size_t)p.sk_Ca * plane;
    // narrow: one pixel per thread, CPD channels per step; wide: one float4 of a tile row per lane, CPQ channels per instruction
    constexpr int SEG = C::PW / 4, LPQ = C::PH * SEG, CPQ = LPQ > 64 ? 1 : 64 / LPQ, NIQ = (LPQ + 63) / 64;
    unsigned poff = 0, qoff = 0;
    unsigned bqoff[LPQ > 64 ? NIQ : 1];
    if constexpr (LPQ > 64) {                        // a channel's interior tile = NIQ wave instructions of float4 segments
      static_assert(LPQ % 64 == 0, "whole waves per channel tile");
#pragma unroll
      for (int k = 0; k < NIQ; ++k) {
        const int sg = k * 64 + lane, r = sg / SEG, q = sg - r * SEG;
        const int yq = min(y0 + r, p.H - 1), xq = min(x0 + 4 * q, p.W - 4);
        bqoff[k] = 4u * (unsigned)((size_t)yq * p.W + xq);
      }
    } else {
      const int pix = tid % C::NPIX, csub = tid / C::NPIX;
      const int y = min(y0 + pix / C::PW, p.H - 1), x = min(x0 + pix % C::PW, p.W - 1);   // clamped: such pixels are never stored
      poff = 4u * (unsigned)((size_t)csub * plane + (size_t)y * p.W + x);
      const int cq = lane / LPQ, rem = lane - cq * LPQ, r = rem / SEG, q = rem - r * SEG;
      const int yq = min(y0 + r, p.H - 1), xq = min(x0 + 4 * q, p.W - 4);
      qoff = 4u * (unsigned)((size_t)cq * plane + (size_t)yq * p.W + xq);
    }
    const int sk_pass = NBUF == 2 ? (NBUF * pass_c * C::PLANE / C::NPIX) / SKC * SKC : pass_c;   // >= SKC: pass_c * PLANE >= 16 * NPIX / 2
    for (int sb = 0; sb < Csk; sb += sk_pass) {
      const int pc = min(sk_pass, Csk - sb);        // multiple of SKC
      if constexpr (LPQ > 64) {
        for (int idx = wave; idx < pc * NIQ; idx += 4) {
          const int cl = idx / NIQ, k = idx - cl * NIQ, c0 = sb + cl;
#pragma unroll
          for (int kk = 0; kk < NIQ; ++kk)
            if (kk == k) lds_dma16((c0 < p.sk_Ca ? qa : qb) + (size_t)c0 * plane, bqoff[kk], xl_base + 4u * (unsigned)(cl * C::NPIX + kk * 256));
        }
      } else if (wide) {                            // the launcher checks sk_Ca % CPQ == 0
        for (int c0 = sb + wave * CPQ; c0 < sb + pc; c0 += 4 * CPQ)
          lds_dma16((c0 < p.sk_Ca ? qa : qb) + (size_t)c0 * plane, qoff, xl_base + 4u * (unsigned)((c0 - sb) * C::NPIX));
      } else {
        for (int c0 = sb; c0 < sb + pc; c0 += CPD)  // sk_Ca % CPD == 0
          lds_dma4((c0 < p.sk_Ca ? qa : qb) + (size_t)c0 * plane, poff, xl_base + 4u * (unsigned)((c0 - sb) * C::NPIX + wave * 64));
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      for (int j = 0; j < pc / SKC; ++j) {
        // one chunk of the projection: SKC channels at the centre tap out of the raw tile; fragments double-buffered
        const float* wc = wl + sc * SG::SL;
        const int issued = dma_unit(u + dist, sn);
        __builtin_amdgcn_sched_barrier(0);
        const float* skc = xl + (size_t)j * SKC * C::NPIX + (lane >> 5) * C::NPIX + (lane & 31);
        if (C::KS == 1 || ks == j % C::KS) {          // K-split tiles: the groups take the projection's chunks in turn
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
        }
        finish_unit(issued);
      }
    }
  }

  // ---- epilogue (as conv_mfma_kernel): store + fused GroupNorm statistics
  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 6] = __builtin_amdgcn_s_memtime(); }
  float* red = wl;                                  // the slabs are dead after the last barrier
  if constexpr (C::KS > 1) {
    // the K groups' partial sums meet in LDS (the input tile is dead): groups 1 ... KS-1 store, group 0 adds them in order
    float* part = xl;                               // [KS - 1][WM * WN waves][TM * TN * 16][64 lanes]
    constexpr int PER_WAVE = C::TM * C::TN * 16 * 64;
    if (ks > 0) {
      float* dst = part + ((ks - 1) * (C::WM * C::WN) + wm * C::WN + wn) * PER_WAVE + lane;
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((i * C::TN + j) * 16 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (ks == 0) {
      for (int g = 1; g < C::KS; ++g) {
        const float* src = part + ((g - 1) * (C::WM * C::WN) + wm * C::WN + wn) * PER_WAVE + lane;
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
          for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * C::TN + j) * 16 + r) * 64];
      }
    }
  }
  if (ks == 0) conv_epilogue_any<C>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
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
    conv_stats_store<C, C::WN>(p, red, n, m0, ty * tiles_x + tx, tiles_x * tiles_y, tid);
  }
}

// -------------------------------------------------------------------------------------------
// host side
static int g_resident = -1;      // -1: default (env MCEDM_CONV_RESIDENT, else on); 0 / 1: forced by mcedm_op_set_conv_resident
void set_conv_resident(int enable) { g_resident = enable; }
static int resident_level() {     // 0: off, > 0: on
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_CONV_RESIDENT"); env = e ? atoi(e) : 1; }
  return variant_choice(KV_CONV_RESIDENT, g_resident, env);
}

static constexpr int LDS_MAX = 160 * 1024;
// big-tile (>= 64 x 64 images) variants: MCEDM_RES_BIG=0 turns them off; MCEDM_RES_BIG_PASS = channels per pass (multiple of 8)
static int big_level() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_RES_BIG"); env = e ? atoi(e) : 1; }
  return env;
}
static int db_level() {          // MCEDM_RES_DB=1: double-buffered short passes at <= 32 x 32 (experiment)
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_RES_DB"); env = e ? atoi(e) : 0; }
  return env;
}
static int big128_level() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_RES_BIG128"); env = e ? atoi(e) : 0; }
  return env;
}
static int big_pass() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_RES_BIG_PASS"); env = (e && atoi(e) >= 8) ? atoi(e) / 8 * 8 : 1 << 20; }
  return env;
}

// Launch plan of one conv: channels resident per pass (0: this conv is not served here), weight slabs in the ring.
struct ResidentPlan { int pass_c = 0, nslab = 0, wide = 0; size_t lds = 0; };

template <class C, bool SPLIT>
static size_t resident_lds_bytes(const ConvArgs& a, int pass_c, int nslab) {
  return sizeof(float) * ((size_t)nslab * SlabGeom<C, SPLIT>::SL + (size_t)((C::BIG || C::DB) ? 2 : 1) * pass_c * C::PLANE) + sizeof(Coef) * (size_t)(a.Ca + a.Cb);
}

// Everything resident in one pass with a ring of 3 slabs when that fits `budget` bytes of LDS; otherwise as many channels
// per pass as fit (a multiple of 16: whole K chunks of the conv and of the projection), at least min_pass.
template <class C, bool SPLIT>
static ResidentPlan resident_plan(const ConvArgs& a, size_t budget, int min_pass) {
  constexpr int LPQ = C::PH * (C::PW / 4);
  constexpr int CPS = C::BIG ? 1 : C::NT / C::PLANE, CPD = C::BIG ? 1 : C::NT / C::NPIX, CPQ = LPQ > 64 ? 1 : 64 / LPQ;
  ResidentPlan r;
  const int Cin = a.Ca + a.Cb, Csk = a.sk_wpk ? a.sk_Ca + a.sk_Cb : 0;
  if (!a.xa || a.Ca <= 0 || (a.Cb > 0 && !a.xb)) return r;
  if (Csk && (!a.sk_xa || a.sk_Ca <= 0 || (a.sk_Cb > 0 && !a.sk_xb) || Csk % SKC)) return r;
  // wide staging (float4 row segments by DMA): un-resampled source whose rows and planes are 16-byte aligned
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  r.wide = a.resample == RS_NONE && a.Ws % 4 == 0 && a.W % 4 == 0 && al16(a.xa) && al16(a.xb) && a.Ca % C::CPW == 0 && Cin % C::CPW == 0 &&
           (!Csk || (al16(a.sk_xa) && al16(a.sk_xb) && a.sk_Ca % CPQ == 0));
  if (!r.wide && (C::BIG || a.Ca % CPS || Cin % CPS || (Csk && a.sk_Ca % CPD))) { r.wide = 0; return r; }   // big tiles: row segments only
  const int need = std::max(ceil_div(Cin, C::KC) * C::KC, Csk);      // channels of the longest input (padded to chunks)
  // a pass is whole K units of the conv and of the projection; big tiles: whole chunks (their projection passes are sized
  // separately in the kernel: both pass buffers together hold >= SKC channels of interior pixels)
  constexpr int G = C::BIG ? C::KC : (C::KS * C::KC > 16 ? C::KS * C::KC : 16);
  static_assert(!C::BIG || 2 * C::KC * C::PLANE >= SKC * C::NPIX, "projection chunk fits the two pass buffers");
  const int all = ceil_div(need, G) * G;
  for (int nslab = 3; nslab >= 2; --nslab) {
    const size_t fixed = resident_lds_bytes<C, SPLIT>(a, 0, nslab);
    if (fixed >= budget) continue;
    int pc = (int)((budget - fixed) / (sizeof(float) * C::PLANE * ((C::BIG || C::DB) ? 2 : 1))) / G * G;
    if (pc > all) pc = all;
    if (pc >= std::min(min_pass, all)) { r.pass_c = pc; r.nslab = nslab; r.lds = resident_lds_bytes<C, SPLIT>(a, pc, nslab); return r; }
  }
  return r;
}

template <class C, int RS, bool SPLIT = false>
static int launch_resident(const ConvArgs& a_in, const ResidentPlan& plan, hipStream_t stream) {
  ConvArgs a = a_in;
  a.dbg = conv_debug_buffer();
  { const int rc = conv_resolve_identity(a); if (rc != MCEDM_OK) return rc; }
  const int tiles_x = ceil_div(a.W, C::PW), tiles_y = ceil_div(a.H, C::PH);
  const int mtiles = ceil_div(a.Cout, C::MT);
  const long long blocks = (long long)a.B * tiles_x * tiles_y * mtiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) { set_error("conv grid out of range (%lld blocks)", blocks); return MCEDM_ERR_INVALID; }
  static std::atomic<bool> attr_set[64];      // zero-initialised; a repeated set is benign, a data race is not
  static std::atomic<int> ncu[64];
  int dev = 0;
  MCEDM_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { set_error("device index %d out of range", dev); return MCEDM_ERR_INVALID; }
  int n_cu = ncu[dev].load(std::memory_order_acquire);
  if (!n_cu) {
    MCEDM_HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    ncu[dev].store(n_cu, std::memory_order_release);
  }
  if (!attr_set[dev].load(std::memory_order_acquire)) {
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)conv_resident_kernel<C, RS, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
    attr_set[dev].store(true, std::memory_order_release);
  }
  char name[96] = "";
  if (prof_enabled())
    snprintf(name, sizeof(name), "conv_resident_kernel<ResCfg<%d, %d, %d, %d, %d, %d, %d, %d%s>, %d, %s>", C::MT, C::PH, C::PW, C::WM, C::WN,
             C::TAPS, C::KC, C::KS, C::DB ? ", 1" : "", RS, SPLIT ? "true" : "false");
  const double px = (double)a.B * a.H * a.W;
  const double skc = a.sk_wpk ? (double)(a.sk_Ca + a.sk_Cb) : 0.0;
  const double flops = 2.0 * px * a.Cout * ((double)(a.Ca + a.Cb) * C::TAPS + skc);
  const double bytes = 4.0 * ((double)a.B * (a.Ca + a.Cb) * a.Hs * a.Ws + px * skc + px * a.Cout * (a.res ? 2 : 1) +
                              (double)a.Cout * ((a.Ca + a.Cb) * C::TAPS + skc));
  ProfScope ps(name, flops, bytes, stream);
  hipLaunchKernelGGL((conv_resident_kernel<C, RS, SPLIT>), dim3((unsigned)blocks), dim3(256), (unsigned)plan.lds, stream, a, tiles_x,
                     tiles_y, mtiles, plan.pass_c, cout_padded(a.Cout), plan.nslab, plan.wide, SPLIT ? n_cu : 0);
  MCEDM_LAUNCH_CHECK("conv_resident_kernel");
  if (a.gsum_tiles) *a.gsum_tiles = SumTiles{tiles_x * tiles_y, tiles_x, C::PH, C::PW, a.gsum_rc == 2 ? 2 : 4};
  return MCEDM_OK;
}

// Returns MCEDM_OK after launching, or -1 when this shape is not served here (the caller falls through to
// conv_mfma_kernel).  The choice depends on the image size, channel counts and resampling mode only.
int try_launch_conv_resident(const ConvArgs& a, int taps, hipStream_t stream) {
  if (resident_level() <= 0) return -1;
  if (a.resample != RS_NONE && !(a.resample == RS_UP && taps == 9)) return -1;
  const bool small = (long long)a.H * a.W <= 256 || a.W < 12;          // the 8 x 8-pixel tile of dispatch()
  const size_t whole_cu = (size_t)LDS_MAX - 1024, half_cu = 80 * 1024 - 512;
  if (taps == 9) {
    typedef ResCfg<64, 8, 8, 2, 2, 9, 8> S;
    typedef ResCfg<64, 8, 16, 1, 4, 9, 8> M;
    typedef ResCfg<32, 4, 8, 1, 1, 9, 8, 4> K;
    if (a.H <= 8 && a.W <= 8 && a.resample == RS_NONE && cout_padded(a.Cout) % 32 == 0) {
      // <= 8 x 8 images: one <64, 8, 8> tile per sample would leave 3/4 of the CUs idle at B = 64.  Tiles of 32 channels
      // x 4 x 8 pixels with the K loop split over the four waves: 4x the workgroups, a quarter of the MFMA chain each.
      // (The K sum is grouped differently from conv_mfma_kernel's: same shape -> same bits, but not equal to that kernel's.)
      const ResidentPlan pl = resident_plan<K, false>(a, whole_cu, 32);
      if (pl.pass_c) return launch_resident<K, RS_NONE>(a, pl, stream);
    }
    if (small) {          // <= 256 workgroups per 64 samples: one per CU, the whole LDS
      if (cout_padded(a.Cout) % 64 != 0) return -1;
      if (db_level() > 0 && a.resample == RS_NONE) {     // short double-buffered passes: the K loop starts after 32 channels
        typedef ResCfg<64, 8, 8, 2, 2, 9, 8, 1, 1> SD;
        ResidentPlan pd = resident_plan<SD, false>(a, whole_cu, 32);
        if (pd.pass_c && pd.nslab == 3) { if (pd.pass_c > 32) pd.pass_c = 32; return launch_resident<SD, RS_NONE>(a, pd, stream); }
      }
      const ResidentPlan pl = resident_plan<S, false>(a, whole_cu, 64);
      if (!pl.pass_c) return -1;
      return a.resample == RS_UP ? launch_resident<S, RS_UP>(a, pl, stream) : launch_resident<S, RS_NONE>(a, pl, stream);
    }
    if (a.W >= 24 && (long long)a.H * a.W <= 1024) {
      // ~32 x 32 images, 512+ workgroups: half-chunk weight slabs and <= 64 resident channels per pass keep a workgroup
      // under 80 KB of LDS, so TWO share a CU and one computes while the other re-stages
      if (cout_padded(a.Cout) % 64 != 0) {          // the ch -> out_channels output conv: one padded 32-channel tile
        typedef ResCfg<32, 8, 16, 1, 4, 9, 8> M32;
        const ResidentPlan p32 = resident_plan<M32, true>(a, half_cu, 32);
        if (!p32.pass_c || p32.nslab != 3 || a.resample != RS_NONE) return -1;
        return launch_resident<M32, RS_NONE, true>(a, p32, stream);
      }
      if (db_level() > 0 && a.resample == RS_NONE) {     // 16-channel double-buffered passes
        typedef ResCfg<64, 8, 16, 1, 4, 9, 8, 1, 1> MD;
        ResidentPlan pd = resident_plan<MD, true>(a, half_cu, 16);
        if (pd.pass_c && pd.nslab == 3 && pd.wide) { if (pd.pass_c > 16) pd.pass_c = 16; return launch_resident<MD, RS_NONE, true>(a, pd, stream); }
      }
      ResidentPlan pl = resident_plan<M, true>(a, half_cu, 32);
      if (!pl.pass_c || pl.nslab != 3) return -1;
      static int pass_env = -1;      // experiments: MCEDM_RES_PASS = channels per pass (multiple of 16, <= the plan's)
      if (pass_env < 0) { const char* e = getenv("MCEDM_RES_PASS"); pass_env = e ? atoi(e) : 0; }
      if (pass_env >= 16 && pass_env % 16 == 0 && pass_env < pl.pass_c) pl.pass_c = pass_env;
      return a.resample == RS_UP ? launch_resident<M, RS_UP, true>(a, pl, stream) : launch_resident<M, RS_NONE, true>(a, pl, stream);
    }
    if ((long long)a.H * a.W >= 4096 && big_level() > 0 && a.resample == RS_NONE && cout_padded(a.Cout) == 64) {
      // >= 64 x 64 images, convs with ONE 64-channel output tile (the ch = 64 networks: the reference's own
      // adm_edm_mcedm_res32 and the DDPM U-Net): conv_mfma_kernel's <64, 8, 32> tile pays its staging (16 + 5 vector
      // loads and ~11 transformed elements per thread and chunk, all of it matrix-pipe time on gfx950) against half the
      // MFMAs of the 128-channel tile and holds 0.64 of peak.  Here the raw tile arrives by DMA as whole 16-byte row
      // segments (6 instructions per wave and 16-channel pass), is transformed in place, and no staging registers exist, so
      // a wave can own EIGHT accumulator blocks (64 channels x 16 x 32 pixels per workgroup) at two workgroups per CU.
      // The pixel tile is a function of the image size only (batch-shard bit-identity).
      static int b16_min = -1;      // MCEDM_RES_B16_MIN: smallest image (pixels) that takes the 16 x 32 tile (experiments)
      if (b16_min < 0) { const char* e = getenv("MCEDM_RES_B16_MIN"); b16_min = e ? atoi(e) : 16384; }
      if ((long long)a.H * a.W >= b16_min) {
        typedef ResCfg<64, 16, 32, 1, 4, 9, 8> B16;
        ResidentPlan pl = resident_plan<B16, true>(a, half_cu, 8);
        if (pl.pass_c && pl.nslab == 3 && pl.wide) { if (pl.pass_c > big_pass()) pl.pass_c = big_pass(); return launch_resident<B16, RS_NONE, true>(a, pl, stream); }
      } else {
        typedef ResCfg<64, 8, 32, 1, 4, 9, 8> B8;
        ResidentPlan pl = resident_plan<B8, true>(a, half_cu, 8);
        if (pl.pass_c && pl.nslab == 3 && pl.wide) { if (pl.pass_c > big_pass()) pl.pass_c = big_pass(); return launch_resident<B8, RS_NONE, true>(a, pl, stream); }
      }
    }
    if ((long long)a.H * a.W >= 4096 && big128_level() > 0 && a.resample == RS_NONE && cout_padded(a.Cout) % 128 == 0 && !a.sk_wpk) {
      // the same for 128-channel output tiles (ch = 128 networks: BASELINE config 3), 8 x 32 pixels, eight accumulator
      // blocks per wave.  MCEDM_RES_BIG128: 0 off, 1 on
      typedef ResCfg<128, 8, 32, 1, 4, 9, 8> B128;
      ResidentPlan pl = resident_plan<B128, true>(a, half_cu, 8);
      if (pl.pass_c && pl.wide) { if (pl.pass_c > big_pass()) pl.pass_c = big_pass(); return launch_resident<B128, RS_NONE, true>(a, pl, stream); }
    }
    return -1;
  }
  typedef ResCfg<64, 8, 8, 2, 2, 1, 16> P;
  if (small && !a.sk_wpk && cout_padded(a.Cout) % 64 == 0) {
    const ResidentPlan pl = resident_plan<P, false>(a, whole_cu, 64);
    if (pl.pass_c) return launch_resident<P, RS_NONE>(a, pl, stream);
  }
  // ~32 x 32 images (the decoder's skip projections there, launched on their own since conv1 runs on the Winograd kernel):
  // 1024 tiles per 64 samples, two workgroups per CU.  conv_mfma_kernel's <32, 8, 32> tile walks 8-16 chunks of
  // load -> barrier -> MFMA -> barrier with one accumulator block per wave and sits at 28 TFLOP/s there.
  static int p32 = -1;               // MCEDM_RES_1X1_32 (A/B runs): 0 keeps these convs on conv_mfma_kernel, 1: 8 x 8-pixel tiles, 2 (default): 8 x 16
  if (p32 < 0) { const char* e = getenv("MCEDM_RES_1X1_32"); p32 = e ? atoi(e) : 2; }
  if (p32 > 0 && !small && a.W >= 24 && (long long)a.H * a.W <= 1024 && !a.sk_wpk && cout_padded(a.Cout) % 64 == 0) {
    if (p32 == 2) {              // 8 x 16-pixel tiles: 512 workgroups per 64 samples = ONE round of two per CU (S32 1750 -> 1781 states/s; 8 x 8: two rounds)
      typedef ResCfg<64, 8, 16, 1, 4, 1, 16> P16;
      const ResidentPlan pl = resident_plan<P16, false>(a, half_cu, 64);
      if (pl.pass_c && pl.wide) return launch_resident<P16, RS_NONE>(a, pl, stream);
    }
    const ResidentPlan pl = resident_plan<P, false>(a, half_cu, 64);
    if (pl.pass_c && pl.wide) return launch_resident<P, RS_NONE>(a, pl, stream);
  }
  return -1;
}

}  // namespace mcedm
