// conv_tile.hpp -- device-side building blocks shared by the implicit-GEMM convolution kernels (conv_mfma.hip,
// conv_resident.hip): tile configuration, input / weight staging, accumulator initialisation (bias + residual),
// the store + GroupNorm-statistics epilogue and the MFMA chunk loop.  Device code only; see conv_mfma.hip for the
// GEMM view and DESIGN.md section 3 for what bounds these kernels.
#pragma once
#include "common.hpp"

namespace mcedm {

// KC = input channels per PACKED weight chunk (8 for 3x3, 16 for 1x1); CPI = packed chunks staged and consumed per
// iteration of the K loop (one load -> barrier -> MFMA -> barrier round trip per KC * CPI channels).  The small tiles
// that serve <= 16 x 16 images run on grids of 64-256 workgroups, one per CU, whose duration is a chain of memory
// latencies, not matrix time: CPI > 1 shortens that chain.
template <int MT_, int PH_, int PW_, int WM_, int WN_, int TAPS_, int KC_, int NT_ = 256, int CPI_ = 1>
struct ConvCfg {
  static constexpr int MT = MT_, PH = PH_, PW = PW_, WM = WM_, WN = WN_, TAPS = TAPS_, KC = KC_;
  static constexpr int CPI = CPI_, KCI = KC_ * CPI_;   // channels per iteration
  static constexpr int NT = NT_;              // threads per workgroup
  static constexpr int HALO = (TAPS == 9) ? 1 : 0;
  static constexpr int PITCH = PW + 2 * HALO;
  static constexpr int ROWS = PH + 2 * HALO;
  static constexpr int PLANE = ROWS * PITCH;
  static constexpr int NPIX = PH * PW;
  static constexpr int TM = MT / WM / 32;    // 32x32 accumulator tiles per wave along M
  static constexpr int TN = NPIX / WN / 32;  // ... along N
  static constexpr int XL = KCI * PLANE;     // floats of the input tile
  static constexpr int WL = TAPS * KCI * MT; // floats of the weight slab: [cpi][tap][ci_local][MT]
  static constexpr int NWAVE = WM * WN;       // waves that own accumulators (the rest only help staging)
  // resident workgroups per CU the kernel is compiled for: <= 64 accumulator registers leave room for a third wave
  // per SIMD (<= 168 VGPRs), which hides more of the staging phases (+5 % on the MT = 64 tiles)
  static constexpr int OCC = (TM * TN <= 4) ? 3 : 2;
  static_assert(NWAVE >= 1 && NWAVE <= NT / 64, "more compute waves than the workgroup has");
  static_assert(TM >= 1 && TN >= 1 && MT % (WM * 32) == 0 && NPIX % (WN * 32) == 0, "tile shape");
  static_assert(MT % 4 == 0 && KC % 2 == 0, "vector widths");
};

// SiLU with the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each); relative error ~1e-6, far inside the
// 1e-4 parity bar, and a third of the VALU work of expf() + IEEE division in the staging path.
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// two elements at a time (v_pk_mul_f32 / v_pk_add_f32 around the two transcendentals each): the same operations as silu_f
__device__ __forceinline__ f32x2 silu_f2(f32x2 v) {
  const f32x2 a = v * -1.44269504088896340736f;
  f32x2 d = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  d = d + 1.0f;
  const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  return v * r;
}

__device__ __forceinline__ float apply_coef(float v, const Coef& c, int act) {
  float t = (v - c.mean) * c.scale + c.offset;
  return act ? silu_f(t) : t;
}

// ---- staging -----------------------------------------------------------------------------------------
// Split into an issue half (global loads -> registers) and a commit half (transform + LDS writes) so the
// loads of chunk c+1 are in flight while the MFMAs of chunk c run.  Everything that does not depend on the
// channel (tile coordinates, bounds, clamped source offsets) is computed once per workgroup; every load is
// unconditional from a clamped address (a guarded load costs a vmcnt(0) round trip in front of the MFMA loop)
// and out-of-image / padded-channel elements are masked to zero at commit.  The per-(sample, channel) transform
// rows are prefetched with the inputs.
template <class C, int RS>
struct TileGeom {
  static constexpr int NL = (RS == RS_DOWN || RS == RS_S2) ? 4 : 1;
  static constexpr int SUB = (C::PLANE + C::NT - 1) / C::NT;
  unsigned boff[SUB][NL];   // clamped BYTE offsets inside one channel plane (the same for every channel)
  unsigned keep[SUB];       // all-ones: element lies inside the image; 0: it is conv zero padding
};

// Weight slab: buffer addressing (one 128-bit descriptor in SGPRs + a 32-bit per-lane byte offset + a scalar
// offset), so no per-load address VGPRs.  (The input planes keep scalar-base pointers + the 32-bit lane offsets
// of TileGeom: a descriptor per channel costs more SGPRs than the kernel has, and spills.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <class C, int RS>
__device__ __forceinline__ void make_geom(const ConvArgs& p, TileGeom<C, RS>& G, int y0, int x0, int tid) {
#pragma unroll
  for (int sub = 0; sub < TileGeom<C, RS>::SUB; ++sub) {
    const int e = tid + sub * C::NT;
    const int r = e / C::PITCH;
    const int c = e - r * C::PITCH;
    // RS_S2: tile element (r, c) stands for the four source pixels (2y + {0,1}, 2x + {0,1}) of output-grid position
    // (y, x) = (y0 + r, x0 + c); tap (a, b) of output pixel (py, px) reads phase (a&1, b&1) at (py + a/2, px + b/2)
    const int y = y0 + r - (RS == RS_S2 ? 0 : C::HALO);
    const int x = x0 + c - (RS == RS_S2 ? 0 : C::HALO);
    const bool inb = (e < C::PLANE) && ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
    G.keep[sub] = inb ? 0xffffffffu : 0u;
    const int yc = inb ? y : 0, xc = inb ? x : 0;
    if constexpr (RS == RS_NONE) {
      G.boff[sub][0] = 4u * (unsigned)(yc * p.Ws + xc);
    } else if constexpr (RS == RS_UP) {
      G.boff[sub][0] = 4u * (unsigned)((yc >> 1) * p.Ws + (xc >> 1));
    } else {
      const unsigned o = 4u * (unsigned)((2 * yc) * p.Ws + 2 * xc);
      G.boff[sub][0] = o; G.boff[sub][1] = o + 4u; G.boff[sub][2] = o + 4u * p.Ws; G.boff[sub][3] = o + 4u * p.Ws + 4u;
    }
  }
}


// This sample's Ca + Cb transform rows -> LDS, once per workgroup; the commit phase reads them as wave-uniform
// broadcasts (24 fewer VGPRs and 8 fewer loads per chunk than carrying the rows of the next chunk in registers).
// gn_on: the rows are derived right here from the per-tile (sum, M2) tables that the PRODUCING convs' epilogues
// wrote (GroupNorm statistics in fp64, fixed-order butterfly over the lanes that share a group; + FiLM), so the
// inference path runs no GroupNorm kernel at all: the input is normalised by the conv that consumes it.
// Otherwise they are copied from the table p.coef (the launcher points a missing table at one identity row).
template <int NT>
__device__ __forceinline__ void stage_coef_rows(const ConvArgs& p, int n, Coef* cfl, int tid) {
  const int C = p.Ca + p.Cb;
  if (!p.gn_on) {
    for (int i = tid; i < C; i += NT) cfl[i] = p.coef[((p.coef_batch ? (size_t)n * C : 0) + i) * p.coef_rows];
    return;
  }
  const GnArgs& g = p.gn;
  const int G = g.groups, cpg = C / G;
  int lpg = NT / G;                                    // lanes per group: a power of two <= 64
  lpg = lpg >= 64 ? 64 : lpg >= 32 ? 32 : lpg >= 16 ? 16 : lpg >= 8 ? 8 : lpg >= 4 ? 4 : lpg >= 2 ? 2 : 1;
  const int gpr = NT / lpg;                            // groups per round
  const int sub = tid % lpg;
  for (int g0 = 0; g0 < G; g0 += gpr) {
    const int gi = g0 + tid / lpg;
    const bool live = gi < G;
    const int c0 = live ? gi * cpg : 0;
    // the group = whole statistic records (4 or 2 channels each, SumTiles::rc), each in xa's or xb's table (may straddle).
    // Chan merge of the per-tile records (sum_t, M2_t about the tile mean), all in fp64:
    //   M2 = sum_t M2_t + sum_t s_t^2 / n_t - (sum_t s_t)^2 / N      (s_t are fp32 values: exact in fp64)
    double s1 = 0, sq = 0, mw = 0;
    const int Himg = g.HW / g.W;
    for (int cb = c0; cb < (live ? c0 + cpg : c0);) {
      const bool in_a = cb < g.Ca;
      const float* sums = in_a ? g.suma : g.sumb;
      const SumTiles& tg = in_a ? g.ta : g.tb;
      const int nrec = ((in_a ? g.Ca : g.Cb) + tg.rc - 1) / tg.rc;
      const int qi = (in_a ? cb : cb - g.Ca) / tg.rc;
      for (int t = sub; t < tg.tiles; t += lpg) {
        const float* row = sums + (((size_t)n * tg.tiles + t) * nrec + qi) * 2;
        const double st = (double)row[0];
        s1 += st; sq += st * st / (double)sum_tile_count(tg, t, Himg, g.W); mw += (double)row[1];
      }
      cb += tg.rc;
    }
    for (int off = lpg >> 1; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); sq += __shfl_xor(sq, off); mw += __shfl_xor(mw, off); }
    if (live) {
      const double N = (double)cpg * g.HW;
      const double m = s1 / N;
      double var = (mw + sq - s1 * s1 / N) / N;
      if (var < 0) var = 0;
      const float mean = (float)m;
      const float rstd = (float)(1.0 / sqrt(var + (double)g.eps));
      for (int k = sub; k < cpg; k += lpg) {
        const int c = c0 + k;
        float sc = 1.f, sh = 0.f;
        if (g.film) {
          const float* f = g.film + (size_t)(g.film_batch ? n : 0) * g.film_stride;
          sc = f[c] + 1.f;
          sh = f[C + c];
        }
        Coef o;
        o.mean = mean;
        o.scale = g.gamma[c] * rstd * sc;
        o.offset = g.beta[c] * sc + sh;
        o.pad = 0.f;
        cfl[c] = o;
      }
    }
  }
}

template <class C, int RS>
struct InputRegs {
  float raw[C::KCI][TileGeom<C, RS>::SUB][TileGeom<C, RS>::NL];
  Coef cf[C::KCI];      // unused (and optimised away) when the transform rows are read from LDS at commit
};

template <class C, int RS, bool COEF_REGS = true>
__device__ __forceinline__ void load_input(const ConvArgs& p, const TileGeom<C, RS>& G, InputRegs<C, RS>& R, int n,
                                           int c0) {
  const int Cin = p.Ca + p.Cb;
  const size_t src_plane = (size_t)p.Hs * p.Ws;
  const float* safe = p.xa ? p.xa : p.xb;     // any valid plane for padded channels (values are discarded)
#pragma unroll
  for (int cil = 0; cil < C::KCI; ++cil) {
    const int ci = c0 + cil;
    const bool in_a = ci < p.Ca;
    const float* src = in_a ? p.xa : p.xb;
    const int cc = in_a ? ci : ci - p.Ca;
    const int CC = in_a ? p.Ca : p.Cb;
    const bool chan_ok = (ci < Cin) && (src != nullptr);
    const float* plane = chan_ok ? src + ((size_t)n * CC + cc) * src_plane : safe;
    // Unconditional load (a guarded one costs a vmcnt(0) round trip per channel): padded channels read a clamped row
    // and are zeroed in store_input; without a table the launcher points coef at one identity row (coef_rows == 0).
    if (COEF_REGS) R.cf[cil] = p.coef[((p.coef_batch ? (size_t)n * Cin : 0) + (ci < Cin ? ci : Cin - 1)) * p.coef_rows];
#pragma unroll
    for (int sub = 0; sub < TileGeom<C, RS>::SUB; ++sub)
#pragma unroll
      for (int q = 0; q < TileGeom<C, RS>::NL; ++q) R.raw[cil][sub][q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(plane) + G.boff[sub][q]);
  }
}

template <class C, int RS>
__device__ __forceinline__ void store_input(const ConvArgs& p, const TileGeom<C, RS>& G, float* xl,
                                            const InputRegs<C, RS>& R0, int c0, int tid, const Coef* cfl = nullptr) {
  const int Cin = p.Ca + p.Cb;
  const InputRegs<C, RS>& R = R0;
  // The transform rows of this sample were staged in LDS by the caller (wave-uniform broadcast reads).  All KCI rows are
  // fetched up front (their LDS latencies overlap; fetched channel by channel behind the per-channel scheduling barrier
  // below, each read's latency is exposed: measured -15 % on the small tiles) unless KCI is large (3 registers per row).
  constexpr bool PRELOAD = C::KCI <= 16;
  Coef rows[PRELOAD ? C::KCI : 1];
  if (PRELOAD && cfl) {
#pragma unroll
    for (int cil = 0; cil < C::KCI; ++cil) rows[cil] = cfl[c0 + cil < Cin ? c0 + cil : Cin - 1];
  }
#pragma unroll
  for (int cil = 0; cil < C::KCI; ++cil) {
    const int ci = c0 + cil;
    const Coef cfr = cfl ? (PRELOAD ? rows[PRELOAD ? cil : 0] : cfl[ci < Cin ? ci : Cin - 1]) : R0.cf[cil];
    const bool chan_ok = (ci < Cin) && ((ci < p.Ca ? p.xa : p.xb) != nullptr);
    const unsigned ckeep = chan_ok ? 0xffffffffu : 0u;
#pragma unroll
    for (int sub = 0; sub < TileGeom<C, RS>::SUB; ++sub) {
      if constexpr (RS == RS_S2) {
        // the four phases go to four planes of the (4x larger) LDS tile: [ci_local][phase][ROWS][PITCH]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float vq = apply_coef(R.raw[cil][sub][q], cfr, p.act);
          vq = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, vq) & (G.keep[sub] & ckeep));
          if ((sub + 1) * C::NT <= C::PLANE || tid + sub * C::NT < C::PLANE) xl[(cil * 4 + q) * C::PLANE + tid + sub * C::NT] = vq;
        }
        continue;
      }
      float v;
      if constexpr (RS == RS_DOWN) {
        // 2x2 box filter of the ACTIVATED source (adm_blocks.py:75-77 runs after silu(norm(x)))
        v = 0.25f * ((apply_coef(R.raw[cil][sub][0], cfr, p.act) + apply_coef(R.raw[cil][sub][1], cfr, p.act)) +
                     (apply_coef(R.raw[cil][sub][2], cfr, p.act) + apply_coef(R.raw[cil][sub][3], cfr, p.act)));
      } else {
        v = apply_coef(R.raw[cil][sub][0], cfr, p.act);
      }
      // zero padding / padded channels as a bit mask: straight-line code (a select makes the compiler branch
      // around the SiLU, ~30 tiny basic blocks per chunk)
      v = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & (G.keep[sub] & ckeep));
      if ((sub + 1) * C::NT <= C::PLANE || tid + sub * C::NT < C::PLANE) xl[cil * C::PLANE + tid + sub * C::NT] = v;
    }
    // one channel at a time: left alone, the scheduler interleaves all KC SiLU chains of this straight-line code
    // and pays for the extra live values with accumulator spills
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- 1x1 convs on rows that are 16-byte addressable (W % 4 == 0, aligned planes): the 8 x 32 tile has no halo, so a thread
// stages FOUR consecutive pixels of a channel with one 16-byte load and one 16-byte LDS store -- KCI / 4 vector-memory
// instructions per thread and chunk instead of KCI (each costs ~60 cycles of issue beside the MFMAs).  Same registers
// (InputRegs::raw, element 4 it + e = pixel e of this thread's quad in channel wave + 4 it), same LDS image, same values.
template <class C>
struct Vec4Geom {
  // the 128- and 64-channel tiles (the 64-channel one runs three workgroups per CU on 168 registers: 13 instead of 6 spilled
  // registers with the second staging path, outside the chunk loop; ref128 179.4 -> 182.8 states/s, repaint128 6.16 -> 6.30)
  static constexpr bool OK = C::TAPS == 1 && (C::MT == 128 || C::MT == 64) && C::PW == 32 && C::PLANE == 256 && C::NT == 256 && C::KCI % 4 == 0;
  unsigned off;     // byte offset of the quad inside a channel plane
  unsigned keep;    // all-ones: inside the image
};
template <class C>
__device__ __forceinline__ void make_vec4_geom(const ConvArgs& p, Vec4Geom<C>& G, int y0, int x0, int tid) {
  const int pix = 4 * (tid & 63);
  const int y = y0 + pix / C::PW, x = x0 + pix % C::PW;
  const bool inb = y < p.H && x < p.W;              // W % 4 == 0: the quad is wholly inside or wholly outside
  G.keep = inb ? 0xffffffffu : 0u;
  G.off = inb ? 4u * (unsigned)(y * p.Ws + x) : 0u;
}
template <class C, int RS>
__device__ __forceinline__ void load_input_v4(const ConvArgs& p, const Vec4Geom<C>& G, InputRegs<C, RS>& R, int n, int c0, int wave) {
  const int Cin = p.Ca + p.Cb;
  const size_t src_plane = (size_t)p.Hs * p.Ws;
  const float* safe = p.xa ? p.xa : p.xb;
#pragma unroll
  for (int it = 0; it < C::KCI / 4; ++it) {
    const int ci = c0 + wave + 4 * it;
    const bool in_a = ci < p.Ca;
    const float* src = in_a ? p.xa : p.xb;
    const int cc = in_a ? ci : ci - p.Ca, CC = in_a ? p.Ca : p.Cb;
    const bool chan_ok = (ci < Cin) && (src != nullptr);
    const float* plane = chan_ok ? src + ((size_t)n * CC + cc) * src_plane : safe;
    const f32x4 q = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(plane) + G.off);
#pragma unroll
    for (int e = 0; e < 4; ++e) R.raw[4 * it + e][0][0] = q[e];
  }
}
template <class C, int RS>
__device__ __forceinline__ void store_input_v4(const ConvArgs& p, const Vec4Geom<C>& G, float* xl, const InputRegs<C, RS>& R, int c0,
                                               int tid, int wave, const Coef* cfl) {
  const int Cin = p.Ca + p.Cb;
#pragma unroll
  for (int it = 0; it < C::KCI / 4; ++it) {
    const int cil = wave + 4 * it, ci = c0 + cil;
    const Coef cfr = cfl[ci < Cin ? ci : Cin - 1];
    const bool chan_ok = (ci < Cin) && ((ci < p.Ca ? p.xa : p.xb) != nullptr);
    const unsigned m = G.keep & (chan_ok ? 0xffffffffu : 0u);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      v[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, apply_coef(R.raw[4 * it + e][0][0], cfr, p.act)) & m);
    *reinterpret_cast<f32x4*>(xl + cil * C::PLANE + 4 * (tid & 63)) = v;
  }
}

// weights: rows of MT floats out of the packed [chunk][tap][ci_local][CoutP] table
template <class C>
struct WeightRegs {
  static constexpr int NV4 = C::WL / 4;
  static constexpr int IT = (NV4 + C::NT - 1) / C::NT;
  f32x4 v[IT];   // native vector type: HIP's float4 class defeats SROA here and lands in scratch
};

// NT threads fetch NT float4 per step = ROWS_IT whole rows of the slab; the per-lane byte offset is the same
// for every step and chunk, the (chunk, step) part is a scalar offset.
template <class C>
struct WeightGeom {
  static constexpr int V4_PER_ROW = C::MT / 4;
  static constexpr int ROWS_IT = C::NT / V4_PER_ROW;
  static_assert(C::NT % V4_PER_ROW == 0, "a staging step covers whole slab rows");
  unsigned voff;        // byte offset of this lane's float4 inside a step
  unsigned voff_last;   // the same for the last step, clamped into the slab when that step is partial
  __amdgpu_buffer_rsrc_t rs;
};

template <class C>
__device__ __forceinline__ void make_wgeom(const float* wpk, WeightGeom<C>& G, int m0, int coutp, int nchunks, int tid) {
  constexpr int V4 = WeightGeom<C>::V4_PER_ROW, NV4 = WeightRegs<C>::NV4, IT = WeightRegs<C>::IT;
  const int row = tid / V4, c4 = tid - row * V4;
  G.voff = 4u * (unsigned)(row * coutp + c4 * 4);
  int il = tid + (IT - 1) * C::NT;
  if (il >= NV4) il = NV4 - 1;
  const int rl = il / V4 - (IT - 1) * WeightGeom<C>::ROWS_IT, cl = il % V4;
  G.voff_last = 4u * (unsigned)(rl * coutp + cl * 4);
  // nchunks = PACKED chunks: a partial last iteration (CPI > 1) reads past the table and gets zeros from the range check
  G.rs = make_rsrc(wpk + m0, 4u * (unsigned)((size_t)nchunks * C::TAPS * C::KC * coutp - m0));
}

template <class C>
__device__ __forceinline__ void load_weights(const WeightGeom<C>& G, WeightRegs<C>& R, int ch, int coutp) {
  constexpr int IT = WeightRegs<C>::IT;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const unsigned soff = 4u * (unsigned)((ch * (C::TAPS * C::KCI) + it * WeightGeom<C>::ROWS_IT) * coutp);
    const unsigned voff = (it == IT - 1) ? G.voff_last : G.voff;
    R.v[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(G.rs, voff, soff, 0));
  }
}

template <class C>
__device__ __forceinline__ void store_weights(float* wl, const WeightRegs<C>& R, int tid) {
#pragma unroll
  for (int it = 0; it < WeightRegs<C>::IT; ++it) {
    const int i = tid + it * C::NT;
    if (i < WeightRegs<C>::NV4) reinterpret_cast<f32x4*>(wl)[i] = R.v[it];
  }
}

// Accumulators start at bias (+ resampled residual): the residual is fetched once, up front, with every load of the
// tile in flight together and overlapping the first chunk's staging.  (Added in the epilogue it costs one
// load -> wait -> store round trip per accumulator tile, ~3x the time of a store-only epilogue when both resident
// workgroups of a CU are busy.)  MODE: 0 bias only, 1 residual at the output or half resolution, 2 residual at
// double resolution (2x2 box filter, adm_blocks.py:75-77).  Out-of-range channels / pixels read clamped
// addresses; they are never stored.
template <class C, int MODE, bool FULL>
__device__ __forceinline__ void conv_init_acc(const ConvArgs& p, f32x16 (&acc)[C::TM][C::TN], int n, int m0, int y0,
                                              int x0, int wm, int wn, int lane) {
  const int sh = (MODE == 1 && p.res_mode == RS_UP) ? 1 : 0;
  const unsigned Wr = (MODE == 2) ? p.W * 2 : (p.W >> sh);
  const unsigned HWr = (MODE == 2) ? (unsigned)p.H * p.W * 4u : (unsigned)(p.H >> sh) * Wr;
  // one descriptor for this sample's residual planes; lane part of the address in voffset, the per-register
  // channel step in the scalar offset (the launcher checks Cout * HWr * 4 < 4 GiB)
  __amdgpu_buffer_rsrc_t rs;
  if (MODE != 0) rs = make_rsrc(p.res + (size_t)n * p.Cout * HWr, 4u * (unsigned)p.Cout * HWr);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.bias, p.bias ? 4u * (unsigned)p.Cout : 0u);   // no bias: zero records
  const __amdgpu_buffer_rsrc_t rb2 = make_rsrc(p.sk_bias, p.sk_bias ? 4u * (unsigned)p.Cout : 0u);   // folded skip projection's bias
#pragma unroll
  for (int i = 0; i < C::TM; ++i) {
    const int cbase = m0 + (wm * C::TM + i) * 32 + 4 * (lane >> 5);     // + (r&3) + 8*(r>>2)
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)   // channels past Cout are out of the descriptor's range and read as 0
      bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, 4u * (unsigned)(cbase + (r & 3) + 8 * (r >> 2)), 0, 0)) +
              __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb2, 4u * (unsigned)(cbase + (r & 3) + 8 * (r >> 2)), 0, 0));
#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
      const int pix = (wn * C::TN + j) * 32 + (lane & 31);
      const unsigned y = min(y0 + pix / C::PW, p.H - 1);
      const unsigned x = min(x0 + pix % C::PW, p.W - 1);
      const unsigned pixoff = (MODE == 2) ? (2u * y) * Wr + 2u * x : (y >> sh) * Wr + (x >> sh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        unsigned voff, soff;
        if (FULL) { voff = 4u * ((unsigned)cbase * HWr + pixoff); soff = 4u * (unsigned)dr * HWr; }
        else { voff = 4u * ((unsigned)min(cbase + dr, p.Cout - 1) * HWr + pixoff); soff = 0u; }
        if (MODE == 0) {
          acc[i][j][r] = bv[r];
        } else if (MODE == 1) {
          acc[i][j][r] = bv[r] + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
        } else {
          const float q00 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
          const float q01 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4u, soff, 0));
          const float q10 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4u * Wr, soff, 0));
          const float q11 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4u * Wr + 4u, soff, 0));
          acc[i][j][r] = bv[r] + 0.25f * ((q00 + q01) + (q10 + q11));
        }
      }
    }
  }
}

// Sum over the 32 lanes of a half-wave, result in every lane: four DPP steps inside the 16-lane rows (vector-ALU moves) and ONE
// cross-row exchange, instead of five __shfl_xor = five ds_bpermute_b32 (LDS-pipe round trips, ~80 per tile and wave in the
// statistics of the epilogue).  A fixed order like the butterfly it replaces: deterministic, independent of the batch.
__device__ __forceinline__ float half_sum32(float v) {
#define MCEDM_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
  v += MCEDM_DPP(v, 0xB1);       // quad_perm [1, 0, 3, 2]
  v += MCEDM_DPP(v, 0x4E);       // quad_perm [2, 3, 0, 1]
  v += MCEDM_DPP(v, 0x141);      // row_half_mirror: lane i <-> 7 - i of its eight
  v += MCEDM_DPP(v, 0x140);      // row_mirror: lane i <-> 15 - i of its row
#undef MCEDM_DPP
  return v + __shfl_xor(v, 16);
}

// FULL: every output channel of the tile exists (m0 + MT <= Cout).  Store-only: nothing here waits on memory.
// STATS (0 / 4 / 2): fused GroupNorm statistics of what is stored, one record per wave and STATS-channel block (4: the
// four accumulator registers of a lane that are consecutive channels; 2: their two halves -- GroupNorm(32) over 64
// channels has 2-channel groups, models/ddim_blocks.py:62-63).  A record is (count, sum, M2) with
// M2 = sum (v - wave mean)^2: two passes over the accumulator registers, so a large mean never cancels in fp32
// (E[x^2] - E[x]^2 on fp32 partial sums loses rstd at |mean|/std ~ 30).  Records go to the LDS slot of the wave
// (3 floats per block); conv_stats_combine merges the waves of the tile in a fixed order.
template <class C, bool FULL, int STATS>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[C::TM][C::TN], int n, int m0, int y0,
                                              int x0, int wm, int wn, int lane, float* red) {
  static_assert(STATS == 0 || STATS == 4 || STATS == 2, "record width");
  constexpr int NR = STATS == 2 ? 8 : 4;            // records per lane: register r belongs to record r / STATS
  constexpr int RSH = STATS == 2 ? 1 : 2;
  const size_t HW = (size_t)p.H * p.W;
#pragma unroll
  for (int i = 0; i < C::TM; ++i) {
    const int cbase = m0 + (wm * C::TM + i) * 32 + 4 * (lane >> 5);     // + (r&3) + 8*(r>>2)
    float gs1[NR], cnt[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) { gs1[q] = 0.f; cnt[q] = 0.f; }
#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
      const int pix = (wn * C::TN + j) * 32 + (lane & 31);
      const int y = y0 + pix / C::PW;
      const int x = x0 + pix % C::PW;
      if (y < p.H && x < p.W) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cbase + (r & 3) + 8 * (r >> 2);
          const float v = acc[i][j][r];
          if (FULL || co < p.Cout) {
            p.out[((size_t)n * p.Cout + co) * HW + (size_t)y * p.W + x] = v;
            if (STATS) { gs1[r >> RSH] += v; cnt[r >> RSH] += 1.f; }
          }
        }
      }
    }
    if (STATS) {
      // block inside the MT tile: quads gl = 8 (wm TM + i) + 2 (r>>2) + (lane>>5), pairs 2 gl + ((r>>1)&1);
      // pixels = the 32-lane half.
      float mean[NR];
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        gs1[q] = half_sum32(gs1[q]); cnt[q] = half_sum32(cnt[q]);
        mean[q] = cnt[q] > 0.f ? gs1[q] / cnt[q] : 0.f;
      }
      float gm2[NR];
#pragma unroll
      for (int q = 0; q < NR; ++q) gm2[q] = 0.f;
#pragma unroll
      for (int j = 0; j < C::TN; ++j) {
        const int pix = (wn * C::TN + j) * 32 + (lane & 31);
        const int y = y0 + pix / C::PW;
        const int x = x0 + pix % C::PW;
        if (y < p.H && x < p.W) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = cbase + (r & 3) + 8 * (r >> 2);
            if (FULL || co < p.Cout) { const float d = acc[i][j][r] - mean[r >> RSH]; gm2[r >> RSH] = fmaf(d, d, gm2[r >> RSH]); }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const float b = half_sum32(gm2[q]);
        if ((lane & 31) == 0) {
          const int quad = (wm * C::TM + i) * 8 + 2 * (STATS == 2 ? q >> 1 : q) + (lane >> 5);
          const int gl = STATS == 2 ? 2 * quad + (q & 1) : quad;
          float* slot = red + (wn * (C::MT / (STATS ? STATS : 4)) + gl) * 3;
          slot[0] = cnt[q]; slot[1] = gs1[q]; slot[2] = b;
        }
      }
    }
  }
}

// The epilogue for the statistics mode a launch asks for (ConvArgs::gsum / gsum_rc); wave-uniform dispatch.
template <class C>
__device__ __forceinline__ void conv_epilogue_any(const ConvArgs& p, f32x16 (&acc)[C::TM][C::TN], int n, int m0, int y0,
                                                  int x0, int wm, int wn, int lane, float* red) {
  const bool full = (m0 + C::MT <= p.Cout);
  if (!p.gsum) {
    if (full) conv_epilogue<C, true, 0>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
    else conv_epilogue<C, false, 0>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
  } else if (p.gsum_rc == 2) {
    if (full) conv_epilogue<C, true, 2>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
    else conv_epilogue<C, false, 2>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
  } else {
    if (full) conv_epilogue<C, true, 4>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
    else conv_epilogue<C, false, 4>(p, acc, n, m0, y0, x0, wm, wn, lane, red);
  }
}

// After conv_epilogue_any and a workgroup barrier: merge the NW waves' records of every block of the tile (fixed order)
// and store them as row (sample n, tile `tile` of `ntiles`) of the table.
template <class C, int NW>
__device__ __forceinline__ void conv_stats_store(const ConvArgs& p, const float* red, int n, int m0, int tile, int ntiles,
                                                 int tid);

// Merge NW per-wave records (count, sum, M2) of one 4-channel block in a fixed order (bitwise reproducible):
// M2 = sum_w [M2_w + n_w (mean_w - mean)^2].  Returns (sum, M2).
template <int NW>
__device__ __forceinline__ void conv_stats_combine(const float* red, int stride_w, float& sum_out, float& m2_out) {
  float nt = 0.f, st = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { nt += red[w * stride_w]; st += red[w * stride_w + 1]; }
  const float mean = nt > 0.f ? st / nt : 0.f;
  float m2 = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float nw = red[w * stride_w], sw = red[w * stride_w + 1];
    const float d = (nw > 0.f ? sw / nw : mean) - mean;
    m2 += red[w * stride_w + 2] + nw * d * d;
  }
  sum_out = st; m2_out = m2;
}

template <class C, int NW>
__device__ __forceinline__ void conv_stats_store(const ConvArgs& p, const float* red, int n, int m0, int tile, int ntiles,
                                                 int tid) {
  const int rc = p.gsum_rc == 2 ? 2 : 4;
  const int NG = C::MT / rc;                      // records of the tile
  for (int e = tid; e < NG; e += C::NT) {
    float sum, m2;
    conv_stats_combine<NW>(red + e * 3, NG * 3, sum, m2);
    const int g = m0 / rc + e;
    const int ngroups = (p.Cout + rc - 1) / rc;
    if (g < ngroups) {
      float* row = p.gsum + (((size_t)n * ntiles + tile) * ngroups + g) * 2;
      row[0] = sum; row[1] = m2;
    }
  }
}

// One K chunk (KC input channels x all taps) of the implicit GEMM out of the LDS slabs.
// UNROLL_TAPS: the tap loop fully unrolled (all LDS offsets immediates, no loop-carried address arithmetic; VALU
// and bubbles in this loop are paid in matrix time): +6...18 % on the small tiles, +0.6 % on <128, 8, 32>.
// S2: stride-2 conv on the 4-phase tile (channel stride 4 planes; tap (a, b) -> phase (a&1, b&1), offset (a/2, b/2)).
struct NoTapHook { __device__ __forceinline__ void operator()(int) const {} };
// T0, T1: the taps [T0, T1) only, out of a slab that starts at tap T0 (the input-resident kernel streams a chunk's
// weights in two halves; the order of the K sum is the same).
// HOOK: called once per tap with (tap - T0), after the first k-step of that tap has been issued (the matrix pipe is busy
// for the next TM * TN * 64 cycles): the input-resident kernel issues one weight-DMA instruction of a later slab there.
template <class C, bool UNROLL_TAPS = true, bool S2 = false, int T0 = 0, int T1 = C::TAPS, class HOOK = NoTapHook>
__device__ __forceinline__ void mfma_chunk(const float* xl, const float* wl, f32x16 (&acc)[C::TM][C::TN], int aoff,
                                           const int (&boff)[C::TN], const HOOK& hook = HOOK()) {
  static_assert(T0 >= 0 && T0 < T1 && T1 <= C::TAPS, "tap range");
  constexpr int CS = S2 ? 4 * C::PLANE : C::PLANE;          // floats between consecutive input channels in LDS
  auto tap_off = [](int tap) {
    if (C::TAPS != 9) return 0;
    const int a = tap / 3, b = tap % 3;
    return S2 ? ((a & 1) * 2 + (b & 1)) * C::PLANE + (a >> 1) * C::PITCH + (b >> 1) : a * C::PITCH + b;
  };
  // Register double-buffered operand fragments: the LDS reads of k-step s+1 are issued before the MFMAs
  // of k-step s; the fragment for the next tap's first k-step is fetched at the end of the current tap.
  float fa[2][C::TM], fb[2][C::TN];
#pragma unroll
  for (int i = 0; i < C::TM; ++i) fa[0][i] = wl[aoff + i * 32];
#pragma unroll
  for (int j = 0; j < C::TN; ++j) fb[0][j] = xl[boff[j] + tap_off(T0)];
  constexpr int TAP_UNROLL = UNROLL_TAPS ? T1 - T0 : 1;
#pragma unroll TAP_UNROLL
  for (int tap = T0; tap < T1; ++tap) {
    const int toff = tap_off(tap);
    const int tn = (tap + 1 < T1) ? tap + 1 : tap;              // clamped: the last prefetch is discarded
    const int toff_n = tap_off(tn);
    const float* wt = wl + aoff + (tap - T0) * C::KC * C::MT;
    const float* wt_n = wl + aoff + (tn - T0) * C::KC * C::MT;
#pragma unroll
    for (int kk = 0; kk < C::KC / 2; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < C::KC / 2) {
#pragma unroll
        for (int i = 0; i < C::TM; ++i) fa[nxt][i] = wt[2 * (kk + 1) * C::MT + i * 32];
#pragma unroll
        for (int j = 0; j < C::TN; ++j) fb[nxt][j] = xl[boff[j] + toff + 2 * (kk + 1) * CS];
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
      if (kk == 0) hook(tap - T0);
    }
  }
}

// ---- LDS-DMA (global -> LDS without passing through VGPRs) ----------------------------------------------------------
// Issued from inline assembly, NOT through __builtin_amdgcn_global_load_lds: with the builtin in a kernel, hipcc's
// wait-count insertion treats every later LDS read as possibly ordered against a pending "flat" operation and degrades
// the counted waits of the MFMA loop (ds_read x2 -> s_waitcnt lgkmcnt(2) -> MFMA) to lgkmcnt(0) after every pair of
// k-steps.  The compiler does not see these loads at all, so: every consumer must be ordered by an explicit
// s_waitcnt vmcnt (+ barrier), and compiler-generated vmcnt waits around them are only ever more conservative (the
// hardware counter includes them).  Address = scalar base + 32-bit per-lane byte offset: no per-load vector arithmetic.
// M0 = LDS byte address of lane 0's element; lane i writes at M0 + i * size.  (M0 is not otherwise used by these kernels;
// it cannot be named in the clobber list.)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
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

// Weight slab of chunk `ch`: the LDS image is the linear [tap][ci_local][MT] slab, float4 number i = tid + it * NT
// written by thread tid in step it (lane-linear per wave, as the instruction needs); a wave past the end of the slab
// issues nothing.  One step of it (for callers that spread the steps over their MFMA loop), and all of them.
template <class C>
struct WeightDma {
  static constexpr int V4 = C::MT / 4, NV4 = C::WL / 4, IT = (NV4 + C::NT - 1) / C::NT, ROWS_IT = C::NT / V4;
  static_assert(NV4 % 64 == 0 && C::NT % V4 == 0, "whole waves in the last staging step, whole rows per step");
  unsigned voff;      // byte offset of this lane's float4 inside a step
  int wave;
  __device__ __forceinline__ void init(int coutp, int tid) {
    voff = 4u * (unsigned)((tid / V4) * coutp + (tid % V4) * 4);
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  }
  __device__ __forceinline__ void step(int it, const float* chunk_base, unsigned wl_addr, int coutp) const {
    if (it * C::NT + wave * 64 < NV4)       // wave-uniform
      lds_dma16(chunk_base + (size_t)it * ROWS_IT * coutp, voff, wl_addr + 16u * (unsigned)(it * C::NT + wave * 64));
  }
};
template <class C>
__device__ __forceinline__ void dma_weights(const float* wpk, float* wl, int ch, int m0, int coutp, int tid) {
  WeightDma<C> d;
  d.init(coutp, tid);
  const float* base = wpk + (size_t)ch * (C::TAPS * C::KC) * coutp + m0;
  const unsigned wa = lds_addr(wl);
#pragma unroll
  for (int it = 0; it < WeightDma<C>::IT; ++it) d.step(it, base, wa, coutp);
}

}  // namespace mcedm
