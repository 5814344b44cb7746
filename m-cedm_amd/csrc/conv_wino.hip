// conv_wino.hip -- the un-resampled 3x3 convolution of the 128-channel levels in Winograd F(2x2, 3x3) form.
//
// Why: on gfx950 the fp32 matrix rate (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s dense) bounds conv_mfma_kernel, whose K loop
// already runs at 97 % of it.  The only thing left is to issue fewer MFMAs: F(2x2, 3x3) computes a
// 2 x 2 output patch from a 4 x 4 input patch with 16 multiplies per (cin, cout) instead of 36 -- 2.25x less matrix work --
// at the price of a 4x larger accumulator footprint (16 Winograd-domain values per 4 outputs) and of input / output
// transforms in vector instructions (which are paid in matrix time on this chip).
//
//   y = A^T [ sum_cin (G g G^T) o (B^T d B) ] A        (Lavin & Gray 2016; B, G, A below)
//
// Mapping to one CU (one 512-thread workgroup = 8 waves, 2 per SIMD):
//   * workgroup tile = 128 output channels x (8 x 16 pixels = 32 patches) x 16 Winograd positions = 64 accumulator blocks
//     of 32 x 32 = half of the CU's register file.  Wave (mb, hf) owns output channels 32 mb .. + 31 and the eight
//     positions (xi, nu) with xi in {2 hf, 2 hf + 1}: 128 accumulator registers.
//   * K loop over chunks of 8 input channels.  Per chunk and wave: 8 positions x 4 k-steps = 32 MFMAs; the A operand
//     (transformed weights U = G g G^T, packed per (chunk, 32-channel block, position, lane) so that a lane's four k-steps
//     are one 16-byte load) comes straight from L2 into registers, one chunk ahead -- no weight slab in LDS; the B operand
//     (transformed input V = B^T d B, LDS layout [position][patch][k parity][k-step]) is one ds_read_b128 per position.
//   * input path per chunk: wave w loads channel w of the chunk (10 x 18 raw patch incl. halo, 3 elements per lane), applies
//     the fused (x - mean) * scale + offset -> SiLU transform and writes the raw tile to LDS; one stage later every thread
//     turns (channel k, patch t, row half) of it into 8 Winograd-domain values (6 ds_read_b64, 16 adds, 8 ds_write_b32).
//     Both LDS stages are double-buffered, a stage is two chunks, ONE barrier per stage, and all of this side work is
//     issued in slices between the MFMAs of the same wave (see the K loop).
//   * epilogue: the output transform is register-local over nu; over xi the two halves exchange one 2 x 32-register
//     partial through LDS; then bias, residual, stores (8-byte, two pixels of a row) and the fused GroupNorm statistics.
// Results differ from the direct kernel in the last bits (a different, equally long fp32 sum: measured 1.5x its error
// against fp64, 0.1 of the rtol 1e-4 / atol 1e-5 bar); they do not depend on the batch size or the grid.
#include <atomic>
#include <cstdlib>

#include "conv_wino.hpp"
#include "pack.hpp"
#include "prof.hpp"

namespace mcedm {

// U = G g G^T for every (cout, cin) of one weight tensor (pack.hpp wino_pack_elem)
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int coutp, int nch,
                                 int tflip) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= coutp * nch * WKC) return;
  wino_pack_elem(w, dst, idx, Cout, Cin, coutp, tflip);
}

// UP: the conv input is the nearest-neighbour 2x up-sampling of the (activated) source (adm_blocks.py:69-73)
// ACT = false: the launch has no activation (p.act == 0: the data-gradient convs of training, whose input is a raw gradient): the
// plain variant's branch-free "evaluate SiLU and select" is compiled out -- ~20 of the ~58 vector instructions per chunk and lane,
// every one of them matrix time (round 5; the forward convs of inference all carry the activation)
template <class C, bool UP, bool ACT = true>
__global__ __launch_bounds__(C::NT, C::MB == 4 ? 1 : 2) void conv_wino_kernel(const ConvArgs p, int tiles_x, int tiles_img, int nch, int mblocks,
                                                                          int per, int mode) {
  constexpr int WSC = C::SC, VBUF = C::VBUF, RBUF = C::RBUF, MB = C::MB;
  extern __shared__ float lds[];
  const int Cin = p.Ca + p.Cb;
  float* const vbuf = lds;
  float* const rbuf = lds + 2 * VBUF;
  Coef* const cfl = reinterpret_cast<Coef*>(lds + C::LDS_ROWS_OFF);
  float* const xch = lds + C::LDS_ROWS_OFF + 4 * Cin;            // epilogue exchange (one 16-register round), then the records
  float* const red = xch + C::XCH_FLOATS;
  float* const bias_l = red + 2 * (C::MT / 2) * 3;                // this workgroup's MT bias values
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave % MB, hf = wave / MB;
  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 0] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 4] = per; }
  // A workgroup takes `per` consecutive pixel tiles of ONE sample (the launcher picks a divisor of the tiles per image) and
  // runs their stages as one stream: the pipeline below never drains between tiles, only the accumulators are written out.
  // mode bit 256 (round 5, MCEDM_WINO_MAP): the tiles_img / per workgroups of a sample share an XCD (ids 8 apart: one L2) and walk the
  // image INTERLEAVED -- workgroup j takes tiles j, j + wps, ... -- so that at any time they work on neighbouring tiles and fetch each
  // other's halo columns out of that L2 (with consecutive tiles per workgroup a tile's 24 x 10 fetch per 16 x 8 pixels had left the
  // 4 MB L2 long before the neighbouring tile asked for its overlap: 1.9x the input from HBM).  Same tiles, same sums: same bits.
  int gt0 = blockIdx.x * per, tstep = 1;
  int n = gt0 / tiles_img, tile0 = gt0 % tiles_img;
  if (mode & 256) {
    const int nwg = gridDim.x, wps = tiles_img / per;
    const int lid = (nwg % 8 == 0) ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    n = lid / wps; tile0 = lid % wps; tstep = wps;
  }
  const int m0 = blockIdx.y * C::MT;
  const size_t HW = (size_t)p.H * p.W, HWs = (size_t)p.Hs * p.Ws;
  const int nst = (nch + WSC - 1) / WSC;                           // stages per tile
  const int G = per * nst;                                        // stages of this workgroup

  // ---- raw staging: wave w stages channels w CPW .. of every chunk: a channel's 10 x 18 patch (tile + halo).
  //   plain input: ONE 16-byte load per lane and channel -- lane = (patch row r = lane / 6, aligned quad lane % 6 of the 24
  //   columns x0 - 4 .. x0 + 19): vector-memory instructions are the expensive thing in this loop (~60 cycles of issue
  //   each), and the 6 columns fetched beyond the patch cost two extra SiLU evaluations per lane instead;
  //   up-sampled input: elements lane, lane + 64, lane + 128 of the 180, one dword load each (source pixel (y/2, x/2)).
  // Geometry (byte offsets, zero-padding masks) of the tile that is being LOADED; the masks of the tile whose registers are
  // waiting to be COMMITTED are kept beside them (the two differ for one trip at a tile boundary).
  constexpr int NR = UP ? RSUB : 4, NG = UP ? RSUB : 1;             // raw registers / geometry entries per lane and channel
  unsigned roff[NG], rkeepL[NG], rkeepC[NG];
  int lofs[4];                                                      // plain input: where the quad's four elements go in the LDS patch
  if (!UP) {
    const int r = lane / 6, qd = lane % 6;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = 4 * qd + e - 3;                              // column inside the 18-wide patch; -3 .. -1 and 18 .. 20 are not part of it
      lofs[e] = (lane < 60 && col >= 0 && col < RPITCH) ? r * RPITCH + col : RROWS * RPITCH + lane % (RPLANE - RROWS * RPITCH);
    }
  }
  auto set_geom = [&](int tile) {
    const int y0 = (tile / tiles_x) * WPH, x0 = (tile % tiles_x) * WPW;
    if (UP) {
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int e = lane + 64 * i;
        const int r = e / RPITCH, c = e - r * RPITCH;
        const int y = y0 - 1 + r, x = x0 - 1 + c;
        const bool inb = e < RROWS * RPITCH && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        rkeepL[i] = inb ? 0xffffffffu : 0u;
        roff[i] = !inb ? 0u : 4u * (unsigned)((y >> 1) * p.Ws + (x >> 1));
      }
    } else {       // W % 16 == 0 and x0 % 16 == 0: a quad lies wholly inside or wholly outside the image, and is 16-byte aligned
      const int y = y0 - 1 + lane / 6, x = x0 - 4 + 4 * (lane % 6);
      const bool inb = lane < 60 && (unsigned)y < (unsigned)p.H && x >= 0 && x < p.W;
      rkeepL[0] = inb ? 0xffffffffu : 0u;
      roff[0] = inb ? 4u * (unsigned)(y * p.W + x) : 0u;
    }
  };
  float raw[WSC][C::CPW][NR];
  int ld_st = 0, ld_tile = tile0, cm_st = 0;                       // stage (within its tile) of the next load / commit
  set_geom(tile0);
  auto raw_load1 = [&](int sc) {
#pragma unroll
    for (int cw = 0; cw < C::CPW; ++cw) {
      const int ci = (ld_st * WSC + sc) * WKC + wave * C::CPW + cw;
      const bool in_a = ci < p.Ca;
      const float* src = in_a ? p.xa : p.xb;
      const int cc = in_a ? ci : ci - p.Ca, CC = in_a ? p.Ca : p.Cb;
      const bool ok = ci < Cin && src != nullptr;
      const float* plane = ok ? src + ((size_t)n * CC + cc) * HWs : (p.xa ? p.xa : p.xb);
      if (UP) {
#pragma unroll
        for (int i = 0; i < NG; ++i) raw[sc][cw][i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(plane) + roff[i]);
      } else {
        const f32x4 q = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(plane) + roff[0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) raw[sc][cw][e] = q[e];
      }
    }
  };
  auto raw_load_next = [&]() { if (++ld_st == nst) { ld_st = 0; ld_tile += tstep; set_geom(ld_tile < tiles_img ? ld_tile : tiles_img - 1); } };
  auto raw_load_done = [&]() {                                     // all chunks of the stage requested: on to the next stage
#pragma unroll
    for (int i = 0; i < NG; ++i) rkeepC[i] = rkeepL[i];            // these registers are committed one trip from now
  };
  auto raw_load = [&]() {                                          // the next stage of the stream
#pragma unroll
    for (int sc = 0; sc < WSC; ++sc) raw_load1(sc);
    raw_load_done();
    raw_load_next();
  };
  // The transform of a lane's four elements is written on PAIRS (v_pk_add / v_pk_fma / v_pk_mul_f32: two elements per vector
  // instruction) and the zero padding is folded into the lane's copy of the row -- (v - mean) * 0 + 0 --: vector instructions
  // do not overlap with the fp32 MFMAs of either wave of the SIMD, so every one removed from this loop is
  // matrix time.  Same operations in the same order as apply_coef(): bit-identical values.
  auto raw_commit1 = [&](int sc, float* rb) {
#pragma unroll
    for (int cw = 0; cw < C::CPW; ++cw) {
      const int kl = wave * C::CPW + cw, ci = (cm_st * WSC + sc) * WKC + kl;
      const bool ok = ci < Cin && (ci < p.Ca ? p.xa : p.xb) != nullptr;
      const Coef cf = cfl[ci < Cin ? ci : Cin - 1];
      const unsigned ck = ok ? 0xffffffffu : 0u;
      if (UP) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          float v = apply_coef(raw[sc][cw][i], cf, p.act);
          v = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & (rkeepC[i] & ck));
          if (i + 1 < RSUB || lane + 64 * i < RROWS * RPITCH) rb[(sc * WKC + kl) * RPLANE + lane + 64 * i] = v;
        }
      } else {
        // zero padding = (v - mean) * 0 + 0 on the lane's copy of the row.  PRECONDITION: finite activations -- an out-of-image
        // lane loads element 0 of the plane, and an Inf / NaN there (or a (v - mean) that overflows) would make the padding NaN
        // instead of 0; the up-sampling variant above masks the finished value bit-wise.  (A network whose activations are
        // not finite has no meaningful output either way; two v_and per pair here would cost ~3 % of the slice's vector work.)
        const unsigned mk = rkeepC[0] & ck;
        const float scm = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, cf.scale) & mk);
        const float ofm = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, cf.offset) & mk);
        f32x2 t[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 v = {raw[sc][cw][2 * h], raw[sc][cw][2 * h + 1]};
          t[h] = (v - cf.mean) * scm + ofm;
        }
        // no branch (a slot must stay ONE basic block, see the K loop): the activation is a template parameter
#pragma unroll
        for (int h = 0; h < 2; ++h) { if constexpr (ACT) t[h] = silu_f2(t[h]); }       // ACT == (p.act != 0): the launcher's choice
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[(sc * WKC + kl) * RPLANE + lofs[i]] = t[i >> 1][i & 1];
      }
    }
  };
  auto commit_done = [&]() { if (++cm_st == nst) cm_st = 0; };
  // ---- input transform: thread = (channel k, patch (ty, tx), row half hf): V[xi][nu] for xi in {2 hf, 2 hf + 1}.
  // In two halves (LDS reads / arithmetic + LDS writes) per chunk of the stage, so that the K loop can spread them out.
  // patch rows ty = mb + MB ti, ti < TI
  // B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] down the rows r0 .. r3 of the 4 x 4 patch.  Both halves evaluate ONE
  // formula, t0 = Y - X, t1 = X + s Z (no per-half selects): hf = 0: (Y, Z, X, s) = (r0, r1, r2, +1) gives (r0 - r2, r1 + r2),
  // hf = 1: (r2, r3, r1, -1) gives (r2 - r1, r1 - r3); the halves differ in two LDS row offsets and a sign.  Along the columns
  // (o0, o3) = (c0, c1) - (c2, c3) and (o1, -o2) = (c1, c1) + (c2, -c2): two packed additions per row; nu = 2 is stored negated
  // and the packed weights carry the same sign (wino_pack_kernel).  8 packed vector instructions per chunk and thread.
  const int tk = lane & 7, ttx = lane >> 3, tty = mb;
  const int tr_yz = tk * RPLANE + (2 * tty + 2 * hf) * RPITCH + 2 * ttx;
  const int tr_x = tk * RPLANE + (2 * tty + 2 - hf) * RPITCH + 2 * ttx;
  const float tsgn = hf ? -1.f : 1.f;
  const f32x2 tpm = {1.f, -1.f};
  const int tr_dst = (8 * hf) * VPOS + (tk & 1) * VH1 + (tty * WTX + ttx) * 4 + (tk >> 1);
  f32x2 td[WSC][C::TI][3][2];                         // rows Y, Z, X; column pairs (c0, c1), (c2, c3)
  auto transform_read1 = [&](int sc, const float* rb) {
#pragma unroll
    for (int ti = 0; ti < C::TI; ++ti)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float* b = rb + sc * WKC * RPLANE + 2 * MB * ti * RPITCH + 2 * j;
        td[sc][ti][0][j] = *reinterpret_cast<const f32x2*>(b + tr_yz);
        td[sc][ti][1][j] = *reinterpret_cast<const f32x2*>(b + tr_yz + RPITCH);
        td[sc][ti][2][j] = *reinterpret_cast<const f32x2*>(b + tr_x);
      }
  };
  auto transform_finish1 = [&](int sc, float* vb) {
#pragma unroll
    for (int ti = 0; ti < C::TI; ++ti) {
      f32x2 t[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        t[0][j] = td[sc][ti][0][j] - td[sc][ti][2][j];
        t[1][j] = td[sc][ti][2][j] + tsgn * td[sc][ti][1][j];
      }
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const f32x2 o03 = t[x][0] - t[x][1];
        const f32x2 c11 = {t[x][0].y, t[x][0].y}, c22 = {t[x][1].x, t[x][1].x};
        const f32x2 o12 = c11 + tpm * c22;
        float* o = vb + sc * 16 * VPOS + tr_dst + MB * ti * WTX * 4 + 4 * x * VPOS;
        o[0 * VPOS] = o03.x;
        o[1 * VPOS] = o12.x;
        o[2 * VPOS] = o12.y;
        o[3 * VPOS] = o03.y;
      }
    }
  };
  auto transform_read = [&](const float* rb) { for (int sc = 0; sc < WSC; ++sc) transform_read1(sc, rb); };
  auto transform_finish = [&](float* vb) { for (int sc = 0; sc < WSC; ++sc) transform_finish1(sc, vb); };

  // ---- prologue (once per workgroup): transform rows of the sample, stages 0 .. 2 of the stream
  stage_coef_rows<C::NT>(p, n, cfl, tid);
  if (tid < C::MT) bias_l[tid] = p.bias ? p.bias[m0 + tid] : 0.f;
  raw_load();
  __syncthreads();
  for (int sc = 0; sc < WSC; ++sc) raw_commit1(sc, rbuf);
  commit_done();
  if (G > 1) { raw_load(); for (int sc = 0; sc < WSC; ++sc) raw_commit1(sc, rbuf + RBUF); commit_done(); }
  if (G > 2) raw_load();                                             // stays in registers until trip 0 commits it
  // this wave's transformed weights: 8 positions x one 16-byte load per chunk
  const size_t ustride = (size_t)mblocks * 16 * 64;                  // f32x4 per chunk
  const f32x4* up = reinterpret_cast<const f32x4*>(p.wino) + ((size_t)(m0 / 32 + mb) * 16 + 8 * hf) * 64 + lane;
  f32x4 ua[8];                                                       // reloaded position by position right after its last use
#pragma unroll
  for (int q = 0; q < 8; ++q) ua[q] = up[q * 64];
  __syncthreads();
  transform_read(rbuf);
  transform_finish(vbuf);
  // Accumulators start at zero, except Winograd position (xi, nu) = (1, 1) -- block 5 of the hf = 0 waves --, which starts at
  // the bias: A^T has ones in column 1 of both rows, so a constant there comes out as that constant in all four pixels.
  // Zeroing is done by the matrix pipe itself -- D = 0 * 0 + 0 with the inline constant as C: eight instructions (512 cycles)
  // instead of 128 v_mov: the in-kernel stamps put the v_mov version at 3500 cycles per tile with both waves of a SIMD in it
  // (a wave issues a vector instruction every ~7 cycles at best here).
  f32x16 acc[8];
  auto init_acc = [&]() {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float z = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      asm volatile("" : "+v"(z));                          // an opaque zero per block: eight matrix instructions, not one + 112 copies
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, z, zero16, 0, 0, 0);
    }
    if (hf == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[5][r] = bias_l[32 * mb + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2)];
    }
  };
  __syncthreads();
  init_acc();

  // ---- the stream of stages: one trip = one stage of WSC chunks
  const int vrd = (8 * hf) * VPOS + (lane >> 5) * VH1 + (lane & 31) * 4;
  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memtime(); }
#ifdef MCEDM_WINO_TIMELINE      // cycle sums of waves 0 and MB (lane 0): MFMA stream, barrier, epilogue -> dbg[8..12] / dbg[13..15, 7] (diagnostic builds only)
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define WINO_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tlast; tlast = t_; }
  unsigned long long sl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, slast = 0;      // mode bit 16: cycles per slot of the stage (wave 0) -> dbg[8..15]
#define WINO_SLOT(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sl[(i) & 7] += t_ - slast; slast = t_; }
#define WINO_SLOT0 { slast = __builtin_amdgcn_s_memtime(); }
  unsigned long long ep[6] = {0, 0, 0, 0, 0, 0}, elast = 0;          // mode bit 32: cycles per phase of the epilogue (wave 0) -> dbg[8..13]
#define WINO_EP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (i) ep[(i) - 1] += t_ - elast; elast = t_; __builtin_amdgcn_sched_barrier(0); }
  const bool no_u = mode & 2;                                       // drop the weight reloads (wrong results; ablation)
#else
#define WINO_STAMP(i)
#define WINO_SLOT(i)
#define WINO_SLOT0
#define WINO_EP(i)
  constexpr bool no_u = false;
#endif
  // One trip = one stage: WSC x 4 slots of [two B-fragment reads for the next slot | eight MFMAs | two weight reloads] and,
  // behind each slot's MFMAs, one slice of the side work that prepares later stages -- the transform of stage g + 1 (LDS
  // reads in slots 0 / 3, arithmetic + LDS writes in slots 2 / 5), the commit of the raw tile of stage g + 2 (slots 1 / 4; its
  // global loads were issued a trip ago) and the loads of stage g + 3 (slot 6); stages past the end of a tile are the first
  // stages of the workgroup's next tile.  Sliced like this the side work costs its issue cycles; as a phase of its own
  // (before or after the MFMAs, or ping-ponged between the two waves of a SIMD) it cost its latencies: 2400 cycles per chunk
  // and wave against 2048 of MFMAs.  Every slice touches buffers no MFMA of this trip reads: one barrier per trip.
  auto side_slice = [&](int slot, int g, int cur) {
    // Every slice runs in every trip (past the end of the stream it works on stale registers and writes LDS buffers nobody
    // reads; the loads stay inside this sample's planes): a slot is then ONE basic block, and the scheduler can place the
    // slice's instructions between the slot's MFMAs (the recipe in the K loop).
    if (WSC == 2) {        // the two chunks' transforms one after the other: their 12 + 12 staging registers are never live together
      // a chunk's raw registers are requested again in the slot right behind their commit: the tile data (HBM) then has seven
      // slots = 7 / 8 of a stage to arrive (both chunks requested together in slot 6 left the first one three slots, ~1.5 us,
      // and its commit waited ~800 cycles for it: in-kernel slot stamps)
      if (slot == 0) transform_read1(0, rbuf + (cur ^ 1) * RBUF);
      if (slot == 1) raw_commit1(0, rbuf + cur * RBUF);
      if (slot == 2) { transform_finish1(0, vbuf + (cur ^ 1) * VBUF); raw_load1(0); }
      if (slot == 3) transform_read1(WSC - 1, rbuf + (cur ^ 1) * RBUF);
      if (slot == 4) { raw_commit1(WSC - 1, rbuf + cur * RBUF); commit_done(); }
      if (slot == 5) { transform_finish1(WSC - 1, vbuf + (cur ^ 1) * VBUF); raw_load1(WSC - 1); raw_load_done(); }
    } else {
      if (slot == 0) transform_read(rbuf + (cur ^ 1) * RBUF);
      if (slot == 1) { raw_commit1(0, rbuf + cur * RBUF); commit_done(); }
      if (slot == 2) { raw_load1(0); raw_load_done(); }
      if (slot == 3) transform_finish1(0, vbuf + (cur ^ 1) * VBUF);
    }
  };
  (void)mode;
  int st = 0, tile = tile0;                                        // stage within the tile, tile of the accumulators
  for (int g = 0; g < G; ++g) {
    const int cur = g & 1;
    WINO_STAMP(0)
    WINO_SLOT0
#pragma unroll
    for (int sc = 0; sc < WSC; ++sc) {
      const int c = st * WSC + sc;
      const f32x4* un = up + (size_t)(c + 1 < nch ? c + 1 : 0) * ustride;      // after a tile's last chunk: chunk 0 again
      const float* vb = vbuf + cur * VBUF + sc * 16 * VPOS + vrd;
      f32x4 b4[2][2];                                      // B fragments of two positions, one pair ahead of the MFMAs
      b4[0][0] = *reinterpret_cast<const f32x4*>(vb);
      b4[0][1] = *reinterpret_cast<const f32x4*>(vb + VPOS);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int qp = 0; qp < 4; ++qp) {
        const int cb = qp & 1;
        if (qp < 3) {
          b4[cb ^ 1][0] = *reinterpret_cast<const f32x4*>(vb + (2 * qp + 2) * VPOS);
          b4[cb ^ 1][1] = *reinterpret_cast<const f32x4*>(vb + (2 * qp + 3) * VPOS);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[2 * qp] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[2 * qp][s], b4[cb][0][s], acc[2 * qp], 0, 0, 0);
          acc[2 * qp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[2 * qp + 1][s], b4[cb][1][s], acc[2 * qp + 1], 0, 0, 0);
        }
        // the next chunk's weights of these two positions: 2 KB per wave every eight MFMAs.  (All of a chunk's weight loads
        // issued together right behind the barrier -- 64 KB per CU at once -- back up the vector memory path and every
        // wave stalls in their issue for 1000-3000 cycles per chunk.)
        if (!no_u) {
          ua[2 * qp] = un[(2 * qp) * 64];
          ua[2 * qp + 1] = un[(2 * qp + 1) * 64];
        }
        // The slice's instructions go BETWEEN the slot's MFMAs, up to WINO_IL_K behind each: its dependent chains (LDS read ->
        // arithmetic -> transcendental -> LDS write) then wait under matrix instructions instead of in front of the next slot
        // (as a block of its own behind the eight MFMAs a commit slice cost +900 cycles for ~25 vector instructions).
        side_slice(sc * 4 + qp, g, cur);
        if (qp < 3) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x486, WINO_IL_K, 0);           // VALU | SALU | DS | transcendental
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (sc * 4 + qp == (WSC == 2 ? 5 : 2)) raw_load_next();                 // the one conditional piece (tile geometry): a block of its own
        __builtin_amdgcn_sched_barrier(0);
        WINO_SLOT(sc * 4 + qp + (WSC == 1 ? 0 : 0))
      }
    }
    WINO_STAMP(1)
    __syncthreads();
    WINO_STAMP(2)
    if (++st < nst) continue;
    st = 0;

    // ---- a tile is complete: output transform, bias, residual, stores, statistics; then the accumulators start over.
    // (The stream's LDS buffers already hold the first stages of the next tile: the exchange area is a region of its own.)
    WINO_EP(0)
    const int y0 = (tile / tiles_x) * WPH, x0 = (tile % tiles_x) * WPW;
    const int pt = lane & 31;                             // this wave stores row hf of every patch: pixels
    const int oy = y0 + 2 * (pt >> 3) + hf, ox = x0 + 2 * (pt & 7);   // (y0 + 2 ty + hf, x0 + 2 tx + {0, 1}), patch = lane & 31
    const int cbase = m0 + 32 * mb + 4 * (lane >> 5);
    // buffer addressing for the tile's 16 + 16 + 16 memory operations: one descriptor per tensor (this sample's planes) in
    // SGPRs, one 32-bit lane offset, the channel step in the scalar offset -- no 64-bit address pair per operation
    const size_t splane = (size_t)p.Cout * HW;                      // this sample's output planes (the launcher checks < 4 GiB)
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out + (size_t)n * splane, 4u * (unsigned)splane);
    const size_t rplane = p.res_mode == RS_UP ? splane >> 2 : p.res_mode == RS_DOWN ? splane << 2 : splane;   // the residual's planes of this sample
    const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(p.res ? p.res + (size_t)n * rplane : nullptr, p.res ? 4u * (unsigned)rplane : 0u);
    const unsigned HWu = (unsigned)HW;
    const unsigned voff = 4u * ((unsigned)cbase * HWu + (unsigned)oy * p.W + ox);
    const unsigned rvoff = 4u * ((unsigned)cbase * (HWu >> 2) + (unsigned)(oy >> 1) * (p.W >> 1) + (ox >> 1));
    // A^T = [[1,1,1,0],[0,1,-1,-1]].  Over nu (register-local), then over xi across the two halves: row 0 of the patch =
    // T0 + T1 + T2, row 1 = T1 - T2 - T3 with T_xi the nu-reduced blocks; half 0 holds T0, T1, half 1 T2, T3.
    float2 rv[16];                                        // residual: in flight during the rest of the transform and the exchange
    f32x16 keep[2], send[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {                          // one pixel column at a time: fewer values live beside the accumulators
      const f32x16 t0 = j == 0 ? acc[0] + acc[1] + acc[2] : acc[1] - acc[2] - acc[3];
      const f32x16 t1 = j == 0 ? acc[4] + acc[5] + acc[6] : acc[5] - acc[6] - acc[7];
      if (hf == 0) { keep[j] = t0 + t1; send[j] = t1; }
      else         { keep[j] = -t0 - t1; send[j] = t0; }
      __builtin_amdgcn_sched_barrier(0);
    }
    // (chunk 0's weights for the next tile came with the stream's reload at this tile's last chunk.  Fetching them AGAIN here --
    // to end the first copy's life at the last MFMA -- was round 3's way to 32 free registers during the transform; without it the
    // kernel spills 3 registers instead of 7, issues 64 fewer vector-memory instructions per tile and CU, and is 0.4 % faster.
    // Exchanging the halves through 16-byte LDS accesses, lane pitch 20 floats, measured 1.5 % slower at 128 channels.)
    __builtin_amdgcn_sched_barrier(0);                    // the accumulators are dead from here to the end of the epilogue
    if (p.res && p.res_mode == RS_DOWN) {
      // residual at double resolution: the 2x2 mean of the source (adm_blocks.py:75-77), two 16-byte loads per channel and
      // pixel pair, four channels in flight at a time (all sixteen would need 128 registers)
      const unsigned dvoff = 4u * ((unsigned)cbase * (HWu << 2) + (unsigned)(2 * oy) * (unsigned)(2 * p.W) + (unsigned)(2 * ox));
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 4) {
        f32x4 ta[4], tb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned so = 4u * (unsigned)(((r0 + k) & 3) + 8 * ((r0 + k) >> 2)) * (HWu << 2);
          ta[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, dvoff, so, 0));
          tb[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, dvoff + 8u * (unsigned)p.W, so, 0));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          rv[r0 + k] = make_float2(0.25f * ((ta[k][0] + ta[k][1]) + (tb[k][0] + tb[k][1])), 0.25f * ((ta[k][2] + ta[k][3]) + (tb[k][2] + tb[k][3])));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (p.res && p.res_mode == RS_UP) {                // residual at half resolution: both pixels of the pair share a source
          const float q = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, rvoff, 4u * (unsigned)dr * (HWu >> 2), 0));
          rv[r] = make_float2(q, q);
        } else {                                           // no residual: a zero-sized descriptor reads zeros
          rv[r] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_res, voff, 4u * (unsigned)dr * HWu, 0));
        }
      }
    }
    WINO_EP(1)                                            // 1: output transform over nu, weight + residual requests
    const int pw = wave ^ MB;
    float v0[16], v1[16];
#pragma unroll
    for (int j = 0; j < 2; ++j) {                         // two rounds of 16 registers through the exchange area
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[(wave * 16 + r) * 64 + lane] = send[j][r];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float got = xch[(pw * 16 + r) * 64 + lane];
        if (j == 0) v0[r] = keep[0][r] + got + rv[r].x;
        else        v1[r] = keep[1][r] + got + rv[r].y;
      }
      if (j == 0) __syncthreads();
    }
    WINO_EP(2)                                            // 2: the two exchange rounds (incl. the wait for the residual)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, make_float2(v0[r], v1[r])), rs_out, voff,
                                            4u * (unsigned)((r & 3) + 8 * (r >> 2)) * HWu, 0);
    WINO_EP(3)                                            // 3: the 16 stores
    if (p.gsum) {
      // fused GroupNorm statistics of what was stored (conv_tile.hpp conv_epilogue): one record per 4-channel block =
      // registers 4 g .. 4 g + 3 of the 32 lanes that share lane >> 5 (or per 2-channel block, gsum_rc == 2: their two
      // halves); this wave holds one pixel row of every patch, its partner wave the other: (count, sum, M2 about the wave's
      // own mean) per wave, merged in a fixed order by conv_stats_store.
      const bool pairs = p.gsum_rc == 2;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float a[2], b[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          a[e] = half_sum32((v0[4 * gq + 2 * e] + v1[4 * gq + 2 * e]) + (v0[4 * gq + 2 * e + 1] + v1[4 * gq + 2 * e + 1]));
        }
        const float cnt = pairs ? 128.f : 256.f;
        const float mean0 = pairs ? a[0] * (1.0f / 128.0f) : (a[0] + a[1]) * (1.0f / 256.0f);
        const float mean1 = pairs ? a[1] * (1.0f / 128.0f) : mean0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float mean = e ? mean1 : mean0;
          float q = 0.f;
#pragma unroll
          for (int r = 4 * gq + 2 * e; r < 4 * gq + 2 * e + 2; ++r) { const float d0 = v0[r] - mean, d1 = v1[r] - mean; q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); }
          b[e] = half_sum32(q);
        }
        if ((lane & 31) == 0) {
          const int quad = 8 * mb + 2 * gq + (lane >> 5);
          if (pairs) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              float* slot = red + (hf * (C::MT / 2) + 2 * quad + e) * 3;
              slot[0] = cnt; slot[1] = a[e]; slot[2] = b[e];
            }
          } else {
            float* slot = red + (hf * (C::MT / 4) + quad) * 3;
            slot[0] = cnt; slot[1] = a[0] + a[1]; slot[2] = b[0] + b[1];
          }
        }
      }
      __syncthreads();
      conv_stats_store<C, 2>(p, red, n, m0, tile, tiles_img, tid);
      // the next epilogue writes `red` only after a whole tile of barriers
    }
    WINO_EP(4)                                            // 4: statistics
    __builtin_amdgcn_sched_barrier(0);
    init_acc();
    tile += tstep;
    WINO_EP(5)                                            // 5: accumulator initialisation
    WINO_STAMP(3)
  }
  if (p.dbg && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p.dbg[blockIdx.x * 16 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 6] = __builtin_amdgcn_s_memtime();
    p.dbg[blockIdx.x * 16 + 3] = __builtin_amdgcn_s_memrealtime();
  }
#ifdef MCEDM_WINO_TIMELINE
  if (p.dbg && tid == 0) for (int i = 0; i < 5; ++i) p.dbg[blockIdx.x * 16 + 8 + i] = ph[i];
  if (p.dbg && tid == 0 && (mode & 16)) for (int i = 0; i < 8; ++i) p.dbg[blockIdx.x * 16 + 8 + i] = sl[i];
  if (p.dbg && tid == 0 && (mode & 32)) for (int i = 0; i < 6; ++i) p.dbg[blockIdx.x * 16 + 8 + i] = ep[i];
  if (p.dbg && tid == 64 * MB && !(mode & 48)) { p.dbg[blockIdx.x * 16 + 13] = ph[1]; p.dbg[blockIdx.x * 16 + 14] = ph[2]; p.dbg[blockIdx.x * 16 + 15] = ph[3]; p.dbg[blockIdx.x * 16 + 7] = ph[0]; }
#endif
}

static int g_wino = -1;      // -1: default (env MCEDM_WINOGRAD, else on); 0 / 1: forced by mcedm_op_set_conv_wino
void set_conv_wino(int enable) { g_wino = enable; }
static int wino_env() {                                        // MCEDM_WINOGRAD=0: never take this kernel
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_WINOGRAD"); env = e ? atoi(e) : 1; }
  return variant_choice(KV_CONV_WINO, g_wino, env);            // the executing plan's choice, the process-wide hook, the environment
}
static int wino_mode_env() {                                   // MCEDM_WINO_MODE: ablation bits of the -DMCEDM_WINO_TIMELINE build (else unused)
  static int env = -1;                                         // | 256: the interleaved, XCD-aware tile map (MCEDM_WINO_MAP, default on)
  if (env < 0) {
    const char* e = getenv("MCEDM_WINO_MODE");
    const char* m = getenv("MCEDM_WINO_MAP");
    env = ((e ? atoi(e) : 0) & ~256) | ((m ? atoi(m) : 1) ? 256 : 0);
  }
  return env;
}

static int wino_min_hw_env() {                                 // MCEDM_WINO_MIN_HW: smallest image (pixels) served; below 32 x 32 the grid
  static int env = -1;                                  // (B * H * W / 128 workgroups) no longer fills the chip
  if (env < 0) { const char* e = getenv("MCEDM_WINO_MIN_HW"); env = e ? atoi(e) : 1024; }
  return env;
}

size_t conv_wino_packed_floats(int Cout, int Cin) { return (size_t)16 * cout_padded(Cout) * (size_t)ceil_div(Cin, WKC) * WKC; }

int launch_pack_conv_wino(const float* w, float* dst, int Cout, int Cin, int transpose_flip, hipStream_t stream) {
  MCEDM_REQUIRE(w && dst && Cout > 0 && Cin > 0, "pack_conv_wino: bad arguments");
  const int coutp = cout_padded(Cout), nch = ceil_div(Cin, WKC);
  hipLaunchKernelGGL(wino_pack_kernel, dim3(ceil_div(coutp * nch * WKC, 256)), dim3(256), 0, stream, w, dst, Cout, Cin, coutp, nch,
                     transpose_flip);
  MCEDM_LAUNCH_CHECK("wino_pack_kernel");
  return MCEDM_OK;
}

bool conv_wino_applicable(const ConvArgs& a, int taps) {
  // the plain variant stages its input with 16-byte loads (raw_load1): misaligned views take the direct kernel
  const bool aligned = a.resample != RS_NONE || (((size_t)a.xa | (size_t)a.xb) & 15) == 0;
  return taps == 9 && a.wino && (a.resample == RS_NONE || a.resample == RS_UP) && a.Cout % 64 == 0 && a.H % WPH == 0 && aligned &&
         a.W % WPW == 0 && (a.Ca + a.Cb) % WKC == 0 && a.Ca % WKC == 0 && !a.sk_wpk && ((size_t)a.wino & 15) == 0 &&
         (!a.res || a.res_mode == RS_NONE || a.res_mode == RS_UP || a.res_mode == RS_DOWN) && (a.Ca + a.Cb) <= 1024;
}

// Tiles per workgroup: a divisor d of the tiles per image (a workgroup stays inside one sample: one set of transform rows),
// chosen for the shortest schedule on `slots` concurrent workgroups -- rounds x (d tiles + the un-overlapped start and end
// of a workgroup, about a fifth of a tile).  A function of the shape and the chip only: results never depend on it.
int wino_tiles_per_wg(long long total, int tiles_img, int slots) {
  int best = 1;
  double best_t = 1e30;
  for (int d = 1; d <= tiles_img; ++d) {
    if (tiles_img % d) continue;
    const long long wgs = total / d;
    const double t = (double)((wgs + slots - 1) / slots) * (d + 0.2);
    if (t < best_t - 1e-9 || (t < best_t + 1e-9 && d > best)) { best_t = t; best = d; }
  }
  return best;
}

template <class C>
static int launch_wino_cfg(const ConvArgs& a, hipStream_t stream) {
  const int tiles_x = a.W / WPW, tiles_img = tiles_x * (a.H / WPH);
  const long long total = (long long)a.B * tiles_img;
  MCEDM_REQUIRE(total > 0 && total <= 0x7fffffffLL, "conv_wino: grid out of range");
  const int Cin = a.Ca + a.Cb, nch = Cin / WKC;
  const int lds_bytes = (C::LDS_ROWS_OFF + 4 * Cin + C::XCH_FLOATS + C::RED_FLOATS) * 4;
  MCEDM_REQUIRE(lds_bytes <= 160 * 1024 / (C::MB == 4 ? 1 : 2), "conv_wino: %d input channels exceed the LDS row table", Cin);
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> ncu[64];
  int dev = 0;
  MCEDM_HIP_TRY(hipGetDevice(&dev));
  MCEDM_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
  if (!attr_set[dev].load(std::memory_order_acquire)) {
    int n_cu = 0;
    MCEDM_HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    ncu[dev].store(n_cu > 0 ? n_cu : 256, std::memory_order_release);
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)conv_wino_kernel<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)conv_wino_kernel<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)conv_wino_kernel<C, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[dev].store(true, std::memory_order_release);
  }
  static int per_env = -1;                                 // MCEDM_WINO_PER: force the tiles per workgroup (A/B runs; must divide)
  if (per_env < 0) { const char* e = getenv("MCEDM_WINO_PER"); per_env = e ? atoi(e) : 0; }
  int per = wino_tiles_per_wg(total, tiles_img, ncu[dev].load(std::memory_order_acquire) * (C::MB == 4 ? 1 : 2));
  if (per_env > 0 && tiles_img % per_env == 0) per = per_env;
  char name[64] = "";
  const bool noact = a.resample != RS_UP && !a.act;
  if (prof_enabled()) snprintf(name, sizeof(name), "conv_wino_kernel<WinoCfg<%d>, %s, %s>", C::MB, a.resample == RS_UP ? "true" : "false",
                               noact ? "false" : "true");   // = rocprofv3's name
  const double px = (double)a.B * a.H * a.W;
  // algorithmic cost = the direct convolution's (2 * MAC); the kernel issues 4 / 9 of these as matrix flops
  ProfScope ps(name, 2.0 * px * a.Cout * (double)Cin * 9,
               4.0 * ((double)a.B * Cin * a.Hs * a.Ws + px * a.Cout * (a.res ? 2 : 1) + (double)a.Cout * Cin * 9), stream);
  if (a.resample == RS_UP)
    hipLaunchKernelGGL((conv_wino_kernel<C, true>), dim3((unsigned)(total / per), a.Cout / C::MT), dim3(C::NT), lds_bytes, stream, a, tiles_x,
                       tiles_img, nch, cout_padded(a.Cout) / 32, per, wino_mode_env());
  else if (noact)
    hipLaunchKernelGGL((conv_wino_kernel<C, false, false>), dim3((unsigned)(total / per), a.Cout / C::MT), dim3(C::NT), lds_bytes, stream, a, tiles_x,
                       tiles_img, nch, cout_padded(a.Cout) / 32, per, wino_mode_env());
  else
    hipLaunchKernelGGL((conv_wino_kernel<C, false>), dim3((unsigned)(total / per), a.Cout / C::MT), dim3(C::NT), lds_bytes, stream, a, tiles_x,
                       tiles_img, nch, cout_padded(a.Cout) / 32, per, wino_mode_env());
  MCEDM_LAUNCH_CHECK("conv_wino_kernel");
  if (a.gsum_tiles) *a.gsum_tiles = SumTiles{tiles_img, tiles_x, WPH, WPW, a.gsum_rc == 2 ? 2 : 4};
  return MCEDM_OK;
}

int launch_conv_wino(const ConvArgs& a_in, hipStream_t stream) {
  ConvArgs a = a_in;
  a.dbg = conv_debug_buffer();
  MCEDM_REQUIRE(conv_wino_applicable(a, 9), "conv_wino: shape not served by the Winograd kernel");
  MCEDM_REQUIRE(a.out && a.B > 0 && (a.resample == RS_UP ? (a.Hs * 2 == a.H && a.Ws * 2 == a.W) : (a.Hs == a.H && a.Ws == a.W)),
                "conv_wino: bad arguments");
  MCEDM_REQUIRE((unsigned long long)a.H * a.W * 4ull * a.Cout * (a.res && a.res_mode == RS_DOWN ? 4 : 1) < (1ull << 32),
                "conv_wino: one sample of the output / residual exceeds the 4 GiB buffer range");
  { const int rc = conv_resolve_identity(a); if (rc != MCEDM_OK) return rc; }
  // 128 output channels per workgroup where they divide (fewer passes over the input), else 64; the 128-channel shape runs on
  // the one-wave-per-SIMD kernel (conv_wino1.hip) only when that is switched on (MCEDM_WINO1=1; off by default)
  if (a.Cout % 128 == 0) {
    const int rc = try_launch_conv_wino1(a, stream);
    if (rc != -1) return rc;
  }
  return a.Cout % 128 == 0 ? launch_wino_cfg<WinoCfg<4>>(a, stream) : launch_wino_cfg<WinoCfg<2>>(a, stream);
}

// plan-time question (build_layout): will an un-resampled Cin -> Cout conv on H x W images be served here?
bool conv_wino_shape_ok(int Cout, int Cin, int H, int W) {
  return wino_env() != 0 && Cout % 64 == 0 && Cin % WKC == 0 && H % WPH == 0 && W % WPW == 0 && (long long)H * W >= wino_min_hw_env();
}
// the dispatcher's choice: enabled, and an image large enough for the grid to fill the chip
bool conv_wino_preferred(const ConvArgs& a) { return wino_env() != 0 && (long long)a.H * a.W >= wino_min_hw_env(); }

}  // namespace mcedm
