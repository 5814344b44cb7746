// conv_wino1.hip -- the Winograd F(2x2, 3x3) convolution of the 128-channel layers with ONE wave per SIMD (round 4).
//
// conv_wino.hip runs 8 waves per CU, two per SIMD: wave (mb, hf) owns eight of the sixteen Winograd positions of a 32-channel
// block -- 128 accumulator registers, all of its 256 VGPRs spoken for -- and the two halves meet in an LDS exchange in the
// epilogue.  What the in-kernel stamps of that kernel showed: its side work is not issue-bound
// (a third fewer vector instructions: +1 %), the epilogue is 12.5 % of the kernel with the matrix pipe idle (exchange rounds,
// three barriers, 450 vector instructions per wave at 7-13 cycles each with two waves of a SIMD in it), and the producer-wave
// split VERDICT r3 asked for cannot be allocated: a kernel's waves all get the same register count, a third wave per SIMD
// means <= 168 VGPRs, and 128 accumulators + 32 weight + 16 fragment registers are 176.
//
// This kernel takes the other road the register file offers: 4 waves per CU, one per SIMD, each with the unified file's full
// 512 registers -- the sixteen accumulator blocks of ALL positions of its 32-channel block in the 256 AccVGPRs (the MFMAs
// read and write them in place), and 256 architectural VGPRs for everything else.  Consequences:
//   * the output transform is register-local in both directions: no LDS exchange, no barrier in the epilogue but the one in
//     front of the statistics table;
//   * there is register room to keep every load a full chunk (weights) or most of a stage (raw tile) ahead;
//   * nothing covers a stall of the wave -- so every side-work instruction is placed BETWEEN the MFMAs of a slot by the
//     scheduling recipe (a slot is one basic block), and every LDS / global result is requested at least a slot before use.
// Same tile (128 channels x 8 x 16 pixels), same packed weight table, same LDS layouts, same order of every sum as
// conv_wino_kernel<WinoCfg<4>>: the two kernels are bit-identical, statistics included (tests/test_hip_wino.py).
#include <atomic>
#include <cstdlib>

#include "conv_wino.hpp"
#include "prof.hpp"

namespace mcedm {

struct Wino1Cfg {
  static constexpr int MB = 4, NT = 256, NW = 4, MT = 128;
  static constexpr int SC = 2;                            // chunks per stage (one barrier per stage)
  static constexpr int CPW = WKC / NW;                    // channels of a chunk staged by one wave: 2
  static constexpr int VBUF = SC * 16 * VPOS, RBUF = SC * WKC * RPLANE;
  static constexpr int LDS_ROWS_OFF = 2 * VBUF + 2 * RBUF;
  static constexpr int RED_FLOATS = 2 * (MT / 2) * 3 + MT;
  static_assert(LDS_ROWS_OFF % 4 == 0, "LDS layout");
};

template <bool UP>
__global__ __launch_bounds__(Wino1Cfg::NT, 1) void conv_wino1_kernel(const ConvArgs p, int tiles_x, int tiles_img, int nch, int mblocks,
                                                                    int per) {
  using C = Wino1Cfg;
  constexpr int WSC = C::SC, VBUF = C::VBUF, RBUF = C::RBUF;
  extern __shared__ float lds[];
  const int Cin = p.Ca + p.Cb;
  float* const vbuf = lds;
  float* const rbuf = lds + 2 * VBUF;
  Coef* const cfl = reinterpret_cast<Coef*>(lds + C::LDS_ROWS_OFF);
  float* const red = lds + C::LDS_ROWS_OFF + 4 * Cin;            // statistics records of the two pixel rows, then the bias row
  float* const bias_l = red + 2 * (C::MT / 2) * 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int mb = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave = 32-channel block
  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 0] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 4] = per; }
  const int gt0 = blockIdx.x * per;
  const int n = gt0 / tiles_img, tile0 = gt0 % tiles_img;
  const int m0 = blockIdx.y * C::MT;
  const size_t HW = (size_t)p.H * p.W, HWs = (size_t)p.Hs * p.Ws;
  const int nst = (nch + WSC - 1) / WSC;                           // stages per tile
  const int G = per * nst;                                        // stages of this workgroup

  // ---- raw staging (as conv_wino.hip): wave w stages channels 2 w, 2 w + 1 of every chunk
  constexpr int NR = UP ? RSUB : 4, NG = UP ? RSUB : 1;
  unsigned roff[NG], rkeepL[NG], rkeepC[NG];
  int lofs[4];
  if (!UP) {
    const int r = lane / 6, qd = lane % 6;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = 4 * qd + e - 3;
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
    } else {
      const int y = y0 - 1 + lane / 6, x = x0 - 4 + 4 * (lane % 6);
      const bool inb = lane < 60 && (unsigned)y < (unsigned)p.H && x >= 0 && x < p.W;
      rkeepL[0] = inb ? 0xffffffffu : 0u;
      roff[0] = inb ? 4u * (unsigned)(y * p.W + x) : 0u;
    }
  };
  float raw[WSC][C::CPW][NR];
  int ld_st = 0, ld_tile = tile0, cm_st = 0;
  set_geom(tile0);
  auto raw_load1 = [&](int sc) {
#pragma unroll
    for (int cw = 0; cw < C::CPW; ++cw) {
      const int ci = (ld_st * WSC + sc) * WKC + mb * C::CPW + cw;
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
  auto raw_load_next = [&]() { if (++ld_st == nst) { ld_st = 0; ++ld_tile; set_geom(ld_tile < tiles_img ? ld_tile : tiles_img - 1); } };
  auto raw_load_done = [&]() {
#pragma unroll
    for (int i = 0; i < NG; ++i) rkeepC[i] = rkeepL[i];
  };
  auto raw_commit1 = [&](int sc, int cw, float* rb) {
    const int kl = mb * C::CPW + cw, ci = (cm_st * WSC + sc) * WKC + kl;
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
      const unsigned mk = rkeepC[0] & ck;
      const float scm = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, cf.scale) & mk);
      const float ofm = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, cf.offset) & mk);
      f32x2 t[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 v = {raw[sc][cw][2 * h], raw[sc][cw][2 * h + 1]};
        t[h] = (v - cf.mean) * scm + ofm;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) { const f32x2 a = silu_f2(t[h]); t[h] = p.act ? a : t[h]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[(sc * WKC + kl) * RPLANE + lofs[i]] = t[i >> 1][i & 1];
    }
  };
  auto commit_done = [&]() { if (++cm_st == nst) cm_st = 0; };

  // ---- input transform: thread = (channel k, patch (ty = wave, tx)): all sixteen V[xi][nu] of the 4 x 4 patch.
  // B^T down the rows: t0 = r0 - r2, t1 = r1 + r2, t2 = r2 - r1, t3 = r1 - r3; along the columns of each:
  // (o0, o3) = (c0, c1) - (c2, c3), (o1, -o2) = (c1, c1) + (c2, -c2) (nu = 2 is stored negated, like the packed weights).
  const int tk = lane & 7, ttx = lane >> 3, tty = mb;
  const int tr_src = tk * RPLANE + (2 * tty) * RPITCH + 2 * ttx;
  const int tr_dst = (tk & 1) * VH1 + (tty * WTX + ttx) * 4 + (tk >> 1);
  const f32x2 tpm = {1.f, -1.f};
  f32x2 td[4][2];                                                   // rows r0 .. r3; column pairs (c0, c1), (c2, c3)
  auto transform_read1 = [&](int sc, const float* rb) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) td[a][j] = *reinterpret_cast<const f32x2*>(rb + sc * WKC * RPLANE + tr_src + a * RPITCH + 2 * j);
  };
  // half = 0: positions xi = 0, 1; half = 1: xi = 2, 3.  The same operations, in the same order, as the two halves of
  // conv_wino.hip's transform (hf = 0: Y - X, X + Z with (Y, Z, X) = (r0, r1, r2); hf = 1: (r2, r3, r1) and X - Z).
  auto transform_finish1 = [&](int sc, int half, float* vb) {
    f32x2 t[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (half == 0) { t[0][j] = td[0][j] - td[2][j]; t[1][j] = td[2][j] + 1.f * td[1][j]; }
      else           { t[0][j] = td[2][j] - td[1][j]; t[1][j] = td[1][j] + -1.f * td[3][j]; }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const f32x2 o03 = t[x][0] - t[x][1];
      const f32x2 c11 = {t[x][0].y, t[x][0].y}, c22 = {t[x][1].x, t[x][1].x};
      const f32x2 o12 = c11 + tpm * c22;
      float* o = vb + sc * 16 * VPOS + tr_dst + (8 * half + 4 * x) * VPOS;
      o[0 * VPOS] = o03.x;
      o[1 * VPOS] = o12.x;
      o[2 * VPOS] = o12.y;
      o[3 * VPOS] = o03.y;
    }
  };

  // ---- prologue (once per workgroup): transform rows of the sample, stages 0 .. 2 of the stream
  stage_coef_rows<C::NT>(p, n, cfl, tid);
  if (tid < C::MT) bias_l[tid] = p.bias ? p.bias[m0 + tid] : 0.f;
  auto raw_load_stage = [&]() { for (int sc = 0; sc < WSC; ++sc) raw_load1(sc); raw_load_done(); raw_load_next(); };
  auto raw_commit_stage = [&](float* rb) {
    for (int sc = 0; sc < WSC; ++sc) for (int cw = 0; cw < C::CPW; ++cw) raw_commit1(sc, cw, rb);
    commit_done();
  };
  raw_load_stage();
  __syncthreads();
  raw_commit_stage(rbuf);
  raw_load_stage();
  raw_commit_stage(rbuf + RBUF);
  raw_load_stage();                                                  // stays in registers until trip 0 commits it
  const size_t ustride = (size_t)mblocks * 16 * 64;                  // f32x4 per chunk
  const f32x4* up = reinterpret_cast<const f32x4*>(p.wino) + ((size_t)(m0 / 32 + mb) * 16) * 64 + lane;
  f32x4 ua[16];                                                      // reloaded position by position right after its last use
#pragma unroll
  for (int q = 0; q < 16; ++q) ua[q] = up[q * 64];
  __syncthreads();
  for (int sc = 0; sc < WSC; ++sc) {
    transform_read1(sc, rbuf);
    transform_finish1(sc, 0, vbuf);
    transform_finish1(sc, 1, vbuf);
  }
  // accumulators: zero from the matrix pipe (D = 0 * 0 + 0), position (1, 1) starts at the bias (A^T's column of ones)
  f32x16 acc[16];
  auto init_acc = [&]() {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float z = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q == 5) continue;
      asm volatile("" : "+v"(z));
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, z, zero16, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[5][r] = bias_l[32 * mb + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2)];
  };
  __syncthreads();

  // ---- the stream of stages.  One trip = one stage = WSC x 8 slots of [two B-fragment reads for the next slot | eight MFMAs
  // (two positions x four k-steps) | two weight reloads], each with one slice of the side work between its MFMAs.
  const int vrd = (lane >> 5) * VH1 + (lane & 31) * 4;
  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memtime(); }
  auto side_slice = [&](int slot, int cur) {
    float* const rb_c = rbuf + cur * RBUF;                          // commit target: the raw tile of stage g + 2
    const float* const rb_t = rbuf + (cur ^ 1) * RBUF;              // transform source: the raw tile of stage g + 1
    float* const vb_t = vbuf + (cur ^ 1) * VBUF;
    const int sc = slot >> 3, s = slot & 7;
    if (s == 0) transform_read1(sc, rb_t);
    if (s == 1) raw_commit1(sc, 0, rb_c);
    if (s == 2) transform_finish1(sc, 0, vb_t);
    if (s == 3) transform_finish1(sc, 1, vb_t);
    if (s == 4) { raw_commit1(sc, 1, rb_c); if (sc == WSC - 1) commit_done(); }
    if (s == 5) { raw_load1(sc); if (sc == WSC - 1) raw_load_done(); }
  };
  // Two nested loops, tiles outside and the tile's stages inside (the stream's state -- stage parity, the side work's stage and
  // tile counters -- runs on across tiles): with the epilogue inside ONE flat loop of stages, hipcc's register allocation split
  // the sixteen accumulator tuples around it and spilled 180 registers inside the K loop.
  int g = 0;
#ifdef MCEDM_WINO_TIMELINE
  unsigned long long tl_loop = 0, tl_epi = 0, tl_t = __builtin_amdgcn_s_memtime();
#define W1_STAMP(acc_) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - tl_t; tl_t = t_; }
#else
#define W1_STAMP(acc_)
#endif
  for (int tile = tile0; tile < tile0 + per; ++tile) {
  init_acc();
  W1_STAMP(tl_epi)
  for (int st = 0; st < nst; ++st, ++g) {
    const int cur = g & 1;
#pragma unroll
    for (int sc = 0; sc < WSC; ++sc) {
      const int c = st * WSC + sc;
      const f32x4* un = up + (size_t)(c + 1 < nch ? c + 1 : 0) * ustride;      // after a tile's last chunk: chunk 0 again
      const float* vb = vbuf + cur * VBUF + sc * 16 * VPOS + vrd;
      f32x4 b4[2][2];
      b4[0][0] = *reinterpret_cast<const f32x4*>(vb);
      b4[0][1] = *reinterpret_cast<const f32x4*>(vb + VPOS);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int qp = 0; qp < 8; ++qp) {
        const int cb = qp & 1;
        if (qp < 7) {
          b4[cb ^ 1][0] = *reinterpret_cast<const f32x4*>(vb + (2 * qp + 2) * VPOS);
          b4[cb ^ 1][1] = *reinterpret_cast<const f32x4*>(vb + (2 * qp + 3) * VPOS);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[2 * qp] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[2 * qp][s], b4[cb][0][s], acc[2 * qp], 0, 0, 0);
          acc[2 * qp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[2 * qp + 1][s], b4[cb][1][s], acc[2 * qp + 1], 0, 0, 0);
        }
        ua[2 * qp] = un[(2 * qp) * 64];
        ua[2 * qp + 1] = un[(2 * qp + 1) * 64];
        side_slice(sc * 8 + qp, cur);
        if (qp < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x486, WINO_IL_K, 0);           // VALU | SALU | DS | transcendental
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (sc == WSC - 1 && qp == 5) raw_load_next();                          // tile geometry: the one conditional piece
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  W1_STAMP(tl_loop)

    // ---- a tile is complete: output transform (register-local), residual, stores, statistics; accumulators start over.
    const int y0 = (tile / tiles_x) * WPH, x0 = (tile % tiles_x) * WPW;
    const int pt = lane & 31;
    const int oy = y0 + 2 * (pt >> 3), ox = x0 + 2 * (pt & 7);          // the patch's pixels (oy + {0, 1}, ox + {0, 1})
    const int cbase = m0 + 32 * mb + 4 * (lane >> 5);
    const size_t splane = (size_t)p.Cout * HW;
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out + (size_t)n * splane, 4u * (unsigned)splane);
    const size_t rplane = p.res_mode == RS_UP ? splane >> 2 : p.res_mode == RS_DOWN ? splane << 2 : splane;
    const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(p.res ? p.res + (size_t)n * rplane : nullptr, p.res ? 4u * (unsigned)rplane : 0u);
    const unsigned HWu = (unsigned)HW, Wb = 4u * (unsigned)p.W;
    const unsigned voff = 4u * ((unsigned)cbase * HWu + (unsigned)oy * p.W + ox);
    const unsigned rvoff = 4u * ((unsigned)cbase * (HWu >> 2) + (unsigned)(oy >> 1) * (p.W >> 1) + (ox >> 1));
    const unsigned dvoff = 4u * ((unsigned)cbase * (HWu << 2) + (unsigned)(2 * oy) * (unsigned)(2 * p.W) + (unsigned)(2 * ox));
    const bool pairs = p.gsum_rc == 2;
    // A^T = [[1,1,1,0],[0,1,-1,-1]] over nu, then over xi, in conv_wino.hip's order of operations, one accumulator REGISTER at a
    // time (16 AccVGPR reads, 16 additions over nu, then the two pixel rows as packed pairs (column 0, column 1) -- which is also
    // the operand of the 8-byte store and the shape of the residual):
    //   T_xi[0] = (M[xi][0] + M[xi][1]) + M[xi][2], T_xi[1] = (M[xi][1] - M[xi][2]) - M[xi][3];
    //   row 0 = ((T0 + T1) + T2) + residual, row 1 = ((-T2 - T3) + T1) + residual
    // registers 4 gq .. 4 gq + 3 (channels cbase + 8 gq + {0 .. 3}) of both pixel rows: transform, residual, stores, statistics
    auto finish_group = [&](int gq, const f32x2 (&rv)[2][4]) {
      f32x2 v[2][4];                                                  // [pixel row][k] = (column 0, column 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = 4 * gq + k;
        f32x2 T[4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
          T[xi].x = (acc[4 * xi][r] + acc[4 * xi + 1][r]) + acc[4 * xi + 2][r];
          T[xi].y = (acc[4 * xi + 1][r] - acc[4 * xi + 2][r]) - acc[4 * xi + 3][r];
        }
        v[0][k] = ((T[0] + T[1]) + T[2]) + rv[0][k];
        v[1][k] = ((-T[2] - T[3]) + T[1]) + rv[1][k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int row = 0; row < 2; ++row)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, v[row][k]),
                                                rs_out, voff + (unsigned)row * Wb, 4u * (unsigned)(k + 8 * gq) * HWu, 0);
      if (p.gsum) {
        // fused GroupNorm statistics: per pixel row the records conv_wino.hip's two half-waves write (count, sum, M2 about the
        // row's own mean), merged in the same fixed order by conv_stats_store: the tables of the two kernels are identical
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          float a[2], b[2];
#pragma unroll
          for (int e = 0; e < 2; ++e)
            a[e] = half_sum32((v[row][2 * e].x + v[row][2 * e].y) + (v[row][2 * e + 1].x + v[row][2 * e + 1].y));
          const float cnt = pairs ? 128.f : 256.f;
          const float mean0 = pairs ? a[0] * (1.0f / 128.0f) : (a[0] + a[1]) * (1.0f / 256.0f);
          const float mean1 = pairs ? a[1] * (1.0f / 128.0f) : mean0;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float mean = e ? mean1 : mean0;
            float q = 0.f;
#pragma unroll
            for (int k = 2 * e; k < 2 * e + 2; ++k) { const float d0 = v[row][k].x - mean, d1 = v[row][k].y - mean; q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); }
            b[e] = half_sum32(q);
          }
          if ((lane & 31) == 0) {
            const int quad = 8 * mb + 2 * gq + (lane >> 5);
            if (pairs) {
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                float* slot = red + (row * (C::MT / 2) + 2 * quad + e) * 3;
                slot[0] = cnt; slot[1] = a[e]; slot[2] = b[e];
              }
            } else {
              float* slot = red + (row * (C::MT / 4) + quad) * 3;
              slot[0] = cnt; slot[1] = a[0] + a[1]; slot[2] = b[0] + b[1];
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      f32x2 rv[2][4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int dr = k + 8 * gq;
        if (p.res && p.res_mode == RS_DOWN) {      // residual at double resolution, 2 x 2 mean (adm_blocks.py:75-77)
          const unsigned so = 4u * (unsigned)dr * (HWu << 2);
#pragma unroll
          for (int row = 0; row < 2; ++row) {
            const f32x4 ta = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, dvoff + (unsigned)(2 * row) * 8u * (unsigned)p.W, so, 0));
            const f32x4 tb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, dvoff + (unsigned)(2 * row + 1) * 8u * (unsigned)p.W, so, 0));
            rv[row][k] = f32x2{0.25f * ((ta[0] + ta[1]) + (tb[0] + tb[1])), 0.25f * ((ta[2] + ta[3]) + (tb[2] + tb[3]))};
          }
        } else if (p.res && p.res_mode == RS_UP) { // half resolution: the four pixels of the patch share one source pixel
          const float q = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, rvoff, 4u * (unsigned)dr * (HWu >> 2), 0));
          rv[0][k] = rv[1][k] = f32x2{q, q};
        } else {                                   // no residual: a zero-sized descriptor reads zeros
          rv[0][k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_res, voff, 4u * (unsigned)dr * HWu, 0));
          rv[1][k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_res, voff + Wb, 4u * (unsigned)dr * HWu, 0));
        }
      }
      finish_group(gq, rv);
    }
    __builtin_amdgcn_sched_barrier(0);                    // the accumulators are dead from here
    if (p.gsum) {
      __syncthreads();
      conv_stats_store<C, 2>(p, red, n, m0, tile, tiles_img, tid);
      // the next epilogue writes `red` only after a whole tile of barriers
    }
    __builtin_amdgcn_sched_barrier(0);
    // chunk 0's weights for the next tile (the stream's reload at the tile's last chunk fetched them too; assigning them here
    // ends that copy's life at the last MFMA, so the 64 registers are free while the accumulators are transformed)
#pragma unroll
    for (int q = 0; q < 16; ++q) ua[q] = up[q * 64];
    W1_STAMP(tl_epi)
  }
  (void)G;
#ifdef MCEDM_WINO_TIMELINE
  if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 16 + 8] = tl_loop; p.dbg[blockIdx.x * 16 + 9] = tl_epi; }
#endif
  if (p.dbg && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p.dbg[blockIdx.x * 16 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[blockIdx.x * 16 + 6] = __builtin_amdgcn_s_memtime();
    p.dbg[blockIdx.x * 16 + 3] = __builtin_amdgcn_s_memrealtime();
  }
}

static int g_wino1 = -1;      // -1: default (env MCEDM_WINO1, else OFF: measured 9 % slower, see the header); 0 / 1: forced by mcedm_op_set_conv_wino1
void set_conv_wino1(int enable) { g_wino1 = enable; }
static int wino1_env() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_WINO1"); env = e ? atoi(e) : 0; }
  return variant_choice(KV_CONV_WINO1, g_wino1, env);
}

// a (already validated by launch_conv_wino) with Cout % 128 == 0: the one-wave-per-SIMD kernel; -1 when switched off
int try_launch_conv_wino1(const ConvArgs& a, hipStream_t stream) {
  using C = Wino1Cfg;
  if (!wino1_env() || a.Cout % C::MT != 0) return -1;
  const int tiles_x = a.W / WPW, tiles_img = tiles_x * (a.H / WPH);
  const long long total = (long long)a.B * tiles_img;
  MCEDM_REQUIRE(total > 0 && total <= 0x7fffffffLL, "conv_wino1: grid out of range");
  const int Cin = a.Ca + a.Cb, nch = Cin / WKC;
  const int lds_bytes = (C::LDS_ROWS_OFF + 4 * Cin + C::RED_FLOATS) * 4;
  MCEDM_REQUIRE(lds_bytes <= 160 * 1024, "conv_wino1: %d input channels exceed the LDS row table", Cin);
  static std::atomic<bool> attr_set[64];
  static std::atomic<int> ncu[64];
  int dev = 0;
  MCEDM_HIP_TRY(hipGetDevice(&dev));
  MCEDM_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
  if (!attr_set[dev].load(std::memory_order_acquire)) {
    int n_cu = 0;
    MCEDM_HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    ncu[dev].store(n_cu > 0 ? n_cu : 256, std::memory_order_release);
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)conv_wino1_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)conv_wino1_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[dev].store(true, std::memory_order_release);
  }
  static int per_env = -1;
  if (per_env < 0) { const char* e = getenv("MCEDM_WINO_PER"); per_env = e ? atoi(e) : 0; }
  int per = wino_tiles_per_wg(total, tiles_img, ncu[dev].load(std::memory_order_acquire));
  if (per_env > 0 && tiles_img % per_env == 0) per = per_env;
  char name[64] = "";
  if (prof_enabled()) snprintf(name, sizeof(name), "conv_wino1_kernel<%s>", a.resample == RS_UP ? "true" : "false");   // = rocprofv3's name
  const double px = (double)a.B * a.H * a.W;
  ProfScope ps(name, 2.0 * px * a.Cout * (double)Cin * 9,
               4.0 * ((double)a.B * Cin * a.Hs * a.Ws + px * a.Cout * (a.res ? 2 : 1) + (double)a.Cout * Cin * 9), stream);
  if (a.resample == RS_UP)
    hipLaunchKernelGGL((conv_wino1_kernel<true>), dim3((unsigned)(total / per), a.Cout / C::MT), dim3(C::NT), lds_bytes, stream, a, tiles_x,
                       tiles_img, nch, cout_padded(a.Cout) / 32, per);
  else
    hipLaunchKernelGGL((conv_wino1_kernel<false>), dim3((unsigned)(total / per), a.Cout / C::MT), dim3(C::NT), lds_bytes, stream, a, tiles_x,
                       tiles_img, nch, cout_padded(a.Cout) / 32, per);
  MCEDM_LAUNCH_CHECK("conv_wino1_kernel");
  if (a.gsum_tiles) *a.gsum_tiles = SumTiles{tiles_img, tiles_x, WPH, WPW, a.gsum_rc == 2 ? 2 : 4};
  return MCEDM_OK;
}

}  // namespace mcedm
