// wgrad_wino.hip -- K10a': weight (and bias) gradient of the un-resampled 3x3 convolution in Winograd F(3x3, 2x2) form.
//
//   dW[co][ci][a][b] = sum_{n, y, x} dY[n][co][y][x] * X'[n][ci][y + a - 1][x + b - 1]            (adm_blocks.py:78-79, backward)
//
// is, per 2x2 tile of dY and the 4x4 tile of X' around it, a correlation with 3x3 outputs and a 2x2 "filter": 16 multiplies per
// (co, ci) and tile instead of 36 (Lavin & Gray 2016, the F(3x3, 2x2) of their section 4.3 "training"):
//
//   dW = A^T [ sum_tiles (G dY_t G^T) o (B^T X'_t B) ] A        B^T as in the forward kernel (conv_wino.hip), G 4x2, A^T 3x4
//
// The sum over tiles is the K dimension of 16 independent GEMMs M_p[co][ci] (p = Winograd position (xi, nu)), K = B*H*W/4.
// tools/wino_wgrad_error.py: fp32 error against fp64 <= 2x the direct form's, < 1 % of the bar gradients are held to.
//
// Mapping to one CU (one 512-thread workgroup, 8 waves, 2 per SIMD):
//   * a workgroup owns ONE xi (a row of the 4x4 position grid), 128 output x 128 input channels and all four nu:
//     4 x 4 x 4 = 64 accumulator blocks of 32 x 32 = half the CU's register file.  Wave w: nu = w >> 1, output-channel half
//     w & 1 (two 32-blocks) x four input-channel blocks = 8 blocks = 128 accumulator registers.  Fixing xi per workgroup
//     makes the transform work per MFMA a quarter of the forward kernel's: a workgroup needs only TWO rows of each 4x4 /
//     2x2 tile (B^T and G have two non-zeros per row), and both operands feed 128 channels of the other.
//   * K loop in stages of 16 tiles = 32 output pixels of one image row pair.  Per stage and wave 64 MFMAs (32x32x2: two tiles
//     per instruction).  Both operands are transformed in registers straight from global memory -- X': one (channel, tile)
//     per thread and task, two 16-byte loads at the tile's (unaligned) first column, 6 vector instructions, 4 LDS dwords;
//     dY: one (channel, aligned quad = two tiles) per task -- and written to LDS in MFMA fragment order
//     [nu][channel][16 tiles], 16-byte slots XOR-swizzled by (channel >> 2) & 3 so that the ds_read_b128 fragment reads and the
//     transform's writes are bank-conflict free without padding: 64 KB per stage, double-buffered, ONE barrier per stage.
//     The registers of stage g + 1 are committed and re-requested for stage g + 2 in slices between the MFMAs of stage g.
//   * every split of the K range stores its partial blocks to [split][16][CoP][CiP]; wgrad_wino_reduce_kernel adds the
//     splits in fp64 in a fixed order, applies the halves dropped from G and A^T . A, and writes the reference layout:
//     no atomics, bitwise reproducible.  The four xi-siblings of a split read the same rows: their workgroup ids are
//     chosen so that they share an XCD (one L2).
//   * the bias gradient (sum of dY) falls out of the xi = 1 workgroup's row sums dY[2ty] + dY[2ty+1].
#include <atomic>
#include <cstdlib>

#include "bwd.hpp"
#include "conv_tile.hpp"
#include "prof.hpp"

namespace mcedm {

constexpr int GW_T = 16;                       // tiles per stage
constexpr int GW_CB = 128;                     // channels per block (output and input)
constexpr int GW_PLANE = GW_CB * GW_T;         // floats per (nu, operand)
constexpr int GW_OP = 4 * GW_PLANE;            // floats per operand and stage
constexpr int GW_STAGE = 2 * GW_OP;            // X' then dY: 16384 floats = 64 KB

struct WgWinoArgs {
  const float* dy; const float* x;             // [B][Co][H][W], [B][Ci][H][W] (X' materialised)
  float* part;                                 // [split][16][cop][cip]
  float* dbp;                                  // [split][cop] or null
  int Co, Ci, B, H, W;
  int cob, cib, cop, cip;                      // 128-channel blocks; padded channel counts of the scratch
  int nact, per, total;                        // splits that own stages, stages per split, stages in all
  int nseg, th;                                // 32-pixel segments per row, tile rows
  unsigned long long* dbg;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int XI>
__device__ __forceinline__ void wgw_body(const WgWinoArgs& p, float* lds, int split, int cb_o, int cb_i) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nu = wave >> 1, coh = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int s_begin = split * p.per;
  const int s_end = s_begin + p.per < p.total ? s_begin + p.per : p.total;
  if (s_begin >= s_end) return;
  const int G = s_end - s_begin;
  const unsigned HW = (unsigned)p.H * p.W;
  const int co0 = cb_o * GW_CB, ci0 = cb_i * GW_CB;
  // the X' descriptor starts 16 bytes in FRONT of the tensor (the launcher guarantees they are readable: x lies inside a scratch
  // allocation): the first tile of a row starts at column -1, which for the tensor's very first row is 4 bytes in front of it.
  // Every out-of-image column is zeroed by the lane masks below; past the END of the tensor the range check returns zeros.
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x - 4, 4u * (unsigned)p.B * p.Ci * HW + 16u);
  const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.dy, 4u * (unsigned)p.B * p.Co * HW);
  constexpr bool YA = XI != 3, YB = XI != 0;                  // which of the tile's two dY rows this xi reads
  constexpr int NYR = (YA && YB) ? 2 : 1;

  // ---- lane constants of the transform tasks
  // X' task j (4 per thread): tile xt = tid & 15 of channel (tid >> 4) + 32 j; first column 2 xt - 1 (+ 4 floats: the descriptor's lead)
  const int xt = tid & 15;
  const unsigned xlb = 4u * ((unsigned)(tid >> 4) * HW + (unsigned)(2 * xt + 3));
  const int xdst = (tid >> 4) * GW_T + 4 * ((xt >> 2) ^ (wave & 3)) + (xt & 3);      // (channel >> 2) & 3 == wave & 3 for every j
  // dY task j (2 per thread): aligned quad yq = tid & 7 (tiles 2 yq, 2 yq + 1) of channel (tid >> 3) + 64 j
  const int yq = tid & 7;
  const unsigned ylb = 4u * ((unsigned)(tid >> 3) * HW + 4u * yq);
  const int ydst = GW_OP + (tid >> 3) * GW_T + 4 * ((yq >> 1) ^ ((tid >> 5) & 3)) + 2 * (yq & 1);
  // MFMA fragments: lane (l31, h), k-group grp: slot 2 grp + h of its channel
  int fo[2];
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) fo[grp] = l31 * GW_T + 4 * ((2 * grp + h) ^ ((l31 >> 2) & 3));

  // ---- geometry of the stage that is loaded next (wave-uniform), masks of the stage waiting in registers
  int lst = s_begin;
  int lseg = lst % p.nseg, lty = (lst / p.nseg) % p.th, ln = lst / (p.nseg * p.th);
  unsigned g_xa, g_xb, g_y; float g_mrow, g_sl, g_sr;
  auto set_geo = [&]() {
    int ya, yb; float mrow = 1.f;
    if (XI == 0) { ya = 2 * lty - 1; yb = 2 * lty + 1; if (ya < 0) { ya = 0; mrow = 0.f; } }
    else if (XI == 3) { ya = 2 * lty; yb = 2 * lty + 2; if (yb > p.H - 1) { yb = p.H - 1; mrow = 0.f; } }
    else { ya = 2 * lty; yb = 2 * lty + 1; }
    const unsigned xplane = (unsigned)(ln * p.Ci + ci0) * HW + (unsigned)lseg * 32u;
    g_xa = 4u * (xplane + (unsigned)ya * p.W);
    g_xb = 4u * (xplane + (unsigned)yb * p.W);
    g_y = 4u * ((unsigned)(ln * p.Co + co0) * HW + (unsigned)(2 * lty + (YA ? 0 : 1)) * p.W + (unsigned)lseg * 32u);
    g_mrow = mrow; g_sl = lseg == 0 ? 0.f : 1.f; g_sr = lseg == p.nseg - 1 ? 0.f : 1.f;
  };
  auto advance = [&]() {                                      // past the end of this split's range: stay on its last stage
    if (lst + 1 < s_end) {
      ++lst;
      if (++lseg == p.nseg) { lseg = 0; if (++lty == p.th) { lty = 0; ++ln; } }
    }
    set_geo();
  };
  set_geo();
  float c_mrow = 1.f;
  unsigned ml = ~0u, mr = ~0u;                                  // bit masks: what lies outside the image may be anything, NaN included
  auto take_masks = [&]() { c_mrow = g_mrow; ml = (xt == 0 && g_sl == 0.f) ? 0u : ~0u; mr = (xt == GW_T - 1 && g_sr == 0.f) ? 0u : ~0u; };

  f32x4 xr[4][2], yr[2][NYR];
  auto load_x = [&](int j) {
    const unsigned cj = 4u * 32u * (unsigned)j * HW;
    xr[j][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xlb + (g_xa + cj), 0, 0));
    xr[j][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xlb + (g_xb + cj), 0, 0));
  };
  auto load_y = [&](int j) {
    const unsigned vo = ylb + (g_y + 4u * 64u * (unsigned)j * HW);
    yr[j][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, vo, 0, 0));
    if (NYR == 2) yr[j][NYR - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, vo, 4u * (unsigned)p.W, 0));
  };
  // B^T row xi down the tile's rows, then B^T along its columns: 4 Winograd-domain values of (channel, tile) -> LDS
  auto commit_x = [&](int j, float* sb) {
    const f32x4 a = xr[j][0], b = xr[j][1];
    f32x4 r;
    if (XI == 0) r = a * c_mrow - b;
    else if (XI == 1) r = a + b;
    else if (XI == 2) r = b - a;
    else r = a - b * c_mrow;
    // (through scalar copies: __builtin_bit_cast applied to the vector-element lvalue r[3] reads element 0 with this hipcc)
    const float e0 = r[0], e3 = r[3];
    const float r0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, e0) & ml);
    const float r3 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, e3) & mr);
    float* d = sb + xdst + j * 32 * GW_T;
    d[0 * GW_PLANE] = r0 - r[2];
    d[1 * GW_PLANE] = r[1] + r[2];
    d[2 * GW_PLANE] = r[2] - r[1];
    d[3 * GW_PLANE] = r[1] - r3;
  };
  // G' = [[1,0],[1,1],[1,-1],[0,1]] (the halves of G are applied by the reduction): row xi, then along the columns of both tiles
  float bsum[2] = {0.f, 0.f};
  auto commit_y = [&](int j, float* sb, bool bias) {
    f32x4 s;
    if (XI == 0 || XI == 3) s = yr[j][0];
    else if (XI == 1) s = yr[j][0] + yr[j][NYR - 1];
    else s = yr[j][0] - yr[j][NYR - 1];
    float* d = sb + ydst + j * 64 * GW_T;
    *reinterpret_cast<f32x2*>(d + 0 * GW_PLANE) = f32x2{s[0], s[2]};
    *reinterpret_cast<f32x2*>(d + 1 * GW_PLANE) = f32x2{s[0] + s[1], s[2] + s[3]};
    *reinterpret_cast<f32x2*>(d + 2 * GW_PLANE) = f32x2{s[0] - s[1], s[2] - s[3]};
    *reinterpret_cast<f32x2*>(d + 3 * GW_PLANE) = f32x2{s[1], s[3]};
    if (XI == 1 && bias) bsum[j] += (s[0] + s[1]) + (s[2] + s[3]);
  };
  const bool do_bias = XI == 1 && cb_i == 0 && p.dbp != nullptr;

  // ---- prologue: stage 0 -> buffer 0, stage 1 -> registers
#pragma unroll
  for (int j = 0; j < 4; ++j) load_x(j);
#pragma unroll
  for (int j = 0; j < 2; ++j) load_y(j);
  take_masks();
  advance();
#pragma unroll
  for (int j = 0; j < 4; ++j) commit_x(j, lds);
#pragma unroll
  for (int j = 0; j < 2; ++j) commit_y(j, lds, do_bias);
#pragma unroll
  for (int j = 0; j < 4; ++j) load_x(j);
#pragma unroll
  for (int j = 0; j < 2; ++j) load_y(j);
  take_masks();
  advance();

  f32x16 acc[2][4];
  {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("" : "+v"(z));
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, z, zero16, 0, 0, 0);
      }
  }
  __syncthreads();

  // ---- the K loop: one trip = one stage.  8 slots of [B fragment of the next slot | 8 MFMAs | one slice of side work]:
  // slots 0-3 commit X' task j of stage g + 1 and request it again for stage g + 2, slots 4-5 the same for the dY tasks, slot 6
  // moves the geometry on.  A slice touches only the OTHER stage buffer: one barrier per trip.
  for (int g = 0; g < G; ++g) {
    const int cur = g & 1;
    const float* fbuf = lds + cur * GW_STAGE;
    float* nbuf = lds + (cur ^ 1) * GW_STAGE;
    const bool cvalid = g + 1 < G;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      const float* xf = fbuf + nu * GW_PLANE + fo[grp];
      const float* yf = fbuf + GW_OP + nu * GW_PLANE + coh * 64 * GW_T + fo[grp];
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(yf);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(yf + 32 * GW_T);
      f32x4 bc = *reinterpret_cast<const f32x4*>(xf);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 bn = bc;
        if (j < 3) bn = *reinterpret_cast<const f32x4*>(xf + (j + 1) * 32 * GW_T);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], bc[s], acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], bc[s], acc[1][j], 0, 0, 0);
        }
        const int slot = grp * 4 + j;
        if (slot < 4) { commit_x(slot, nbuf); load_x(slot); }
        else if (slot < 6) { commit_y(slot - 4, nbuf, do_bias && cvalid); load_y(slot - 4); }
        if (j < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x286, 3, 0);            // VALU | SALU | DS write
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (slot == 6) { take_masks(); advance(); }
        __builtin_amdgcn_sched_barrier(0);
        bc = bn;
      }
    }
    __syncthreads();
  }

  // ---- this split's partial blocks -> [split][xi * 4 + nu][co][ci] (plain stores; 128-byte segments per half-wave)
  float* out = p.part + ((size_t)split * 16 + XI * 4 + nu) * p.cop * p.cip;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (2 * coh + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ci = ci0 + 32 * j + l31;
        out[(size_t)co * p.cip + ci] = acc[i][j][r];
      }
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v = bsum[j];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
      if (yq == 0) p.dbp[(size_t)split * p.cop + co0 + (tid >> 3) + 64 * j] = v;
    }
  }
}

__global__ __launch_bounds__(512, 1) void wgrad_wino_kernel(const WgWinoArgs p) {
  extern __shared__ float lds[];
  // workgroup id -> (split, sibling): the 4 * cib * cob siblings of a split sit 8 ids apart = on the same XCD (ids go round-robin
  // over the 8 XCDs), so that three of their four reads of the same dY / X' rows hit that XCD's L2
  const int nsib = 4 * p.cib * p.cob;
  const int bid = blockIdx.x;
  const int grp8 = bid / (8 * nsib), rem = bid - grp8 * 8 * nsib;
  const int sib = rem >> 3, split = grp8 * 8 + (rem & 7);
  if (split >= p.nact) return;
  const int xi = sib & 3, blk = sib >> 2;
  const int cb_i = blk % p.cib, cb_o = blk / p.cib;
  switch (xi) {
    case 0: wgw_body<0>(p, lds, split, cb_o, cb_i); break;
    case 1: wgw_body<1>(p, lds, split, cb_o, cb_i); break;
    case 2: wgw_body<2>(p, lds, split, cb_o, cb_i); break;
    default: wgw_body<3>(p, lds, split, cb_o, cb_i); break;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The 1x1 weight gradient on the same machinery (round 5):  dW[co][ci] = sum_px dY[co][px] X[ci][px]  is a plain GEMM over the
// pixels.  The direct kernel runs it at 0.45 of the matrix rate (a 64 x 64 block per workgroup: 32 MFMAs per wave between two
// barriers); here a workgroup owns 128 x 128 channels and the stage loop, LDS layout and slot structure of wgw_body, with the
// four Winograd columns nu standing for the four PIXELS of an aligned quad -- both "transforms" are the identity: an X task is
// one aligned 16-byte load and four LDS dwords, a stage is 64 consecutive pixels of a sample's plane.  The four accumulator sets
// are four interleaved partial sums; every (split, nu) is one slice [co][ci] of the direct kernel's scratch layout, so
// wgrad_reduce_kernel (wgrad_mfma.hip) adds them unchanged (fp64, fixed order).  The operand is read in place, from xa or xb.
__global__ __launch_bounds__(512, 1) void wgrad_gemm1_kernel(const WgWinoArgs p, const float* __restrict__ xb, int Ca) {
  extern __shared__ float lds[];
  const int nsib = p.cib * p.cob;
  const int bid = blockIdx.x;
  const int grp8 = bid / (8 * nsib), rem = bid - grp8 * 8 * nsib;
  const int sib = rem >> 3, split = grp8 * 8 + (rem & 7);
  if (split >= p.nact) return;
  const int cb_i = sib % p.cib, cb_o = sib / p.cib;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nu = wave >> 1, coh = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int s_begin = split * p.per;
  const int s_end = s_begin + p.per < p.total ? s_begin + p.per : p.total;
  if (s_begin >= s_end) return;
  const int G = s_end - s_begin;
  const unsigned HW = (unsigned)p.H * p.W;
  const int co0 = cb_o * GW_CB, ci0 = cb_i * GW_CB;
  // this workgroup's 128 input channels lie in ONE source (Ca % 128 == 0)
  const bool in_a = ci0 < Ca;
  const int Csrc = in_a ? Ca : p.Ci - Ca, cs0 = in_a ? ci0 : ci0 - Ca;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(in_a ? p.x : xb, 4u * (unsigned)p.B * Csrc * HW);
  const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.dy, 4u * (unsigned)p.B * p.Co * HW);

  const int xt = tid & 15;
  const unsigned xlb = 4u * ((unsigned)(tid >> 4) * HW + 4u * (unsigned)xt);
  const int xdst = (tid >> 4) * GW_T + 4 * ((xt >> 2) ^ (wave & 3)) + (xt & 3);
  const int yq = tid & 7;
  const unsigned ylb = 4u * ((unsigned)(tid >> 3) * HW + 8u * yq);
  const int ydst = GW_OP + (tid >> 3) * GW_T + 4 * ((yq >> 1) ^ ((tid >> 5) & 3)) + 2 * (yq & 1);
  int fo[2];
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) fo[grp] = l31 * GW_T + 4 * ((2 * grp + h) ^ ((l31 >> 2) & 3));

  int lst = s_begin;
  int lseg = lst % p.nseg, ln = lst / p.nseg;
  unsigned g_x, g_y;
  auto set_geo = [&]() {
    g_x = 4u * ((unsigned)(ln * Csrc + cs0) * HW + (unsigned)lseg * 64u);
    g_y = 4u * ((unsigned)(ln * p.Co + co0) * HW + (unsigned)lseg * 64u);
  };
  auto advance = [&]() {
    if (lst + 1 < s_end) {
      ++lst;
      if (++lseg == p.nseg) { lseg = 0; ++ln; }
    }
    set_geo();
  };
  set_geo();

  f32x4 xr[4], yr[2][2];
  auto load_x = [&](int j) { xr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xlb + (g_x + 4u * 32u * (unsigned)j * HW), 0, 0)); };
  auto load_y = [&](int j) {
    const unsigned vo = ylb + (g_y + 4u * 64u * (unsigned)j * HW);
    yr[j][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, vo, 0, 0));
    yr[j][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, vo, 16, 0));
  };
  auto commit_x = [&](int j, float* sb) {
    float* d = sb + xdst + j * 32 * GW_T;
    d[0 * GW_PLANE] = xr[j][0];
    d[1 * GW_PLANE] = xr[j][1];
    d[2 * GW_PLANE] = xr[j][2];
    d[3 * GW_PLANE] = xr[j][3];
  };
  float bsum[2] = {0.f, 0.f};
  auto commit_y = [&](int j, float* sb, bool bias) {
    const f32x4 a = yr[j][0], b = yr[j][1];
    float* d = sb + ydst + j * 64 * GW_T;
    *reinterpret_cast<f32x2*>(d + 0 * GW_PLANE) = f32x2{a[0], b[0]};
    *reinterpret_cast<f32x2*>(d + 1 * GW_PLANE) = f32x2{a[1], b[1]};
    *reinterpret_cast<f32x2*>(d + 2 * GW_PLANE) = f32x2{a[2], b[2]};
    *reinterpret_cast<f32x2*>(d + 3 * GW_PLANE) = f32x2{a[3], b[3]};
    if (bias) bsum[j] += ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
  };
  const bool do_bias = cb_i == 0 && p.dbp != nullptr;

#pragma unroll
  for (int j = 0; j < 4; ++j) load_x(j);
#pragma unroll
  for (int j = 0; j < 2; ++j) load_y(j);
  advance();
#pragma unroll
  for (int j = 0; j < 4; ++j) commit_x(j, lds);
#pragma unroll
  for (int j = 0; j < 2; ++j) commit_y(j, lds, do_bias);
#pragma unroll
  for (int j = 0; j < 4; ++j) load_x(j);
#pragma unroll
  for (int j = 0; j < 2; ++j) load_y(j);
  advance();

  f32x16 acc[2][4];
  {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("" : "+v"(z));
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, z, zero16, 0, 0, 0);
      }
  }
  __syncthreads();

  for (int g = 0; g < G; ++g) {
    const int cur = g & 1;
    const float* fbuf = lds + cur * GW_STAGE;
    float* nbuf = lds + (cur ^ 1) * GW_STAGE;
    const bool cvalid = g + 1 < G;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      const float* xf = fbuf + nu * GW_PLANE + fo[grp];
      const float* yf = fbuf + GW_OP + nu * GW_PLANE + coh * 64 * GW_T + fo[grp];
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(yf);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(yf + 32 * GW_T);
      f32x4 bc = *reinterpret_cast<const f32x4*>(xf);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 bn = bc;
        if (j < 3) bn = *reinterpret_cast<const f32x4*>(xf + (j + 1) * 32 * GW_T);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], bc[s], acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], bc[s], acc[1][j], 0, 0, 0);
        }
        const int slot = grp * 4 + j;
        if (slot < 4) { commit_x(slot, nbuf); load_x(slot); }
        else if (slot < 6) { commit_y(slot - 4, nbuf, do_bias && cvalid); load_y(slot - 4); }
        if (j < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x286, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (slot == 6) advance();
        __builtin_amdgcn_sched_barrier(0);
        bc = bn;
      }
    }
    __syncthreads();
  }

  // slice (split, nu) of the direct kernel's scratch: [split * 4 + nu][co][ci]
  float* out = p.part + ((size_t)split * 4 + nu) * p.cop * p.cip;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (2 * coh + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ci = ci0 + 32 * j + l31;
        out[(size_t)co * p.cip + ci] = acc[i][j][r];
      }
  if (do_bias) {                                              // slice split * 4 carries the bias partial, the other three zeros
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v = bsum[j];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
      if (yq == 0) {
        const int co = co0 + (tid >> 3) + 64 * j;
        p.dbp[((size_t)split * 4 + 0) * p.cop + co] = v;
        p.dbp[((size_t)split * 4 + 1) * p.cop + co] = 0.f;
        p.dbp[((size_t)split * 4 + 2) * p.cop + co] = 0.f;
        p.dbp[((size_t)split * 4 + 3) * p.cop + co] = 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The 64-channel form (the reference's own ch = 64 network, adm_edm_mcedm_res32: every 3x3 conv is 64 -> 64 or 128 -> 64).
// With 64 x 64 channels per workgroup the whole 4 x 4 position grid fits one CU: wave w owns xi = w >> 1, 32 output channels
// (w & 1) x 64 input channels x four nu = 8 accumulator blocks.  A stage is 8 tiles (16 output pixels of one row pair: W % 16
// == 0); an X' task is (channel, tile) with all four rows of the 4x4 tile (16 values out), a dY task is (channel, quad, half of
// the xi rows).  LDS [position 16][channel 64][8 tiles], the two 16-byte slots of a channel swapped by (channel >> 3) & 1:
// the same 64 KB per stage, the same conflict-free reads and writes, one barrier per stage (32 MFMAs per wave).
constexpr int G6_T = 8;
constexpr int G6_CB = 64;
constexpr int G6_PLANE = G6_CB * G6_T;         // floats per (position, operand)
constexpr int G6_OP = 16 * G6_PLANE;           // floats per operand and stage
constexpr int G6_STAGE = 2 * G6_OP;            // 64 KB

__global__ __launch_bounds__(512, 1) void wgrad_wino64_kernel(const WgWinoArgs p) {
  extern __shared__ float lds[];
  const int nsib = p.cib * p.cob;
  const int bid = blockIdx.x;
  const int grp8 = bid / (8 * nsib), rem = bid - grp8 * 8 * nsib;
  const int sib = rem >> 3, split = grp8 * 8 + (rem & 7);
  if (split >= p.nact) return;
  const int cb_i = sib % p.cib, cb_o = sib / p.cib;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave >> 1, coh = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int s_begin = split * p.per;
  const int s_end = s_begin + p.per < p.total ? s_begin + p.per : p.total;
  if (s_begin >= s_end) return;
  const int G = s_end - s_begin;
  const unsigned HW = (unsigned)p.H * p.W;
  const unsigned W4 = 4u * (unsigned)p.W;
  const int co0 = cb_o * G6_CB, ci0 = cb_i * G6_CB;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x - 4, 4u * (unsigned)p.B * p.Ci * HW + 16u);      // see wgw_body
  const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.dy, 4u * (unsigned)p.B * p.Co * HW);

  // X' task: tile xt = tid & 7 of channel tid >> 3 (a wave: 8 channels, (channel >> 3) & 1 == wave & 1)
  const int xt = tid & 7;
  const unsigned xlb = 4u * ((unsigned)(tid >> 3) * HW + (unsigned)(2 * xt + 3));
  const int xdst = (tid >> 3) * G6_T + 4 * ((xt >> 2) ^ (wave & 1)) + (xt & 3);
  // dY task: quad yq = tid & 3 (tiles 2 yq, 2 yq + 1) of channel (tid & 255) >> 2; waves 0-3 write the rows xi = 0, 1, waves 4-7 xi = 3, 2
  const int yq = tid & 3, yc = (tid & 255) >> 2;
  const int yh = wave >> 2;
  const unsigned ylb = 4u * ((unsigned)yc * HW + 4u * yq);
  const int ydst = G6_OP + yc * G6_T + 4 * ((yq >> 1) ^ ((yc >> 3) & 1)) + 2 * (yq & 1);
  const int xiP = yh ? 3 : 0, xiQ = yh ? 2 : 1;
  // MFMA fragments: lane (l31, h) reads slot h of its channel: k-step s <-> tile 4 h + s
  const int fo = l31 * G6_T + 4 * (h ^ ((l31 >> 3) & 1));

  int lst = s_begin;
  int lseg = lst % p.nseg, lty = (lst / p.nseg) % p.th, ln = lst / (p.nseg * p.th);
  unsigned g_x1, g_up, g_dn, g_y; float g_mt, g_mb; bool g_l, g_r;
  auto set_geo = [&]() {
    g_x1 = 4u * ((unsigned)(ln * p.Ci + ci0) * HW + (unsigned)(2 * lty) * p.W + (unsigned)lseg * 16u);   // row 2 ty
    g_up = lty == 0 ? 0u : W4;               // row 2 ty - 1 = g_x1 - g_up (the top tile row reads row 0 again and drops it)
    g_dn = lty == p.th - 1 ? W4 : 2u * W4;   // row 2 ty + 2 = g_x1 + g_dn (the bottom tile row reads row H - 1 again)
    g_mt = lty == 0 ? 0.f : 1.f; g_mb = lty == p.th - 1 ? 0.f : 1.f;
    g_y = 4u * ((unsigned)(ln * p.Co + co0) * HW + (unsigned)(2 * lty) * p.W + (unsigned)lseg * 16u);
    g_l = lseg == 0; g_r = lseg == p.nseg - 1;
  };
  auto advance = [&]() {
    if (lst + 1 < s_end) {
      ++lst;
      if (++lseg == p.nseg) { lseg = 0; if (++lty == p.th) { lty = 0; ++ln; } }
    }
    set_geo();
  };
  set_geo();
  float c_mt = 1.f, c_mb = 1.f;
  unsigned ml = ~0u, mr = ~0u;
  auto take_masks = [&]() { c_mt = g_mt; c_mb = g_mb; ml = (xt == 0 && g_l) ? 0u : ~0u; mr = (xt == G6_T - 1 && g_r) ? 0u : ~0u; };

  f32x4 xr[4], yr[2];
  auto load_x = [&]() {
    xr[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xlb + (g_x1 - g_up), 0, 0));
    xr[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xlb + g_x1, 0, 0));
    xr[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xlb + (g_x1 + W4), 0, 0));
    xr[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xlb + (g_x1 + g_dn), 0, 0));
  };
  auto load_y = [&]() {
    yr[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, ylb + g_y, 0, 0));
    yr[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, ylb + (g_y + W4), 0, 0));
  };
  auto x_row = [&](const f32x4 r, float* d) {                 // B^T along the columns of one transformed row -> nu = 0..3
    const float e0 = r[0], e3 = r[3];
    const float r0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, e0) & ml);
    const float r3 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, e3) & mr);
    d[0 * G6_PLANE] = r0 - r[2];
    d[1 * G6_PLANE] = r[1] + r[2];
    d[2 * G6_PLANE] = r[2] - r[1];
    d[3 * G6_PLANE] = r[1] - r3;
  };
  auto commit_x = [&](float* sb) {
    float* d = sb + xdst;
    x_row(xr[0] * c_mt - xr[2], d);
    x_row(xr[1] + xr[2], d + 4 * G6_PLANE);
    x_row(xr[2] - xr[1], d + 8 * G6_PLANE);
    x_row(xr[1] - xr[3] * c_mb, d + 12 * G6_PLANE);
  };
  auto y_row = [&](const f32x4 s, float* d) {                 // G' along the columns of both tiles
    *reinterpret_cast<f32x2*>(d + 0 * G6_PLANE) = f32x2{s[0], s[2]};
    *reinterpret_cast<f32x2*>(d + 1 * G6_PLANE) = f32x2{s[0] + s[1], s[2] + s[3]};
    *reinterpret_cast<f32x2*>(d + 2 * G6_PLANE) = f32x2{s[0] - s[1], s[2] - s[3]};
    *reinterpret_cast<f32x2*>(d + 3 * G6_PLANE) = f32x2{s[1], s[3]};
  };
  float bsum = 0.f;
  auto commit_y = [&](float* sb, bool bias) {
    const f32x4 a = yr[0], b = yr[1];
    const f32x4 P = yh ? b : a;                               // rows xi = 0 / 3 of G' dY
    const f32x4 Q = yh ? a - b : a + b;                       // rows xi = 1 / 2
    y_row(P, sb + ydst + xiP * 4 * G6_PLANE);
    y_row(Q, sb + ydst + xiQ * 4 * G6_PLANE);
    if (bias) bsum += (Q[0] + Q[1]) + (Q[2] + Q[3]);
  };
  const bool do_bias = yh == 0 && cb_i == 0 && p.dbp != nullptr;

  load_x(); load_y();
  take_masks();
  advance();
  commit_x(lds); commit_y(lds, do_bias);
  load_x(); load_y();
  take_masks();
  advance();

  f32x16 acc[4][2];
  {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("" : "+v"(z));
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, z, zero16, 0, 0, 0);
      }
  }
  __syncthreads();

  // one trip = one stage: 4 slots (nu) of [fragments of the next slot | 8 MFMAs | a slice of side work on the OTHER buffer]
  for (int g = 0; g < G; ++g) {
    const int cur = g & 1;
    const float* fx = lds + cur * G6_STAGE + xi * 4 * G6_PLANE + fo;
    const float* fy = fx + G6_OP + coh * 32 * G6_T;
    float* nbuf = lds + (cur ^ 1) * G6_STAGE;
    const bool cvalid = g + 1 < G;
    f32x4 a = *reinterpret_cast<const f32x4*>(fy);
    f32x4 b0 = *reinterpret_cast<const f32x4*>(fx);
    f32x4 b1 = *reinterpret_cast<const f32x4*>(fx + 32 * G6_T);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      f32x4 an = a, b0n = b0, b1n = b1;
      if (nu < 3) {
        an = *reinterpret_cast<const f32x4*>(fy + (nu + 1) * G6_PLANE);
        b0n = *reinterpret_cast<const f32x4*>(fx + (nu + 1) * G6_PLANE);
        b1n = *reinterpret_cast<const f32x4*>(fx + (nu + 1) * G6_PLANE + 32 * G6_T);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0[s], acc[nu][0], 0, 0, 0);
        acc[nu][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b1[s], acc[nu][1], 0, 0, 0);
      }
      if (nu == 0) { commit_x(nbuf); load_x(); }
      else if (nu == 1) { commit_y(nbuf, do_bias && cvalid); load_y(); }
      if (nu < 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (nu == 0) __builtin_amdgcn_sched_group_barrier(0x286, 9, 0);           // VALU | SALU | DS write
        else __builtin_amdgcn_sched_group_barrier(0x286, 5, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (nu == 2) { take_masks(); advance(); }
      __builtin_amdgcn_sched_barrier(0);
      a = an; b0 = b0n; b1 = b1n;
    }
    __syncthreads();
  }

  float* out = p.part + ((size_t)split * 16 + xi * 4) * p.cop * p.cip;
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + coh * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ci = ci0 + 32 * j + l31;
        out[((size_t)nu * p.cop + co) * p.cip + ci] = acc[nu][j][r];
      }
  if (do_bias) {
    float v = bsum;
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2);
    if (yq == 0) p.dbp[(size_t)split * p.cop + co0 + yc] = v;
  }
}

// dW[co][ci][a][b] = sum_{xi, nu} A^T[a][xi] A^T[b][nu] c[xi] c[nu] sum_split part[split][xi][nu][co][ci]
//   A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]]   (the last column carries the sign of B^T's last row (0,1,0,-1))
//   c = (1, 1/2, 1/2, 1): the halves of G = [[1,0],[1/2,1/2],[1/2,-1/2],[0,1]] that the kernel's transform leaves out.
// One workgroup per (co, 64 input channels, slice of the splits): thread (p, ci) sums position p over the splits slice, slice + nsl,
// ... in fp64 (eight interleaved chains: a fixed order).  nsl == 1 (layers with >= 256 such blocks): the 16 sums meet in LDS and
// threads (tap, ci) apply A^T . A.  nsl > 1 (a 64 x 64 layer has only 64 blocks: 1 TB/s): the slice sums go to `mid`
// [slice][16][co][ci] in fp64 and wgrad_wino_finish_kernel adds them in slice order.  db[co] = sum_split dbp.
__device__ __forceinline__ void wgw_taps(const double (*m)[64], int tid, int cl, int co, int ci, int Cin, float* __restrict__ dw) {
  if (tid < 9 * 64 && ci < Cin) {
    const int tap = tid >> 6, a = tap / 3, b = tap % 3;
    // rows of A^T as (xi, sign) lists: a = 0: +0 +1 +2; a = 1: +1 -2; a = 2: +1 +2 -3
    double r = 0.0;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const double wa = a == 0 ? (x < 3 ? 1.0 : 0.0) : a == 1 ? (x == 1 ? 1.0 : x == 2 ? -1.0 : 0.0) : (x == 0 ? 0.0 : x == 3 ? -1.0 : 1.0);
      if (wa == 0.0) continue;
      double t = 0.0;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const double wb = b == 0 ? (n < 3 ? 1.0 : 0.0) : b == 1 ? (n == 1 ? 1.0 : n == 2 ? -1.0 : 0.0) : (n == 0 ? 0.0 : n == 3 ? -1.0 : 1.0);
        if (wb != 0.0) t += wb * m[x * 4 + n][cl];
      }
      r += wa * t;
    }
    dw[((size_t)co * Cin + ci) * 9 + tap] = (float)r;
  }
}
__device__ __forceinline__ double wgw_pos_scale(int pos) {
  const int xi = pos >> 2, nu = pos & 3;
  return ((xi == 1 || xi == 2) ? 0.5 : 1.0) * ((nu == 1 || nu == 2) ? 0.5 : 1.0);
}
// one wave: lane k adds the splits k, k + 64, ... in order, then a fixed shuffle tree (a single thread walking 256 splits was the
// critical path of the whole reduction: 0.2 us per dependent L2 read)
__device__ __forceinline__ void wgw_bias(const float* __restrict__ dbp, float* __restrict__ db, int co, int cop, int nact, int lane) {
  double s = 0.0;
  for (int k = lane; k < nact; k += 64) s += (double)dbp[(size_t)k * cop + co];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) db[co] = (float)s;
}

__global__ __launch_bounds__(1024) void wgrad_wino_reduce_kernel(const float* __restrict__ part, const float* __restrict__ dbp,
                                                                 float* __restrict__ dw, float* __restrict__ db, double* __restrict__ mid,
                                                                 int Cout, int Cin, int cop, int cip, int nact, int nsl) {
  __shared__ double m[16][64];
  const int tid = threadIdx.x, cl = tid & 63, pos = tid >> 6;
  const int citiles = cip / 64;
  const int blk = blockIdx.x / nsl, sl = blockIdx.x - blk * nsl;
  const int co = blk / citiles, ci = (blk % citiles) * 64 + cl;
  const size_t block = (size_t)16 * cop * cip;
  const float* src = part + ((size_t)pos * cop + co) * cip + ci;
  // the eight loads of a trip are independent and stay in flight together (four chains: 2.7 TB/s on the 67 MB of partial blocks)
  double ch[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const size_t step = (size_t)nsl * block;
  int sp = sl;
  for (; sp + 7 * nsl < nact; sp += 8 * nsl) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)sp * block + u * step];
#pragma unroll
    for (int u = 0; u < 8; ++u) ch[u] += (double)v[u];
  }
  for (int u = 0; sp < nact; sp += nsl, ++u) ch[u] += (double)src[(size_t)sp * block];
  const double s0 = ch[0] + ch[4], s1 = ch[1] + ch[5], s2 = ch[2] + ch[6], s3 = ch[3] + ch[7];
  const double tot = (s0 + s1) + (s2 + s3);
  if (nsl > 1) {
    mid[(((size_t)sl * 16 + pos) * cop + co) * cip + ci] = tot;
  } else {
    m[pos][cl] = tot * wgw_pos_scale(pos);
    __syncthreads();
    wgw_taps(m, tid, cl, co, ci, Cin, dw);
  }
  if (db && blockIdx.x % (citiles * nsl) == 0 && tid < 64) wgw_bias(dbp, db, co, cop, nact, tid);
}

__global__ __launch_bounds__(1024) void wgrad_wino_finish_kernel(const double* __restrict__ mid, float* __restrict__ dw, int Cin, int cop,
                                                                 int cip, int nsl) {
  __shared__ double m[16][64];
  const int tid = threadIdx.x, cl = tid & 63, pos = tid >> 6;
  const int citiles = cip / 64;
  const int co = blockIdx.x / citiles, ci = (blockIdx.x % citiles) * 64 + cl;
  const size_t block = (size_t)16 * cop * cip;
  const double* src = mid + ((size_t)pos * cop + co) * cip + ci;
  double t = src[0];
  for (int k = 1; k < nsl; ++k) t += src[(size_t)k * block];
  m[pos][cl] = t * wgw_pos_scale(pos);
  __syncthreads();
  wgw_taps(m, tid, cl, co, ci, Cin, dw);
}

static int g_wgw = -1;       // -1: default (env MCEDM_WGRAD_WINO, else on); 0 / 1: forced by mcedm_op_set_wgrad_wino
void set_wgrad_wino(int enable) { g_wgw = enable; }
static int wgw_env() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_WGRAD_WINO"); env = e ? atoi(e) : 1; }
  return variant_choice(KV_WGRAD_WINO, g_wgw, env);
}

static int wgw64_env() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_WGRAD_WINO64"); env = e ? atoi(e) : 1; }
  return env;
}

// splits of the K range: one round of one workgroup per CU over the siblings of a split (4 * cib * cob in the 128-channel form,
// cib * cob in the 64-channel form), a multiple of 8 (XCD mapping).  A function of the channel counts only, so that the scratch
// can be sized at plan time.
static int wgw_nsplit(int nsib) {
  int ns = 256 / nsib;
  ns = ns / 8 * 8;
  return ns < 8 ? 8 : ns;
}
// 0: not served, 1: the 128-channel form (one xi per workgroup), 2: the 64-channel form (all 16 positions per workgroup)
static int wgw_form_of(int Cout, int Cin, int taps, int W) {
  if (taps != 9 || Cout <= 0 || Cin <= 0) return 0;
  if (Cout % GW_CB == 0 && Cin % GW_CB == 0 && W % 32 == 0) return 1;
  if (Cout % G6_CB == 0 && Cin % G6_CB == 0 && W % 16 == 0 && wgw64_env()) return 2;
  return 0;
}
// slices of the split range in the reduction: at least 256 workgroups of (co, 64 ci, slice); 1: no second kernel
static int wgw_reduce_slices(int Cout, int Cin) {
  const int blocks = Cout * (Cin / 64);
  if (blocks >= 256 || blocks < 1) return 1;
  const int n = (256 + blocks - 1) / blocks;
  return n > 8 ? 8 : n;
}
static int wgw_form_nsplit(int form, int Cout, int Cin) {
  return form == 1 ? wgw_nsplit(4 * (Cout / GW_CB) * (Cin / GW_CB)) : wgw_nsplit((Cout / G6_CB) * (Cin / G6_CB));
}
static size_t wgw_part_floats(int Cout, int Cin) {        // partial blocks + bias partials of the form with the larger split count
  const size_t per = (size_t)16 * Cout * Cin + Cout;
  size_t need = 0;
  if (Cout % GW_CB == 0 && Cin % GW_CB == 0) need = (size_t)wgw_form_nsplit(1, Cout, Cin) * per;
  if (Cout % G6_CB == 0 && Cin % G6_CB == 0 && wgw64_env()) {
    const size_t n = (size_t)wgw_form_nsplit(2, Cout, Cin) * per;
    if (n > need) need = n;
  }
  return need;
}
size_t wgrad_wino_scratch_floats(int Cout, int Cin, int taps) {
  if (taps != 9 || Cout <= 0 || Cin <= 0) return 0;
  const size_t need = wgw_part_floats(Cout, Cin);
  if (need == 0) return 0;
  const int nsl = wgw_reduce_slices(Cout, Cin);
  return need + (nsl > 1 ? 2 * (size_t)nsl * 16 * Cout * Cin : 0);       // + the reduction's fp64 slice sums
}

static int wgw_form(const WgradArgs& a, int taps, int qkv_heads) {
  const int Cin = a.Ca + a.Cb;
  if (!wgw_env() || qkv_heads != 0) return 0;
  const int form = wgw_form_of(a.Cout, Cin, taps, a.W);
  if (!form || a.H % 2 != 0 || a.H < 2 || a.B < 1) return 0;
  if ((reinterpret_cast<size_t>(a.dy) & 15) != 0) return 0;
  const unsigned long long HW = (unsigned long long)a.H * a.W;
  if (!(4ull * a.B * Cin * HW + 16 < (1ull << 32) && 4ull * a.B * a.Cout * HW < (1ull << 32))) return 0;      // 32-bit buffer offsets
  // the 64-channel form pays a fixed 16 x Cout x Cin floats per split (store + reduction): with fewer than 4 stages of 8 tiles
  // per split the direct kernel is as fast or faster (64 -> 64 at 32^2, B = 32: 4 stages, 47 against 46 us; tools/wgrad_wino_ab.py)
  // (a switch FORCED on -- mcedm_op_set_wgrad_wino(1) or the plan's variant -- serves every legal shape: the parity tests' small cases)
  if (form == 2 && variant_choice(KV_WGRAD_WINO, g_wgw, -1) != 1) {
    static int min_per = -1;
    if (min_per < 0) { const char* e = getenv("MCEDM_WGRAD_WINO64_MIN"); min_per = e ? atoi(e) : 4; }
    const long long total = (long long)a.B * (a.H / 2) * (a.W / 16);
    if (total < (long long)min_per * wgw_form_nsplit(2, a.Cout, Cin)) return 0;
  }
  return form;
}
bool wgrad_wino_applicable(const WgradArgs& a, int taps, int qkv_heads) { return wgw_form(a, taps, qkv_heads) != 0; }

int launch_wgrad_wino(const WgradArgs& a, const float* x, float* dw, float* db, hipStream_t s) {
  const int form = wgw_form(a, 9, 0);
  MCEDM_REQUIRE(form != 0, "wgrad_wino: shape not served by the Winograd weight-gradient kernels");
  MCEDM_REQUIRE(a.dy && x && a.dwp && dw, "wgrad_wino: null pointer");
  const int Cin = a.Ca + a.Cb;
  const unsigned long long HW = (unsigned long long)a.H * a.W;
  const int cb = form == 1 ? GW_CB : G6_CB, segw = form == 1 ? 32 : 16;
  WgWinoArgs p{};
  p.dy = a.dy; p.x = x; p.Co = a.Cout; p.Ci = Cin; p.B = a.B; p.H = a.H; p.W = a.W;
  p.cob = a.Cout / cb; p.cib = Cin / cb; p.cop = a.Cout; p.cip = Cin;
  p.nseg = a.W / segw; p.th = a.H / 2;
  p.total = a.B * p.th * p.nseg;
  const int nsplit_max = wgw_form_nsplit(form, a.Cout, Cin);
  int nsplit = nsplit_max;
  if (nsplit > p.total) nsplit = p.total;
  p.per = ceil_div(p.total, nsplit);
  p.nact = ceil_div(p.total, p.per);
  p.part = a.dwp;
  p.dbp = db ? a.dwp + (size_t)nsplit_max * 16 * a.Cout * Cin : nullptr;
  const int nsib = (form == 1 ? 4 : 1) * p.cib * p.cob;
  const int grid = ceil_div(p.nact, 8) * 8 * nsib;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  MCEDM_HIP_TRY(hipGetDevice(&dev));
  MCEDM_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
  if (!attr_set[dev].load(std::memory_order_acquire)) {
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad_wino64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[dev].store(true, std::memory_order_release);
  }
  {
    const double flops = 2.0 * a.B * (double)HW * a.Cout * Cin * 9;       // algorithmic (the direct form's); 4 / 9 are issued
    ProfScope ps(form == 1 ? "wgrad_wino_kernel" : "wgrad_wino64_kernel", flops, 4.0 * a.B * (double)HW * (a.Cout + Cin), s);
    if (form == 1) hipLaunchKernelGGL(wgrad_wino_kernel, dim3(grid), dim3(512), 2 * GW_STAGE * sizeof(float), s, p);
    else hipLaunchKernelGGL(wgrad_wino64_kernel, dim3(grid), dim3(512), 2 * G6_STAGE * sizeof(float), s, p);
    MCEDM_LAUNCH_CHECK("wgrad_wino_kernel");
  }
  {
    const double elems = 16.0 * a.Cout * Cin;
    int nsl = wgw_reduce_slices(a.Cout, Cin);
    if (nsl > p.nact) nsl = 1;
    // the fp64 slice sums live behind the partial blocks and the bias partials of the LARGEST split count these channels can get
    const size_t mid_off = wgw_part_floats(a.Cout, Cin);
    double* mid = reinterpret_cast<double*>(a.dwp + mid_off);
    MCEDM_REQUIRE(nsl == 1 || (reinterpret_cast<size_t>(mid) & 7) == 0, "wgrad_wino: scratch not 8-byte aligned");
    {
      ProfScope ps("wgrad_wino_reduce_kernel", p.nact * elems, 4.0 * (p.nact * elems + 9.0 * a.Cout * Cin), s);
      hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3(a.Cout * (Cin / 64) * nsl), dim3(1024), 0, s, p.part, p.dbp, dw, db, mid, a.Cout, Cin,
                         p.cop, p.cip, p.nact, nsl);
      MCEDM_LAUNCH_CHECK("wgrad_wino_reduce_kernel");
    }
    if (nsl > 1) {
      ProfScope ps("wgrad_wino_finish_kernel", nsl * elems, 8.0 * nsl * elems + 36.0 * a.Cout * Cin, s);
      hipLaunchKernelGGL(wgrad_wino_finish_kernel, dim3(a.Cout * (Cin / 64)), dim3(1024), 0, s, mid, dw, Cin, p.cop, p.cip, nsl);
      MCEDM_LAUNCH_CHECK("wgrad_wino_finish_kernel");
    }
  }
  return MCEDM_OK;
}

// ---- the 1x1 GEMM form
static int wgg_env() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_WGRAD_GEMM1"); env = e ? atoi(e) : 1; }
  return env;
}
static bool wgg_shape_ok(int Cout, int Cin, int taps) { return taps == 1 && Cout > 0 && Cin > 0 && Cout % GW_CB == 0 && Cin % GW_CB == 0; }
// slices ([co][ci] blocks + bias rows) of the LARGEST split count these channel counts can get
size_t wgrad_gemm1_scratch_floats(int Cout, int Cin, int taps) {
  if (!wgg_shape_ok(Cout, Cin, taps) || !wgg_env()) return 0;
  return (size_t)4 * wgw_nsplit((Cout / GW_CB) * (Cin / GW_CB)) * ((size_t)Cout * Cin + Cout);
}
// the operand must be readable in place: one tensor, or cat(xa, xb) with Ca a multiple of 128
bool wgrad_gemm1_applicable(const WgradArgs& a, int taps, const float* x, bool in_place_concat) {
  const int Cin = a.Ca + a.Cb;
  if (!wgg_env() || !wgw_env() || !wgg_shape_ok(a.Cout, Cin, taps)) return false;
  const unsigned long long HW = (unsigned long long)a.H * a.W;
  if (HW % 64 != 0 || a.B < 1) return false;
  if (in_place_concat && (a.Ca % GW_CB != 0 || !a.xa || !a.xb)) return false;
  if (!in_place_concat && !x) return false;
  const size_t bits = reinterpret_cast<size_t>(a.dy) | (in_place_concat ? (reinterpret_cast<size_t>(a.xa) | reinterpret_cast<size_t>(a.xb)) : reinterpret_cast<size_t>(x));
  if (bits & 15) return false;
  if (4ull * a.B * Cin * HW >= (1ull << 32) || 4ull * a.B * a.Cout * HW >= (1ull << 32)) return false;
  // enough stages per split to pay for the slices (as for the 64-channel Winograd form)
  const long long total = (long long)a.B * (long long)(HW / 64);
  return total >= 4ll * wgw_nsplit((a.Cout / GW_CB) * (Cin / GW_CB));
}

// x: the single-tensor operand, or nullptr for cat(a.xa, a.xb) in place.  *nslices: slices written (for wgrad_reduce_kernel)
int launch_wgrad_gemm1(const WgradArgs& a, const float* x, float* dbp, int* nslices, hipStream_t s) {
  const int Cin = a.Ca + a.Cb;
  const unsigned long long HW = (unsigned long long)a.H * a.W;
  WgWinoArgs p{};
  p.dy = a.dy; p.x = x ? x : a.xa; p.Co = a.Cout; p.Ci = Cin; p.B = a.B; p.H = a.H; p.W = a.W;
  p.cob = a.Cout / GW_CB; p.cib = Cin / GW_CB; p.cop = a.Cout; p.cip = Cin;
  p.nseg = (int)(HW / 64); p.th = 0;
  p.total = a.B * p.nseg;
  int nsplit = wgw_nsplit(p.cib * p.cob);
  if (nsplit > p.total) nsplit = p.total;
  p.per = ceil_div(p.total, nsplit);
  p.nact = ceil_div(p.total, p.per);
  p.part = a.dwp;
  p.dbp = dbp;
  const int nsib = p.cib * p.cob;
  const int grid = ceil_div(p.nact, 8) * 8 * nsib;
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  MCEDM_HIP_TRY(hipGetDevice(&dev));
  MCEDM_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
  if (!attr_set[dev].load(std::memory_order_acquire)) {
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad_gemm1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[dev].store(true, std::memory_order_release);
  }
  ProfScope ps("wgrad_gemm1_kernel", 2.0 * a.B * (double)HW * a.Cout * Cin, 4.0 * a.B * (double)HW * (a.Cout + Cin), s);
  // a single-tensor operand is "all in xa": Ca = Cin
  hipLaunchKernelGGL(wgrad_gemm1_kernel, dim3(grid), dim3(512), 2 * GW_STAGE * sizeof(float), s, p, x ? nullptr : a.xb, x ? Cin : a.Ca);
  MCEDM_LAUNCH_CHECK("wgrad_gemm1_kernel");
  *nslices = 4 * p.nact;
  return MCEDM_OK;
}

}  // namespace mcedm
