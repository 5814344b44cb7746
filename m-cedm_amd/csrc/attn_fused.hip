// attn_fused.hip -- the whole attention part of a UNetBlock at 8 x 8 resolution in ONE launch (inference):
//     z = proj(attention(qkv(group_norm(y)))) + y           (models/adm_blocks.py:103-109, 174-180)
// for T = 64 tokens, C = 64 channels, one head of 64.  One workgroup (4 waves) owns one sample: the sample is 16 KB, so
// GroupNorm (two passes over registers), the three GEMMs and the softmax never leave the CU, and what were three
// launches (1x1 qkv conv, attention kernel, 1x1 proj conv: 11 + 14 + 11 us on 64-192 workgroups, latency chains all
// three) is one.  All products run on the fp32 MFMA (v_mfma_f32_32x32x2_f32); operands come from LDS, the weight
// fragments (A operands of qkv / proj) straight from the packed 1x1 tables, all requested before anything waits.
//   layout of a 32 x 32 MFMA result: lane l, register r  <->  row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
// The fused GroupNorm statistics of z (for the next block's conv0) are emitted exactly as a conv epilogue would.
#include <atomic>
#include <cstdlib>

#include "common.hpp"
#include "conv_tile.hpp"
#include "prof.hpp"

namespace mcedm {

struct AttnBlockArgs {
  const float* y;        // [B][64][64]  block output before attention (also the residual)
  float* z;              // [B][64][64]
  const float* gamma; const float* beta; float eps; int groups;   // norm2
  const float* wq; const float* bq;     // packed 1x1 qkv table [64][192] (rows = input channel), packed bias [192]: q | k | v
  const float* wp; const float* bp;     // packed 1x1 proj table [64][64], bias [64]
  float* gsum;           // [B][1][16][2] (sum, M2) per 4-channel block of z, or null
  int B;
};

typedef ConvCfg<64, 8, 8, 2, 2, 1, 16> AttnTile;      // the output tile as the conv epilogue sees it

__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(256) void attn_block64_kernel(AttnBlockArgs a) {
  constexpr int T = 64, C = 64, VP = 65;           // VP: pitch of the transposed V tile (conflict-free both ways)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* X = lds;                     // [C][T] raw y (residual)
  float* XN = X + C * T;              // [C][T] group-normalised y; later the attention output O[c][q]
  float* Qs = XN + C * T;             // [C][T] q / 8
  float* Ks = Qs + C * T;             // [C][T] k
  float* Vt = Ks + C * T;             // [T][VP] v transposed
  float (*smax)[T] = reinterpret_cast<float (*)[T]>(Vt + T * VP);
  float (*ssum)[T] = smax + 2;
  float* red = reinterpret_cast<float*>(ssum + 2);     // [2][16][3] statistics records of the epilogue

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int n = blockIdx.x;
  const float* y = a.y + (size_t)n * C * T;

  // ---- weight fragments of the qkv GEMM: wave w owns token block nb = w & 1 and output row blocks 3 (w >> 1) + {0, 1, 2}
  const int nb = wave & 1, mg = wave >> 1;
  float wa[3][32];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int s = 0; s < 32; ++s) wa[i][s] = a.wq[(size_t)(2 * s + lh) * 192 + 32 * (3 * mg + i) + l31];

  // ---- y -> registers (16 values per thread: channel tid / 4, pixels 16 (tid & 3) ...), GroupNorm over 4 channels x 64
  const int c = tid >> 2, q4 = tid & 3;
  f32x4 xr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) xr[j] = *reinterpret_cast<const f32x4*>(y + c * T + 16 * q4 + 4 * j);
  float s1 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s1 += (xr[j].x + xr[j].y) + (xr[j].z + xr[j].w);
  const int cpg = C / a.groups;                      // channels per group: 4 (the launcher checks), i.e. 16 threads
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) s1 += __shfl_xor(s1, off);
  const float mean = s1 / (float)(cpg * T);
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 d = xr[j] - mean;
    m2 += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
  }
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) m2 += __shfl_xor(m2, off);
  const float rstd = 1.0f / sqrtf(m2 / (float)(cpg * T) + a.eps);
  const float gsc = rstd * a.gamma[c], gof = a.beta[c];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    *reinterpret_cast<f32x4*>(X + c * T + 16 * q4 + 4 * j) = xr[j];
    *reinterpret_cast<f32x4*>(XN + c * T + 16 * q4 + 4 * j) = (xr[j] - mean) * gsc + gof;
  }
  __syncthreads();

  // ---- qkv = Wqkv . xn + b   (192 x 64 = 6 x 2 blocks; three per wave, sharing the B fragment)
  {
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = a.bq[32 * (3 * mg + i) + mfma_row(r, lane)];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float b = XN[(2 * s + lh) * T + 32 * nb + l31];
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i][s], b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int mb = 3 * mg + i;                     // 0,1: q   2,3: k   4,5: v   (packed row order q | k | v)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * (mb & 1) + mfma_row(r, lane), tok = 32 * nb + l31;
        if (mb < 2) Qs[row * T + tok] = acc[i][r] * 0.125f;           // 1 / sqrt(64), exact
        else if (mb < 4) Ks[row * T + tok] = acc[i][r];
        else Vt[tok * VP + row] = acc[i][r];
      }
    }
  }
  // proj weight fragments: wave w owns output rows 32 (w >> 1) ..., tokens 32 (w & 1) ...; in flight during the softmax
  float wpj[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) wpj[s] = a.wp[(size_t)(2 * s + lh) * C + 32 * mg + l31];
  __syncthreads();

  // ---- scores, transposed: S[key][query] = sum_c k[c][key] q[c][query]; wave w: key block kb = w >> 1, query block qb = w & 1
  const int kb = wave >> 1, qb = wave & 1;
  f32x16 sc;
#pragma unroll
  for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < 32; ++s)
    sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(2 * s + lh) * T + 32 * kb + l31], Qs[(2 * s + lh) * T + 32 * qb + l31], sc, 0, 0, 0);
  // softmax over the 64 keys of query 32 qb + l31: this lane holds 16 keys, its partner lane (^32) the other 16 of the
  // key block, the wave (kb ^ 1, qb) the other key block
  float mx = sc[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  if (lh == 0) smax[kb][32 * qb + l31] = mx;
  __syncthreads();
  const float mfin = fmaxf(smax[0][32 * qb + l31], smax[1][32 * qb + l31]);
  float pr[16], rs = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { pr[r] = expf(sc[r] - mfin); rs += pr[r]; }
  rs += __shfl_xor(rs, 32);
  if (lh == 0) ssum[kb][32 * qb + l31] = rs;

  // ---- partial O[c][query] over this wave's key block: register r contracts keys 32 kb + (r & 3) + 8 (r >> 2) (+ 4 in the
  // upper half-wave), exactly where the probabilities already sit as B operands
  f32x16 o[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * kb + mfma_row(r, lane);
      o[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vt[key * VP + 32 * cb + l31], pr[r], o[cb], 0, 0, 0);
    }
  }
  // the two key blocks are summed through LDS (XN is free): kb = 1 stores, kb = 0 adds, normalises, stores O
  if (kb == 1) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) XN[(32 * cb + mfma_row(r, lane)) * T + 32 * qb + l31] = o[cb][r];
  }
  __syncthreads();
  if (kb == 0) {
    const float inv = 1.0f / (ssum[0][32 * qb + l31] + ssum[1][32 * qb + l31]);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* slot = XN + (32 * cb + mfma_row(r, lane)) * T + 32 * qb + l31;
        *slot = (o[cb][r] + *slot) * inv;
      }
  }
  __syncthreads();

  // ---- z = Wproj . O + b + y; store + statistics as a conv epilogue on the <64, 8, 8> tile (wm = cout block, wn = token block)
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = 32 * wm + mfma_row(r, lane);
    acc[0][0][r] = a.bp[co] + X[co * T + 32 * wn + l31];
  }
#pragma unroll
  for (int s = 0; s < 32; ++s)
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wpj[s], XN[(2 * s + lh) * T + 32 * wn + l31], acc[0][0], 0, 0, 0);
  ConvArgs p{};
  p.out = a.z; p.Cout = C; p.H = 8; p.W = 8; p.gsum = a.gsum;
  if (a.gsum) conv_epilogue<AttnTile, true, 4>(p, acc, n, 0, 0, 0, wm, wn, lane, red);
  else conv_epilogue<AttnTile, true, 0>(p, acc, n, 0, 0, 0, wm, wn, lane, red);
  if (a.gsum) {
    __syncthreads();
    constexpr int NG = C / 4;
    if (tid < NG) {
      float sum, m2o;
      conv_stats_combine<AttnTile::WN>(red + tid * 3, NG * 3, sum, m2o);
      float* row = a.gsum + ((size_t)n * NG + tid) * 2;
      row[0] = sum; row[1] = m2o;
    }
  }
}

static int g_attn_fused = -1;      // -1: env MCEDM_ATTN_FUSED (default on); test hook
void set_attn_fused(int enable) { g_attn_fused = enable; }
bool attn_block_fused_applicable(int C, int heads, int H, int W, int groups) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_ATTN_FUSED"); env = e ? atoi(e) : 1; }
  return variant_choice(KV_ATTN_FUSED, g_attn_fused, env) != 0 && C == 64 && heads == 1 && H == 8 && W == 8 && groups == 16;
}

// y, z: [B][64][8][8]; wq / bq, wp / bp: the packed 1x1 tables of the block's qkv and proj convs
int launch_attn_block64(const float* y, float* z, const float* gamma, const float* beta, float eps, int groups, const float* wq,
                        const float* bq, const float* wp, const float* bp, float* gsum, SumTiles* gsum_tiles, int B,
                        hipStream_t stream) {
  MCEDM_REQUIRE(y && z && wq && bq && wp && bp && gamma && beta && B > 0 && groups == 16, "attn_block64: bad arguments");
  AttnBlockArgs a{y, z, gamma, beta, eps, groups, wq, bq, wp, bp, gsum, B};
  // algorithmic cost: qkv 2*192*64*64, scores + PV 2 * 2*64*64*64, proj 2*64*64*64 flops per sample; y in, z out, weights
  ProfScope ps("attn_block64_kernel", (double)B * 2.0 * 64 * 64 * (192 + 64 + 64 + 64),
               4.0 * ((double)B * 2 * 64 * 64 + 64.0 * (192 + 64)), stream);
  constexpr int lds_bytes = (4 * 64 * 64 + 64 * 65 + 4 * 64 + 2 * 16 * 3) * (int)sizeof(float);
  static std::atomic<bool> attr_set[64];      // zero-initialised; a repeated set is benign, a data race is not
  int dev = 0;
  MCEDM_HIP_TRY(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && !attr_set[dev].load(std::memory_order_acquire)) {
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)attn_block64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(attn_block64_kernel, dim3(B), dim3(256), lds_bytes, stream, a);
  MCEDM_LAUNCH_CHECK("attn_block64_kernel");
  if (gsum_tiles) *gsum_tiles = SumTiles{1, 1, 8, 8};
  return MCEDM_OK;
}

}  // namespace mcedm
