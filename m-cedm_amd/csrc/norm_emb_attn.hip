// norm_emb_attn.hip -- K1 (GroupNorm statistics -> per-channel transform table), K6 (sigma
// embedding MLP + every FiLM affine row) and K5 (fused softmax attention on fp32 MFMA).
#include <cstdlib>
#include <atomic>
#include "common.hpp"
#include "prof.hpp"

namespace mcedm {

// =========================================================================================
// K1: GroupNorm statistics.  One workgroup per (sample, group); the group's cpg*HW floats are
// contiguous in NCHW.  Sums are taken relative to the group's first element (shifted-data
// variance), per-thread in fp32 and across threads in fp64.  HBM-bound: reads the tensor once.
// Emits, per channel, the transform the consuming conv applies while staging its input:
//   norm(x)*(film_scale+1) + film_shift  ==  (x - mean) * [g*rstd*(1+s)] + [b*(1+s) + t]
// (adm_blocks.py:94-97 and :163-166).
// =========================================================================================
__global__ __launch_bounds__(256) void gn_coef_kernel(GnArgs a) {
  const int C = a.Ca + a.Cb;
  const int cpg = C / a.groups;
  const int n = blockIdx.x / a.groups;
  const int g = blockIdx.x % a.groups;
  const int c0 = g * cpg;
  // a group of the virtual concat cat(xa, xb) may straddle the boundary: the source is chosen per channel
  auto plane = [&](int c) {
    return (c < a.Ca) ? a.xa + ((size_t)n * a.Ca + c) * a.HW : a.xb + ((size_t)n * a.Cb + (c - a.Ca)) * a.HW;
  };
  const int N = cpg * a.HW;
  const float shift = plane(c0)[0];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int tid = threadIdx.x;
  for (int cl = 0; cl < cpg; ++cl) {
    const float* src = plane(c0 + cl);
    if ((a.HW & 3) == 0) {
      const float4* s4 = reinterpret_cast<const float4*>(src);
      const int N4 = a.HW >> 2;
      for (int i = tid; i < N4; i += 256) {
        const float4 v = s4[i];
        const float d0 = v.x - shift, d1 = v.y - shift, d2 = v.z - shift, d3 = v.w - shift;
        s1[0] += d0; s1[1] += d1; s1[2] += d2; s1[3] += d3;
        s2[0] += d0 * d0; s2[1] += d1 * d1; s2[2] += d2 * d2; s2[3] += d3 * d3;
      }
    } else {
      for (int i = tid; i < a.HW; i += 256) {
        const float d = src[i] - shift;
        s1[0] += d; s2[0] += d * d;
      }
    }
  }
  double t1 = ((double)s1[0] + (double)s1[1]) + ((double)s1[2] + (double)s1[3]);
  double t2 = ((double)s2[0] + (double)s2[1]) + ((double)s2[2] + (double)s2[3]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    t1 += __shfl_xor(t1, off);
    t2 += __shfl_xor(t2, off);
  }
  __shared__ double red[2][4];
  __shared__ float stat[2];
  if ((tid & 63) == 0) { red[0][tid >> 6] = t1; red[1][tid >> 6] = t2; }
  __syncthreads();
  if (tid == 0) {
    const double S1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double S2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double m = S1 / N;
    double var = S2 / N - m * m;
    if (var < 0) var = 0;
    const float mean = (float)((double)shift + m);
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    stat[0] = mean; stat[1] = rstd;
    if (a.stats) { a.stats[((size_t)n * a.groups + g) * 2] = mean; a.stats[((size_t)n * a.groups + g) * 2 + 1] = rstd; }
  }
  __syncthreads();
  if (tid < cpg) {
    const int c = c0 + tid;
    const float mean = stat[0], rstd = stat[1];
    float sc = 1.f, sh = 0.f;
    if (a.film) {
      const float* f = a.film + (size_t)(a.film_batch ? n : 0) * a.film_stride;
      sc = f[c] + 1.f;
      sh = f[C + c];
    }
    Coef o;
    o.mean = mean;
    o.scale = a.gamma[c] * rstd * sc;
    o.offset = a.beta[c] * sc + sh;
    o.pad = 0.f;
    a.coef[(size_t)n * C + c] = o;
  }
}

int launch_gn_coef(const GnArgs& a, hipStream_t stream) {
  const int C = a.Ca + a.Cb;
  MCEDM_REQUIRE(a.groups > 0 && C % a.groups == 0, "group_norm: C=%d not divisible by groups=%d", C, a.groups);
  const int cpg = C / a.groups;
  MCEDM_REQUIRE(cpg <= 256, "group_norm: too many channels per group (%d)", cpg);
  MCEDM_REQUIRE(a.xa != nullptr && (a.Cb == 0 || a.xb != nullptr), "group_norm: null input");
  ProfScope ps("gn_coef_kernel", 3.0 * a.B * (double)C * a.HW, 4.0 * a.B * (double)C * a.HW, stream);
  hipLaunchKernelGGL(gn_coef_kernel, dim3(a.B * a.groups), dim3(256), 0, stream, a);
  MCEDM_LAUNCH_CHECK("gn_coef_kernel");
  return MCEDM_OK;
}

// K1': the same table from the (sum, M2 about the tile mean) records per 4 channels that the PRODUCING conv's epilogue
// wrote per output tile (ConvArgs::gsum): no pass over the tensor at all.  One wave per (sample, group): lanes
// stride over the producer's tiles (fp64 partial sums), a fixed-order butterfly combines them (bitwise reproducible,
// independent of the batch size), lanes 0..cpg-1 then write the group's channel rows.
__global__ __launch_bounds__(256) void gn_coef_from_sums_kernel(GnArgs a) {
  const int C = a.Ca + a.Cb;
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);       // (sample, group)
  if (pair >= a.B * a.groups) return;
  const int n = pair / a.groups, g = pair - n * a.groups;
  const int cpg = C / a.groups;
  const int c0 = g * cpg;
  // the group is a run of statistic records (4 or 2 channels each, SumTiles::rc); each lives in xa's or xb's table (a
  // group may straddle the concat boundary, see gn_sums_usable).
  // Chan merge of the per-tile records (sum_t, M2_t about the tile mean) in fp64, as in conv_tile.hpp stage_coef_rows
  double s1 = 0, sq = 0, mw = 0;
  const int Himg = a.HW / a.W;
  for (int cb = c0; cb < c0 + cpg;) {
    const bool in_a = cb < a.Ca;
    const float* sums = in_a ? a.suma : a.sumb;
    const SumTiles& tg = in_a ? a.ta : a.tb;
    const int nrec = ((in_a ? a.Ca : a.Cb) + tg.rc - 1) / tg.rc;
    const int qi = (in_a ? cb : cb - a.Ca) / tg.rc;
    for (int t = lane; t < tg.tiles; t += 64) {
      const float* row = sums + (((size_t)n * tg.tiles + t) * nrec + qi) * 2;
      const double st = (double)row[0];
      s1 += st; sq += st * st / (double)sum_tile_count(tg, t, Himg, a.W); mw += (double)row[1];
    }
    cb += tg.rc;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); sq += __shfl_xor(sq, off); mw += __shfl_xor(mw, off); }
  const double N = (double)cpg * a.HW;
  const double m = s1 / N;
  double var = (mw + sq - s1 * s1 / N) / N;
  if (var < 0) var = 0;
  const float mean = (float)m;
  const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  if (a.stats && lane == 0) { a.stats[((size_t)n * a.groups + g) * 2] = mean; a.stats[((size_t)n * a.groups + g) * 2 + 1] = rstd; }
  for (int k = lane; k < cpg; k += 64) {
    const int c = c0 + k;
    float sc = 1.f, sh = 0.f;
    if (a.film) {
      const float* f = a.film + (size_t)(a.film_batch ? n : 0) * a.film_stride;
      sc = f[c] + 1.f;
      sh = f[C + c];
    }
    Coef o;
    o.mean = mean;
    o.scale = a.gamma[c] * rstd * sc;
    o.offset = a.beta[c] * sc + sh;
    o.pad = 0.f;
    a.coef[(size_t)n * C + c] = o;
  }
}

bool gn_sums_usable(const GnArgs& a) {
  const int C = a.Ca + a.Cb;
  if (a.groups <= 0 || C % a.groups != 0) return false;
  const int cpg = C / a.groups;
  if (a.suma == nullptr || (a.Cb != 0 && a.sumb == nullptr) || a.W <= 0 || a.HW % a.W != 0 || a.ta.tiles <= 0 ||
      (a.Cb != 0 && a.tb.tiles <= 0))
    return false;
  // every group must be a whole number of records of each source it touches, and records must not straddle the sources
  const int ra = a.ta.rc, rb = a.Cb ? a.tb.rc : ra;
  if ((ra != 2 && ra != 4) || (rb != 2 && rb != 4)) return false;
  return cpg % ra == 0 && cpg % rb == 0 && a.Ca % ra == 0 && a.Ca % rb == 0;
}

int launch_gn_coef_from_sums(const GnArgs& a, hipStream_t stream) {
  const int C = a.Ca + a.Cb;
  MCEDM_REQUIRE(a.groups > 0 && C % a.groups == 0, "group_norm: C=%d not divisible by groups=%d", C, a.groups);
  if (!gn_sums_usable(a))
    return launch_gn_coef(a, stream);            // no fused statistics for this shape: one pass over the tensor
  hipLaunchKernelGGL(gn_coef_from_sums_kernel, dim3(ceil_div(a.B * a.groups, 4)), dim3(256), 0, stream, a);
  MCEDM_LAUNCH_CHECK("gn_coef_from_sums_kernel");
  return MCEDM_OK;
}

// =========================================================================================
// K6: PositionalEmbedding -> map_layer0 -> SiLU -> map_layer1 -> SiLU (adm_blocks.py:192-199,
// 367-379) and all per-block `affine` rows (adm_blocks.py:163) in one launch.
// One workgroup per noise label (n = 1 while sampling, n = B while training).
// =========================================================================================
__device__ __forceinline__ float silu_e(float v) { return v / (1.0f + expf(-v)); }

// dot(x[0..n), w[0..n)) with the 64 lanes of a wave striding over k (coalesced 256-byte row segments) and a
// fixed-order butterfly: every lane returns the sum.  x in LDS, w in global memory.
__device__ __forceinline__ float wave_dot(const float* x, const float* __restrict__ w, int n, int lane) {
  float s = 0.f;
  for (int k = lane; k < n; k += 64) s = fmaf(x[k], w[k], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  return s;
}

// R rows at once: all R * n / 64 weight loads of a lane are issued before anything waits on them (the kernel is a chain
// of memory latencies: one row per iteration cost ~0.5 us each, 33 us per launch at ch = 256).  Per row the arithmetic
// is wave_dot's: the same fmaf chain per lane and the same butterfly, hence the same bits.
template <int R>
__device__ __forceinline__ void wave_dot_rows(const float* x, const float* __restrict__ w, int stride, int rows_left, int n,
                                              int lane, float (&out)[R]) {
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) s[r] = 0.f;
  for (int k = lane; k < n; k += 64) {
    const float xv = x[k];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = fmaf(xv, w[(size_t)(r < rows_left ? r : 0) * stride + k], s[r]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] += __shfl_xor(s[r], off);
#pragma unroll
  for (int r = 0; r < R; ++r) out[r] = s[r];
}

// grid (n, nsplit), 16 waves: every workgroup evaluates the small mapping MLP (2 x ch^2 MACs), then its slice of the
// `rows` FiLM rows of all blocks; four output rows per wave and step.  (One thread per row read the weight matrices
// with a stride of ch floats between lanes and ran on a single workgroup while sampling: 137 us at ch = 512.)
constexpr int EMB_NT = 1024, EMB_R = 4;
__global__ __launch_bounds__(EMB_NT) void embedding_kernel(EmbArgs a) {
  extern __shared__ float sm[];
  float* e0 = sm;            // [ch]
  float* e1 = sm + a.ch;     // [ch]
  constexpr int NW = EMB_NT / 64;
  const int n = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ch = a.ch, half = a.ch / 2;
  const float x = a.labels[n];
  for (int k = tid; k < ch; k += EMB_NT) {
    const float arg = x * a.freqs[k < half ? k : k - half];
    e0[k] = (k < half) ? cosf(arg) : sinf(arg);
  }
  __syncthreads();
  float s[EMB_R];
  for (int j = wave * EMB_R; j < ch; j += NW * EMB_R) {
    wave_dot_rows<EMB_R>(e0, a.w0 + (size_t)j * ch, ch, ch - j, ch, lane, s);
    if (lane == 0)
      for (int r = 0; r < EMB_R && j + r < ch; ++r) e1[j + r] = silu_e(s[r] + a.b0[j + r]);
  }
  __syncthreads();
  for (int j = wave * EMB_R; j < ch; j += NW * EMB_R) {
    wave_dot_rows<EMB_R>(e1, a.w1 + (size_t)j * ch, ch, ch - j, ch, lane, s);
    if (lane == 0)
      for (int r = 0; r < EMB_R && j + r < ch; ++r) {
        const float v = silu_e(s[r] + a.b1[j + r]);
        e0[j + r] = v;   // e0 is free after the first barrier pair
        if (a.emb && split == 0) a.emb[(size_t)n * ch + j + r] = v;
      }
  }
  __syncthreads();
  const int per = (a.rows + nsplit - 1) / nsplit;
  const int r1 = min(a.rows, (split + 1) * per);
  for (int r0 = split * per + wave * EMB_R; r0 < r1; r0 += NW * EMB_R) {
    wave_dot_rows<EMB_R>(e0, a.waff + (size_t)r0 * ch, ch, r1 - r0, ch, lane, s);
    if (lane == 0)
      for (int r = 0; r < EMB_R && r0 + r < r1; ++r) a.film[(size_t)n * a.rows + r0 + r] = s[r] + a.baff[r0 + r];
  }
}

int launch_embedding(const EmbArgs& a, hipStream_t stream) {
  MCEDM_REQUIRE(a.n > 0 && a.ch > 0 && a.ch % 2 == 0, "embedding: bad shape n=%d ch=%d", a.n, a.ch);
  // enough workgroups to occupy the chip when n is small (sampling: one noise level per batch)
  int nsplit = a.rows / 64;
  if (nsplit > 512 / a.n) nsplit = 512 / a.n;
  if (nsplit < 1) nsplit = 1;
  hipLaunchKernelGGL(embedding_kernel, dim3(a.n, nsplit), dim3(EMB_NT), 2 * a.ch * sizeof(float), stream, a);
  MCEDM_LAUNCH_CHECK("embedding_kernel");
  return MCEDM_OK;
}

// =========================================================================================
// K5: attention  a[c][q] = sum_k softmax_k(q.k/8)[q][k] * v[c][k]   (adm_blocks.py:103-109,174-178)
// One wave per 32 queries of one (sample, head); keys in tiles of 32; head_dim = 64; fp32 MFMA.
// The score tile is computed TRANSPOSED (rows = keys, column = this lane's query) so the softmax
// row reduction is 15 in-register ops + one cross-half shuffle, and the probabilities are already
// in B-operand position for the P.V product (no LDS, no transposes).  The reference materialises
// the [T x T] weight matrix; this kernel never does.
// qkv layout (written by the packed qkv 1x1 conv): [B][heads][3][64][T].
// =========================================================================================
__global__ __launch_bounds__(64) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                       int T) {
  const int lane = threadIdx.x;
  const int l31 = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * 32;
  const size_t bh = blockIdx.y;
  const float* Q = qkv + (bh * 3 + 0) * 64 * (size_t)T;
  const float* K = qkv + (bh * 3 + 1) * 64 * (size_t)T;
  const float* V = qkv + (bh * 3 + 2) * 64 * (size_t)T;
  const int q = q0 + l31;
  const int qc = q < T ? q : T - 1;

  float qreg[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) qreg[s] = Q[(size_t)(2 * s + h) * T + qc] * 0.125f;   // 1/sqrt(64), exact

  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

  for (int k0 = 0; k0 < T; k0 += 32) {
    const bool full = (k0 + 32 <= T);
    const int kk = k0 + l31;
    const int kc = kk < T ? kk : T - 1;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 32; ++st) {
      const float a = K[(size_t)(2 * st + h) * T + kc];
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qreg[st], s, 0, 0, 0);
    }
    // s[r] = score(query l31, key k0 + (r&3) + 8*(r>>2) + 4*h)
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (!full && key >= T) s[r] = -INFINITY;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    const float alpha = expf(m - m_new);
    float p[16];
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = expf(s[r] - m_new);
      rs += p[r];
    }
    rs += __shfl_xor(rs, 32);
    l = l * alpha + rs;
    m = m_new;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    // O[c][q] += sum_key V[c][key] * P[key][q]; MFMA step r contracts the key pair held by the two lane halves
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* vrow = V + (size_t)(32 * i + l31) * T;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v4[4];
        const int kb = k0 + 8 * g + 4 * h;
        if (full && (T & 3) == 0) {
          const float4 t = *reinterpret_cast<const float4*>(vrow + kb);
          v4[0] = t.x; v4[1] = t.y; v4[2] = t.z; v4[3] = t.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] = vrow[(kb + e) < T ? (kb + e) : T - 1];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v4[e], p[4 * g + e], o[i], 0, 0, 0);
      }
    }
  }
  if (q < T) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        out[(bh * 64 + c) * (size_t)T + q] = o[i][r] * inv;
      }
  }
}

// The same attention with the KEY range split over the KW waves of a workgroup (flash-decoding style).  One wave per 32
// queries means B * heads * T / 32 waves per launch: 1024 at T = 1024, B = 32, one head (the 32 x 32 level of the ch = 64
// networks), i.e. ONE wave per SIMD, and every K / V load latency and every MFMA -> softmax -> MFMA dependency sits exposed
// (measured 47 TFLOP/s).  Here wave w of a workgroup takes the 32-key tiles w, w + KW, ... of the same 32 queries, so a
// SIMD holds KW waves that cover each other's latencies; the partial (max, sum, output) triples meet in LDS and wave 0
// merges them in wave order (fixed order: bitwise reproducible, independent of the batch size).  Scores are kept in the
// log2 domain (q is pre-scaled by log2(e) / 8) so the exponentials are single v_exp_f32 instructions.
template <int KW>
__global__ __launch_bounds__(64 * KW) __attribute__((amdgpu_waves_per_eu(3, 3))) void attention_split_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T) {
  __shared__ float part[KW > 1 ? (KW - 1) * 34 * 64 : 1];        // waves 1 .. KW-1: o[32], m, l per lane
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * 32;
  const size_t bh = blockIdx.y;
  const float* Q = qkv + (bh * 3 + 0) * 64 * (size_t)T;
  const float* K = qkv + (bh * 3 + 1) * 64 * (size_t)T;
  const float* V = qkv + (bh * 3 + 2) * 64 * (size_t)T;
  const int q = q0 + l31;

  // the query tile, scaled, in LDS: the KW waves of the workgroup share it (kept in registers it costs each wave 32 VGPRs,
  // and the kernel must stay under 170 to run three waves per SIMD)
  __shared__ float qs[64 * 32];
  for (int e = threadIdx.x; e < 64 * 32; e += 64 * KW) {
    const int c = e >> 5, qq = q0 + (e & 31);
    qs[e] = Q[(size_t)c * T + (qq < T ? qq : T - 1)] * (0.125f * 1.44269504088896340736f);
  }
  __syncthreads();
  const float* qrow = qs + h * 32 + l31;          // + 64 * st: channel 2 st + h, query l31

  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

  for (int k0 = 32 * wave; k0 < T; k0 += 32 * KW) {
    const bool full = (k0 + 32 <= T);
    const int kk = k0 + l31;
    const int kc = kk < T ? kk : T - 1;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    // K[2 st + h][kc]: scalar row base + one 32-bit per-lane byte offset (32 per-lane 64-bit row addresses would cost 64 VGPRs)
    const unsigned koff = 4u * (unsigned)(h * T + kc);
#pragma unroll
    for (int st = 0; st < 32; ++st) {
      const float a = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(K + (size_t)(2 * st) * T) + koff);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qrow[64 * st], s, 0, 0, 0);
    }
    // s[r] = log2(e) * score(query l31, key k0 + (r&3) + 8*(r>>2) + 4*h)
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (!full && key >= T) s[r] = -INFINITY;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float p[16];
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      rs += p[r];
    }
    rs += __shfl_xor(rs, 32);
    l = l * alpha + rs;
    m = m_new;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* vrow = V + (size_t)(32 * i + l31) * T;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v4[4];
        const int kb = k0 + 8 * g + 4 * h;
        if (full && (T & 3) == 0) {
          const float4 t = *reinterpret_cast<const float4*>(vrow + kb);
          v4[0] = t.x; v4[1] = t.y; v4[2] = t.z; v4[3] = t.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] = vrow[(kb + e) < T ? (kb + e) : T - 1];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v4[e], p[4 * g + e], o[i], 0, 0, 0);
      }
    }
  }
  if (KW > 1) {
    if (wave > 0) {
      float* dst = part + (size_t)(wave - 1) * 34 * 64 + lane;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(16 * i + r) * 64] = o[i][r];
      dst[32 * 64] = m; dst[33 * 64] = l;
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 1; w < KW; ++w) {            // a wave that saw no key tile holds m = -inf, l = 0: it merges to nothing
      const float* src = part + (size_t)(w - 1) * 34 * 64 + lane;
      const float mw = src[32 * 64], lw = src[33 * 64];
      const float m_new = fmaxf(m, mw);
      const float a0 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - m_new);
      const float a1 = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - m_new);
      l = l * a0 + lw * a1;
      m = m_new;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = o[i][r] * a0 + src[(16 * i + r) * 64] * a1;
    }
  }
  if (q < T) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        out[(bh * 64 + c) * (size_t)T + q] = o[i][r] * inv;
      }
  }
}

// Long sequences (T >= 512: the 32 x 32 level of the ch = 64 networks, one head over 1024 tokens): the waves of a workgroup
// take four DIFFERENT query tiles and stream the SAME key / value tiles, which are staged once per workgroup in LDS
// (double-buffered, one barrier per tile) instead of being read from L2 by every wave: a quarter of the L2 traffic of the
// split kernel (512 MB per launch at B = 32) and no dependence on latency-hiding by occupancy.  KH = 2: eight waves, waves
// 4..7 take the odd 32-key tiles of the same four query tiles (two waves per SIMD: one wave's softmax runs under the
// other's MFMAs) and the halves are merged in LDS at the end, half 0 first (a fixed order: bitwise reproducible).
// K tile [64 ch][32 KH keys]; V tile transposed to [32 KH keys][64 ch] with pitch 65: both MFMA operand reads lane-linear.
template <int KH>
__global__ __launch_bounds__(256 * KH) void attention_lds_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int xmap) {
  constexpr int VP = 65, NK = 32 * KH;                     // pitch of the transposed V tile; keys per staged tile
  constexpr int KS_F = 64 * NK, VS_F = NK * VP;
  extern __shared__ float lds[];                           // ks[2][KS_F] | vs[2][VS_F]; reused for the merge
  float* ks = lds;
  float* vs = lds + 2 * KS_F;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = wave & 3, hf = wave >> 2;
  const int l31 = lane & 31, h = lane >> 5;
  int bx_, by_;
  xcd_group_map(xmap, bx_, by_);
  const int q0 = (bx_ * 4 + qt) * 32;
  const size_t bh = by_;
  const float* Q = qkv + (bh * 3 + 0) * 64 * (size_t)T;
  const float* K = qkv + (bh * 3 + 1) * 64 * (size_t)T;
  const float* V = qkv + (bh * 3 + 2) * 64 * (size_t)T;
  const int q = q0 + l31;
  const int qc = q < T ? q : T - 1;
  float qreg[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) qreg[s] = Q[(size_t)(2 * s + h) * T + qc] * (0.125f * 1.44269504088896340736f);

  // staging: 64 rows x 8 KH float4 per tile, two per thread: row c0 (+ 32), keys 4 kq .. 4 kq + 3
  const int c0 = tid / (8 * KH), kq = tid % (8 * KH);
  f32x4 rk[2], rv[2];
  auto fetch = [&](int k0) {                                 // T % 4 == 0 and 16-byte aligned rows (checked by the launcher)
    const int kk = min(k0 + 4 * kq, T - 4);                  // past the end: re-reads valid keys; they are masked below
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      rk[u] = *reinterpret_cast<const f32x4*>(K + (size_t)(c0 + 32 * u) * T + kk);
      rv[u] = *reinterpret_cast<const f32x4*>(V + (size_t)(c0 + 32 * u) * T + kk);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<f32x4*>(&ks[buf * KS_F + (c0 + 32 * u) * NK + 4 * kq]) = rk[u];
#pragma unroll
      for (int e = 0; e < 4; ++e) vs[buf * VS_F + (4 * kq + e) * VP + c0 + 32 * u] = rv[u][e];
    }
  };
  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  fetch(0);
  commit(0);
  __syncthreads();
  const int ntile = (T + NK - 1) / NK;
  for (int it = 0; it < ntile; ++it) {
    const int k0 = it * NK + 32 * hf, buf = it & 1;
    if (it + 1 < ntile) fetch((it + 1) * NK);
    const bool full = (k0 + 32 <= T);
    const float* kt = ks + buf * KS_F + h * NK + 32 * hf + l31;   // + 2 NK st: channel 2 st + h, key 32 hf + l31
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 32; ++st) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[2 * NK * st], qreg[st], s, 0, 0, 0);
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (!full && key >= T) s[r] = -INFINITY;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    if (m_new > -INFINITY) {                                 // a wave whose first tile lies wholly past T has nothing to add yet
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      float p[16];
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
        rs += p[r];
      }
      rs += __shfl_xor(rs, 32);
      l = l * alpha + rs;
      m = m_new;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      // O[c][q] += sum_key V[c][key] P[key][q]: MFMA step r contracts keys (r&3) + 8 (r>>2) + 4 h of the wave's 32-key tile
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float* vt = vs + buf * VS_F + 32 * hf * VP + 32 * i + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(vt[((r & 3) + 8 * (r >> 2) + 4 * h) * VP], p[r], o[i], 0, 0, 0);
      }
    }
    if (it + 1 < ntile) commit(buf ^ 1);                     // the other buffer: its last readers passed the previous barrier
    __syncthreads();
  }
  if (KH == 2) {                                             // merge the two key halves (all tiles are past their last barrier)
    float* mg = lds + (size_t)(qt * 64 + lane) * 35;         // pitch 35: conflict-free for lane-linear access
    if (hf == 1) {
      mg[0] = m;
      mg[1] = l;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) mg[2 + 16 * i + r] = o[i][r];
    }
    __syncthreads();
    if (hf == 0) {
      const float m1 = mg[0], l1 = mg[1];
      const float mm = fmaxf(m, m1);
      const float a0 = __builtin_amdgcn_exp2f(m - mm), a1 = m1 > -INFINITY ? __builtin_amdgcn_exp2f(m1 - mm) : 0.f;
      l = l * a0 + l1 * a1;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = o[i][r] * a0 + mg[2 + 16 * i + r] * a1;
    }
  }
  if (hf == 0 && q < T) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        out[(bh * 64 + c) * (size_t)T + q] = o[i][r] * inv;
      }
  }
}

template <int KH>
static int launch_attention_lds(const float* qkv, float* out, int B, int heads, int T, hipStream_t stream) {
  constexpr int lds_bytes = (2 * 64 * 32 * KH + 2 * 32 * KH * 65) * 4;
  static_assert(KH == 1 || lds_bytes >= 4 * 64 * 35 * 4, "merge scratch");
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  MCEDM_HIP_TRY(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && !attr_set[dev].load(std::memory_order_acquire)) {
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)attention_lds_kernel<KH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set[dev].store(true, std::memory_order_release);
  }
  static int xmap = -1;                                    // MCEDM_ATTN_XCD=0: workgroups in dispatch order (A/B runs)
  if (xmap < 0) { const char* e = getenv("MCEDM_ATTN_XCD"); xmap = e ? atoi(e) : 1; }
  hipLaunchKernelGGL(attention_lds_kernel<KH>, dim3(ceil_div(T, 128), B * heads), dim3(256 * KH), lds_bytes, stream, qkv, out, T, xmap);
  return MCEDM_OK;
}

static int attn_split_env() {          // MCEDM_ATTN_SPLIT=0: the one-wave-per-query-tile kernel everywhere (A/B runs)
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_ATTN_SPLIT"); env = e ? atoi(e) : 1; }
  return env;
}

static int attn_lds_min_env() {        // MCEDM_ATTN_LDS_MIN: shortest sequence that takes the LDS-staged kernel (A/B runs)
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_ATTN_LDS_MIN"); env = e ? atoi(e) : 512; }
  return env;
}

int launch_attention(const float* qkv, float* out, int B, int heads, int T, hipStream_t stream) {
  MCEDM_REQUIRE(B > 0 && heads > 0 && T > 0, "attention: empty shape");
  MCEDM_REQUIRE((long long)B * heads <= 65535, "attention: B*heads too large for grid.y");
  ProfScope ps("attention_kernel", 4.0 * B * heads * (double)T * T * 64, 4.0 * 4 * B * heads * 64.0 * T, stream);
  // the split factor is a function of T only (never of the batch size): results are identical under batch sharding
  const dim3 grid(ceil_div(T, 32), B * heads);
  // LDS-staged kernel for long sequences, chosen by T alone like the split factor (at B * heads < 24 it fills fewer CUs than
  // the split kernel would: 91 vs 45 us at B = 8 -- accepted for the batch-sharding invariance).  MCEDM_ATTN_SPLIT=2: split
  // kernel for every T, 3: the four-wave variant of the LDS kernel (A/B switches)
  const bool aligned = (T % 4 == 0) && ((reinterpret_cast<size_t>(qkv) & 15) == 0);
  const int mode = attn_split_env();
  if (mode >= 1 && mode != 2 && T >= attn_lds_min_env() && aligned) {
    const int rc = mode == 3 ? launch_attention_lds<1>(qkv, out, B, heads, T, stream) : launch_attention_lds<2>(qkv, out, B, heads, T, stream);
    if (rc != MCEDM_OK) return rc;
  }
  else if (attn_split_env() && T >= 256) hipLaunchKernelGGL(attention_split_kernel<4>, grid, dim3(256), 0, stream, qkv, out, T);
  else if (attn_split_env() && T >= 128) hipLaunchKernelGGL(attention_split_kernel<2>, grid, dim3(128), 0, stream, qkv, out, T);
  else hipLaunchKernelGGL(attention_kernel, grid, dim3(64), 0, stream, qkv, out, T);
  MCEDM_LAUNCH_CHECK("attention_kernel");
  return MCEDM_OK;
}

}  // namespace mcedm
