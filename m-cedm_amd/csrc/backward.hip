// backward.hip -- mcedm_edm_denoise_backward: reverse schedule of the U-Net + EDM preconditioning.
// Consumes the activations, GroupNorm statistics and transform tables the training-mode forward left in the
// workspace (nothing is freed in that mode) and writes dLoss/dparam for every parameter.
//
// Per convolution (in reverse): weight/bias gradient (wgrad_mfma.hip, input transform recomputed on the fly),
// data gradient (the forward conv kernel on transposed + mirrored packed weights), then GroupNorm/FiLM/SiLU
// backward (bwd_kernels.hip) which also adds the skip-path gradient and accumulates into tensors that feed
// several consumers (U-Net skip connections).
#include <algorithm>
#include <cmath>
#include <vector>

#include "bwd.hpp"
#include "plan.hpp"

namespace mcedm {

static inline int grid_for(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b < 2048 ? (b ? b : 1) : 2048);
}

// dF = c_out[n] * dD     (D = c_skip x + c_out F, mcedm.py:210)
__global__ void scale_by_cout_kernel(const float* __restrict__ dD, const float* __restrict__ coefs4, int n_sigma,
                                     size_t per_sample, size_t total, float* __restrict__ dF) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    dF[i] = dD[i] * coefs4[4 * (n_sigma == 1 ? 0 : i / per_sample) + 1];
}

// recompute the embedding MLP and keep its pre-activations: pe, u1 = W0 pe + b0, u2 = W1 silu(u1) + b1
__device__ __forceinline__ float silu_b(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float dsilu_b(float v) { const float s = 1.0f / (1.0f + expf(-v)); return s * (1.0f + v * (1.0f - s)); }

__global__ __launch_bounds__(256) void emb_save_kernel(const float* __restrict__ labels, const float* __restrict__ freqs,
                                                       const float* __restrict__ w0, const float* __restrict__ b0,
                                                       const float* __restrict__ w1, const float* __restrict__ b1, int ch,
                                                       float* __restrict__ pe, float* __restrict__ u1,
                                                       float* __restrict__ u2) {
  extern __shared__ float sm[];
  float* e0 = sm; float* e1 = sm + ch;
  const int n = blockIdx.x, tid = threadIdx.x, half = ch / 2;
  const float x = labels[n];
  for (int k = tid; k < ch; k += 256) {
    const float arg = x * freqs[k < half ? k : k - half];
    const float v = (k < half) ? cosf(arg) : sinf(arg);
    e0[k] = v; pe[(size_t)n * ch + k] = v;
  }
  __syncthreads();
  for (int j = tid; j < ch; j += 256) {
    float s = 0.f;
    for (int k = 0; k < ch; ++k) s = fmaf(e0[k], w0[(size_t)j * ch + k], s);
    s += b0[j];
    u1[(size_t)n * ch + j] = s; e1[j] = silu_b(s);
  }
  __syncthreads();
  for (int j = tid; j < ch; j += 256) {
    float s = 0.f;
    for (int k = 0; k < ch; ++k) s = fmaf(e1[k], w1[(size_t)j * ch + k], s);
    u2[(size_t)n * ch + j] = s + b1[j];
  }
}

// out = silu(u) (mode 0)  |  out = g * silu'(u) (mode 1)
__global__ void silu_map_kernel(const float* __restrict__ u, const float* __restrict__ g, float* __restrict__ out, size_t n,
                                int mode) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = mode ? g[i] * dsilu_b(u[i]) : silu_b(u[i]);
}

// [B, 2c, hw] -> two dense [B, c, hw] tensors (the data gradient of combine_enc's concatenated input)
__global__ void split_channels_kernel(const float* __restrict__ src, size_t half, size_t total, float* __restrict__ a,
                                      float* __restrict__ b) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / (2 * half), r = i - n * 2 * half;
    if (r < half) a[n * half + r] = src[i]; else b[n * half + (r - half)] = src[i];
  }
}

// out[c] = sum_r A[r][c]
__global__ void colsum_kernel(const float* __restrict__ A, int rows, int cols, int lda, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
#pragma unroll 8
  for (int r = 0; r < rows; ++r) s += A[(size_t)r * lda + c];
  out[c] = s;
}

struct BwdScratch {
  std::vector<size_t> g;     // gradient buffer per forward tensor id (NONE when not an activation)
  size_t dact, dskip, xact, ab, wg, dfilm, lse, gF, pe, u1, u2, t1, t2, t3, emb, sync, sync_bytes, total;
};

static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

static BwdScratch make_scratch(const mcedm_plan& P, const Layout& L, int B, int H, int W) {
  BwdScratch S;
  size_t cur = 0;
  auto take = [&](size_t bytes) { size_t o = cur; cur += align_up(bytes ? bytes : 4, 256); return o; };
  S.g.assign(L.t.size(), NONE);
  auto want = [&](int id) { if (id >= 0 && S.g[id] == NONE) S.g[id] = take(L.t[id].bytes); };
  want(L.t0);
  size_t max_dact = 0, max_wg = 0, max_c = 0, max_lse = 0;
  size_t bi = 0;
  auto conv_scr = [&](const ConvP& c) { max_wg = max_sz(max_wg, wgrad_scratch_floats(c.cout, c.cin, c.taps)); };
  conv_scr(P.conv_in); conv_scr(P.conv_out);
  if (P.desc.dx_mode == MCEDM_DX_ENC) {      // gradients of conv_in's output, of dx_enc's output and of GELU(dx_enc.0)
    want(L.xf); want(L.d2); want(L.g1);
    conv_scr(P.dx_enc0); conv_scr(P.dx_enc2); conv_scr(P.combine);
    max_dact = (size_t)B * 2 * P.conv_in.cout * H * W;      // combine_enc's data gradient, both halves
  }
  for (auto* v : {&P.enc, &P.dec})
    for (const BlockP& b : *v) {
      const BlockLayout& bl = L.blocks[bi++];
      want(bl.h); want(bl.y); want(bl.qkv); want(bl.a); want(bl.z);
      conv_scr(b.conv0); conv_scr(b.conv1);
      if (b.skip_kernel == 1) conv_scr(b.skip);
      if (b.attn) { conv_scr(b.qkv); conv_scr(b.proj); max_lse = max_sz(max_lse, (size_t)B * b.heads * bl.H * bl.W * 2); }
      max_dact = max_sz(max_dact, (size_t)B * b.cin * bl.H * bl.W);
      max_dact = max_sz(max_dact, (size_t)B * b.cout * bl.H * bl.W);
      max_c = max_sz(max_c, (size_t)std::max(b.cin, b.cout));
    }
  max_dact = max_sz(max_dact, (size_t)B * P.out_norm.C * H * W);
  max_c = max_sz(max_c, (size_t)P.out_norm.C);
  const int ch = P.desc.ch;
  S.dact = take(max_dact * 4);
  S.dskip = take(max_dact * 4);
  S.xact = take(max_dact * 4);
  S.ab = take((size_t)B * max_c * 2 * 4);
  S.wg = take(max_wg * 4);
  S.dfilm = take((size_t)B * P.film_rows * 4);
  S.lse = take(max_lse * 4);
  S.gF = take((size_t)B * P.desc.out_channels * H * W * 4);
  S.pe = take((size_t)B * ch * 4); S.u1 = take((size_t)B * ch * 4); S.u2 = take((size_t)B * ch * 4);
  S.t1 = take((size_t)B * ch * 4); S.t2 = take((size_t)B * ch * 4); S.t3 = take((size_t)B * ch * 4);
  S.emb = take((size_t)B * ch * 4);
  S.sync_bytes = gn_bwd_sync_words(B, 32) * sizeof(unsigned);      // gn_bwd_lds_kernel's counters and piece sums (<= 32 groups)
  S.sync = take(S.sync_bytes);
  S.total = cur;
  return S;
}

size_t backward_scratch_bytes(const mcedm_plan& P, const Layout& L, int B, int H, int W) {
  return make_scratch(P, L, B, H, W).total;
}

struct Ctx;
static GnBwdArgs with_sync(const Ctx& c, GnBwdArgs g);

struct Ctx {
  const mcedm_plan& P;
  const Layout& L;
  const BwdScratch& S;
  char* act;        // forward activations
  char* scr;        // backward scratch
  const float* pk;
  float* const* grads;
  int B, n_noise;
  hipStream_t s;
  std::vector<char> have;   // gradient buffer of tensor id already holds a contribution
  const float* emb = nullptr;                 // [n_noise][ch] sigma embedding (recomputed up front)
  int n_buckets = 0;                          // overlap hook: record bucket_events[k] once every parameter with index
  const int32_t* bucket_first = nullptr;      // >= bucket_first[k] has its gradient enqueued
  void* const* bucket_events = nullptr;

  float* T(int id) const { return id < 0 ? nullptr : reinterpret_cast<float*>(act + L.t[id].off); }
  Coef* CF(int id) const { return id < 0 ? nullptr : reinterpret_cast<Coef*>(act + L.t[id].off); }
  float* G(int id) const { return id < 0 ? nullptr : reinterpret_cast<float*>(scr + S.g[id]); }
  float* X(size_t off) const { return reinterpret_cast<float*>(scr + off); }
};
static GnBwdArgs with_sync(const Ctx& c, GnBwdArgs g) { g.sync = reinterpret_cast<unsigned*>(c.X(c.S.sync)); return g; }

// plain data-gradient conv: out[B, c.cin, H, W] = conv(dy[B, c.cout, H, W], transposed+mirrored weights)
static int dgrad(const Ctx& c, const ConvP& cv, const float* dy, int H, int W, float* out) {
  ConvArgs a{};
  a.xa = dy; a.Ca = cv.cout;
  a.Hs = H; a.Ws = W; a.H = H; a.W = W;
  a.wpk = c.pk + cv.wpk_dgrad;
  a.wino = cv.wino_dgrad != NONE ? c.pk + cv.wino_dgrad : nullptr;
  a.out = out; a.Cout = cv.cin; a.B = c.B;
  return launch_conv(a, cv.taps, c.s);
}

static int norm_param_grads(const Ctx& c, const NormP& nm, const float* film, int film_stride, float* dfilm) {
  return launch_gn_param_grads(c.X(c.S.ab), c.pk + nm.gamma, c.pk + nm.beta, film, c.n_noise > 1 ? 1 : 0, film_stride,
                               c.B, nm.C, c.grads[nm.w], c.grads[nm.b], dfilm, c.P.film_rows, c.s);
}

static int block_backward(Ctx& c, const BlockP& b, const BlockLayout& bl) {
  int rc;
  const int B = c.B, H = bl.H, W = bl.W;
  const float* xa = c.T(bl.xa);
  const float* xb = c.T(bl.xb);
  const int Ca = c.L.t[bl.xa].C, Cb = bl.xb >= 0 ? c.L.t[bl.xb].C : 0;
  const int rs = b.up ? RS_UP : (b.down ? RS_DOWN : RS_NONE);
  float* dact = c.X(c.S.dact);
  float* wg = c.X(c.S.wg);
  const float* dy = c.G(bl.out);      // gradient of the block output (every consumer has already contributed)
  if (b.attn) {
    const float* dz = c.G(bl.z);
    // z = proj(a) + y
    WgradArgs wp{dz, c.T(bl.a), nullptr, b.cout, 0, nullptr, 0, 0, RS_NONE, H, W, H, W, b.cout, B, wg, nullptr};
    if ((rc = launch_wgrad(wp, 1, c.grads[b.proj.w], c.grads[b.proj.b], 0, c.X(c.S.xact), c.s))) return rc;
    if ((rc = dgrad(c, b.proj, dz, H, W, c.G(bl.a)))) return rc;
    if ((rc = launch_attention_bwd(c.T(bl.qkv), c.T(bl.a), c.G(bl.a), c.G(bl.qkv), c.X(c.S.lse), B, b.heads, H * W, c.s))) return rc;
    // qkv = conv1x1(norm2(y))   (rows in packed order)
    // order: data gradient -> GroupNorm backward (which also writes the conv's normalised input, the weight gradient's
    // operand: no separate materialisation pass) -> weight gradient
    if ((rc = dgrad(c, b.qkv, c.G(bl.qkv), H, W, dact))) return rc;
    GnBwdArgs g2{dact, RS_NONE, c.T(bl.y), nullptr, b.cout, 0, H, W, B, b.norm2.groups, c.CF(bl.coef2), c.T(bl.stats2),
                 c.pk + b.norm2.gamma, nullptr, 0, 0, 0, c.G(bl.y), nullptr, 0, dz, 1, b.cout, c.X(c.S.ab), c.X(c.S.xact)};
    if ((rc = launch_gn_bwd(with_sync(c, g2), c.s))) return rc;     // g[y] = dz (residual) + norm2 path
    if ((rc = norm_param_grads(c, b.norm2, nullptr, 0, nullptr))) return rc;
    WgradArgs wq{c.G(bl.qkv), c.T(bl.y), nullptr, b.cout, 0, c.CF(bl.coef2), 1, 0, RS_NONE, H, W, H, W, 3 * b.cout, B, wg, nullptr};
    if ((rc = launch_wgrad(wq, 1, c.grads[b.qkv.w], c.grads[b.qkv.b], b.heads, c.X(c.S.xact), c.s, true))) return rc;
    dy = c.G(bl.y);
  }
  // y = conv1(silu(film(norm1(h)))) + skip(x)
  if ((rc = dgrad(c, b.conv1, dy, H, W, dact))) return rc;
  const float* film = reinterpret_cast<const float*>(c.act + c.L.t[c.L.film].off) + b.film_row0;
  GnBwdArgs g1{dact, RS_NONE, c.T(bl.h), nullptr, b.cout, 0, H, W, B, b.norm1.groups, c.CF(bl.coef1), c.T(bl.stats1),
               c.pk + b.norm1.gamma, film, c.n_noise > 1 ? 1 : 0, c.P.film_rows, 1, c.G(bl.h), nullptr, 0, nullptr, 0, 0,
               c.X(c.S.ab), c.X(c.S.xact)};
  if ((rc = launch_gn_bwd(with_sync(c, g1), c.s))) return rc;
  if ((rc = norm_param_grads(c, b.norm1, film, c.P.film_rows, c.X(c.S.dfilm) + b.film_row0))) return rc;
  WgradArgs w1{dy, c.T(bl.h), nullptr, b.cout, 0, c.CF(bl.coef1), 1, 1, RS_NONE, H, W, H, W, b.cout, B, wg, nullptr};
  if ((rc = launch_wgrad(w1, 9, c.grads[b.conv1.w], c.grads[b.conv1.b], 0, c.X(c.S.xact), c.s, true))) return rc;
  // skip path -> extra gradient for x
  const float* add = dy;
  int add_mode = 1;
  if (b.skip_kernel == 1) {
    if (rs != RS_NONE) { set_error("backward: resampling 1x1 skip conv is outside the hot path (%s)", b.key.c_str()); return MCEDM_ERR_UNSUPPORTED; }
    WgradArgs wsk{dy, xa, xb, Ca, Cb, nullptr, 0, 0, RS_NONE, H, W, H, W, b.cout, B, wg, nullptr};
    if ((rc = launch_wgrad(wsk, 1, c.grads[b.skip.w], c.grads[b.skip.b], 0, c.X(c.S.xact), c.s))) return rc;
    if ((rc = dgrad(c, b.skip, dy, H, W, c.X(c.S.dskip)))) return rc;
    add = c.X(c.S.dskip);
  } else if (b.skip_kernel == 0) {
    add_mode = 2;     // resample-only skip: map dy back through the same resampling
  }
  // h = conv0(resample(silu(norm0(x))))
  const bool fuse0 = rs == RS_NONE;        // the un-resampled conv's operand comes out of the GroupNorm backward
  WgradArgs w0{c.G(bl.h), xa, xb, Ca, Cb, c.CF(bl.coef0), 1, 1, rs, bl.Hin, bl.Win, H, W, b.cout, B, wg, nullptr};
  if (!fuse0 && (rc = launch_wgrad(w0, 9, c.grads[b.conv0.w], c.grads[b.conv0.b], 0, c.X(c.S.xact), c.s))) return rc;
  if ((rc = dgrad(c, b.conv0, c.G(bl.h), H, W, dact))) return rc;
  GnBwdArgs g0{dact, rs, xa, xb, Ca, Cb, bl.Hin, bl.Win, B, b.norm0.groups, c.CF(bl.coef0), c.T(bl.stats0),
               c.pk + b.norm0.gamma, nullptr, 0, 0, 1, c.G(bl.xa), c.G(bl.xb), 0, add, add_mode, b.cin, c.X(c.S.ab),
               fuse0 ? c.X(c.S.xact) : nullptr};
  // both halves of a concat input share one accumulate flag: run the kernel once per distinct state
  const bool ha = c.have[bl.xa] != 0, hb = bl.xb >= 0 ? c.have[bl.xb] != 0 : ha;
  if (bl.xb >= 0 && ha != hb) {
    // mixed state: bring the fresh buffer to "already holds zeros" so one accumulate pass is exact
    const int fresh = ha ? bl.xb : bl.xa;
    MCEDM_HIP_TRY(hipMemsetAsync(c.G(fresh), 0, c.L.t[fresh].bytes, c.s));
    g0.accumulate = 1;
  } else {
    g0.accumulate = ha ? 1 : 0;
  }
  if ((rc = launch_gn_bwd(with_sync(c, g0), c.s))) return rc;
  c.have[bl.xa] = 1;
  if (bl.xb >= 0) c.have[bl.xb] = 1;
  if ((rc = norm_param_grads(c, b.norm0, nullptr, 0, nullptr))) return rc;
  if (fuse0 && (rc = launch_wgrad(w0, 9, c.grads[b.conv0.w], c.grads[b.conv0.b], 0, c.X(c.S.xact), c.s, true))) return rc;
  // this block's affine layer (film = emb Waff^T + baff): dWaff[r][k] = sum_n dfilm[n][r] emb[n][k], dbaff[r] = sum_n dfilm[n][r].
  // Done here, not after the last block, so that the block's whole parameter range is complete (gradient buckets).
  const int ch = c.P.desc.ch, R = c.P.film_rows;
  const float* dfilm = c.X(c.S.dfilm) + b.film_row0;
  if ((rc = launch_small_gemm(dfilm, c.emb, c.grads[b.aff_w], 2 * b.cout, ch, c.n_noise, R, ch, ch, 1, 0, 0, c.s))) return rc;
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(2 * b.cout, 64)), dim3(64), 0, c.s, dfilm, c.n_noise, 2 * b.cout, R,
                     c.grads[b.aff_b]);
  MCEDM_LAUNCH_CHECK("colsum_kernel");
  for (int k = 0; k < c.n_buckets; ++k)
    if (c.bucket_first[k] == b.norm0.w) MCEDM_HIP_TRY(hipEventRecord((hipEvent_t)c.bucket_events[k], c.s));
  return MCEDM_OK;
}

}  // namespace mcedm

using namespace mcedm;

// first parameter index of every unit whose gradients complete together, in completion order of the backward:
// (out_norm, out_conv), decoder blocks reversed, encoder blocks reversed, conv_in, the mapping MLP (index 0)
static std::vector<int> bucket_candidates(const mcedm_plan& P) {
  std::vector<int> c;
  for (size_t i = P.dec.size(); i-- > 0;) c.push_back(P.dec[i].norm0.w);
  for (size_t i = P.enc.size(); i-- > 0;) c.push_back(P.enc[i].norm0.w);
  c.push_back(0);
  return c;
}

extern "C" int mcedm_unet_grad_buckets(const mcedm_plan* plan, int max_buckets, int32_t* first_param, int* n_buckets) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && first_param && n_buckets && max_buckets >= 1, "grad_buckets: bad argument");
  const mcedm_plan& P = *plan;
  int64_t total = 0;
  for (const ParamInfo& pi : P.params) total += pi.numel;
  const std::vector<int> cand = bucket_candidates(P);
  std::vector<int64_t> prefix(P.params.size() + 1, 0);
  for (size_t i = 0; i < P.params.size(); ++i) prefix[i + 1] = prefix[i] + P.params[i].numel;
  int n = 0;
  int hi = (int)P.params.size();
  for (int cfirst : cand) {
    if (n == max_buckets - 1) break;
    if (cfirst == 0) break;
    if (prefix[hi] - prefix[cfirst] >= total / max_buckets) { first_param[n++] = cfirst; hi = cfirst; }
  }
  first_param[n++] = 0;
  *n_buckets = n;
  return MCEDM_OK;
}

static int denoise_backward_impl(const mcedm_plan* plan, const void* packed, const float* const* params,
                                 const float* x, const float* dx, int n_sigma, const float* cond,
                                 const float* dD, float* const* grads, void* workspace, size_t workspace_bytes,
                                 int B, int H, int W, int n_buckets, const int32_t* bucket_first,
                                 void* const* bucket_events, void* stream) {
  MCEDM_REQUIRE(plan && packed && params && x && dD && grads && workspace, "denoise_backward: null argument");
  const mcedm_plan& P = *plan;
  MCEDM_REQUIRE(dx == nullptr || P.desc.dx_mode != MCEDM_DX_NONE, "denoise_backward: dx given to a plan without dx_cond");
  if (n_buckets > 0) {
    MCEDM_REQUIRE(bucket_first && bucket_events, "denoise_backward: null bucket arrays");
    const std::vector<int> cand = bucket_candidates(P);
    for (int k = 0; k < n_buckets; ++k) {
      MCEDM_REQUIRE(bucket_events[k] != nullptr, "denoise_backward: bucket event %d is null", k);
      MCEDM_REQUIRE(std::find(cand.begin(), cand.end(), bucket_first[k]) != cand.end(),
                    "denoise_backward: bucket %d starts at parameter %d, which is not the first parameter of a block", k, bucket_first[k]);
      MCEDM_REQUIRE(k == 0 || bucket_first[k] < bucket_first[k - 1], "denoise_backward: bucket starts must decrease");
    }
    MCEDM_REQUIRE(bucket_first[n_buckets - 1] == 0, "denoise_backward: the last bucket must start at parameter 0");
  }
  for (size_t i = 0; i < P.params.size(); ++i)
    MCEDM_REQUIRE(grads[i] != nullptr, "denoise_backward: grads[%zu] (%s) is null", i, P.params[i].name.c_str());
  Layout L;
  int rc = build_layout(P, B, H, W, 1, n_sigma, &L);
  if (rc) return rc;
  const Header hd = header_for(P, B, H, W);
  const BwdScratch S = make_scratch(P, L, B, H, W);
  if (hd.total + L.total_bytes + S.total > workspace_bytes) {
    set_error("denoise_backward: workspace too small (%zu < %zu bytes)", workspace_bytes, hd.total + L.total_bytes + S.total);
    return MCEDM_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  char* act = at<char>(workspace, hd.total);
  char* scr = act + L.total_bytes;
  const float* pk = (const float*)packed;
  Ctx c{P, L, S, act, scr, pk, grads, B, n_sigma, s, std::vector<char>(L.t.size(), 0)};
  c.n_buckets = n_buckets; c.bucket_first = bucket_first; c.bucket_events = bucket_events;
  MCEDM_HIP_TRY(hipMemsetAsync(c.X(S.sync), 0, S.sync_bytes, s));      // zero once per pass; every GroupNorm backward leaves it zero
  const int ch = P.desc.ch;
  const size_t per = (size_t)P.desc.out_channels * H * W;

  // sigma embedding and its pre-activations (the blocks' affine gradients need emb): pe, u1 = W0 pe + b0,
  // u2 = W1 silu(u1) + b1, emb = silu(u2)
  const int n = n_sigma, R = P.film_rows;
  float* dfilm = c.X(S.dfilm);
  float* pe = c.X(S.pe); float* u1 = c.X(S.u1); float* u2 = c.X(S.u2);
  float* t1 = c.X(S.t1); float* t2 = c.X(S.t2); float* t3 = c.X(S.t3);
  float* emb = c.X(S.emb);
  hipLaunchKernelGGL(emb_save_kernel, dim3(n), dim3(256), 2 * ch * sizeof(float), s, at<float>(workspace, hd.c_noise),
                     pk + P.freqs, pk + P.w0, pk + P.b0, pk + P.w1, pk + P.b1, ch, pe, u1, u2);
  MCEDM_LAUNCH_CHECK("emb_save_kernel");
  const size_t ne = (size_t)n * ch;
  const int ge = (int)((ne + 255) / 256);
  hipLaunchKernelGGL(silu_map_kernel, dim3(ge), dim3(256), 0, s, u2, nullptr, emb, ne, 0);
  MCEDM_LAUNCH_CHECK("silu_map_kernel");
  c.emb = emb;

  // dF = c_out * dD
  hipLaunchKernelGGL(scale_by_cout_kernel, dim3(grid_for(per * B)), dim3(256), 0, s, dD, at<float>(workspace, hd.coefs4),
                     n_sigma, per, per * B, c.X(S.gF));
  MCEDM_LAUNCH_CHECK("scale_by_cout_kernel");
  // out = out_conv(silu(out_norm(last)))
  const TRef& last = L.t[L.last];
  WgradArgs wo{c.X(S.gF), c.T(L.last), nullptr, last.C, 0, c.CF(L.coef_out), 1, 1, RS_NONE, H, W, H, W,
               P.desc.out_channels, B, c.X(S.wg), nullptr};
  if ((rc = launch_wgrad(wo, 9, grads[P.conv_out.w], grads[P.conv_out.b], 0, c.X(c.S.xact), s))) return rc;
  if ((rc = dgrad(c, P.conv_out, c.X(S.gF), H, W, c.X(S.dact)))) return rc;
  GnBwdArgs go{c.X(S.dact), RS_NONE, c.T(L.last), nullptr, last.C, 0, H, W, B, P.out_norm.groups, c.CF(L.coef_out),
               c.T(L.stats_out), pk + P.out_norm.gamma, nullptr, 0, 0, 1, c.G(L.last), nullptr, 0, nullptr, 0, 0, c.X(S.ab)};
  if ((rc = launch_gn_bwd(with_sync(c, go), s))) return rc;
  c.have[L.last] = 1;
  if ((rc = norm_param_grads(c, P.out_norm, nullptr, 0, nullptr))) return rc;

  // blocks in reverse execution order
  const size_t nenc = P.enc.size(), ndec = P.dec.size();
  for (size_t i = ndec; i-- > 0;)
    if ((rc = block_backward(c, P.dec[i], L.blocks[nenc + i]))) return rc;
  for (size_t i = nenc; i-- > 0;)
    if ((rc = block_backward(c, P.enc[i], L.blocks[i]))) return rc;

  // dx_cond head (adm_blocks.py:352-362), in reverse: t0 = combine_enc(cat(xf, d2)), d2 = dx_enc.2(GELU(dx_enc.0(dx)))
  const float* dy_in = c.G(L.t0);
  if (P.desc.dx_mode == MCEDM_DX_ENC) {
    const int c0 = P.conv_in.cout;
    const size_t half = (size_t)c0 * H * W;
    WgradArgs wc{c.G(L.t0), c.T(L.xf), c.T(L.d2), c0, c0, nullptr, 0, 0, RS_NONE, H, W, H, W, c0, B, c.X(S.wg), nullptr};
    if ((rc = launch_wgrad(wc, 9, grads[P.combine.w], grads[P.combine.b], 0, c.X(c.S.xact), s))) return rc;
    if ((rc = dgrad(c, P.combine, c.G(L.t0), H, W, c.X(S.dact)))) return rc;
    hipLaunchKernelGGL(split_channels_kernel, dim3(grid_for(2 * half * B)), dim3(256), 0, s, c.X(S.dact), half, 2 * half * B,
                       c.G(L.xf), c.G(L.d2));
    MCEDM_LAUNCH_CHECK("split_channels_kernel");
    if (dx) {
      WgradArgs w2{c.G(L.d2), c.T(L.g1), nullptr, c0, 0, nullptr, 0, 0, RS_NONE, H, W, H, W, c0, B, c.X(S.wg), nullptr};
      if ((rc = launch_wgrad(w2, 9, grads[P.dx_enc2.w], grads[P.dx_enc2.b], 0, c.X(c.S.xact), s))) return rc;
      if ((rc = dgrad(c, P.dx_enc2, c.G(L.d2), H, W, c.G(L.g1)))) return rc;
      if ((rc = launch_gelu(c.T(L.d1), c.G(L.g1), c.G(L.g1), (size_t)B * half, 1, s))) return rc;      // in place: d GELU
      WgradArgs w1{c.G(L.g1), dx, nullptr, P.desc.dx_channels, 0, nullptr, 0, 0, RS_NONE, H, W, H, W, c0, B, c.X(S.wg), nullptr};
      if ((rc = launch_wgrad(w1, 9, grads[P.dx_enc0.w], grads[P.dx_enc0.b], 0, c.X(c.S.xact), s))) return rc;
    } else {                                  // dx None: zero features, no path to dx_enc (adm_blocks.py:355-357)
      for (const ConvP* cv : {&P.dx_enc0, &P.dx_enc2}) {
        MCEDM_HIP_TRY(hipMemsetAsync(grads[cv->w], 0, (size_t)P.params[cv->w].numel * sizeof(float), s));
        MCEDM_HIP_TRY(hipMemsetAsync(grads[cv->b], 0, (size_t)P.params[cv->b].numel * sizeof(float), s));
      }
    }
    dy_in = c.G(L.xf);
  }
  // conv_in: weight / bias gradient only (its inputs carry no gradient)
  const bool catdx = P.desc.dx_mode == MCEDM_DX_CAT;
  WgradArgs wi{dy_in, cond, catdx ? c.T(L.xdx) : x, P.desc.cond_channels, P.desc.in_channels + (catdx ? P.desc.dx_channels : 0),
               at<Coef>(workspace, hd.coef_in), n_sigma > 1 ? 1 : 0, 0, RS_NONE, H, W, H, W, P.conv_in.cout, B, c.X(S.wg), nullptr};
  if ((rc = launch_wgrad(wi, 9, grads[P.conv_in.w], grads[P.conv_in.b], 0, c.X(c.S.xact), s))) return rc;

  // mapping MLP: film = emb Waff^T + baff, emb = silu(u2), u2 = W1 silu(u1) + b1, u1 = W0 pe + b0
  // demb[n][k] = sum_r dfilm[n][r] Waff[r][k]
  if ((rc = launch_small_gemm(dfilm, pk + P.waff, t2, n, ch, R, R, ch, ch, 0, 0, 0, s))) return rc;       // t2 = demb
  hipLaunchKernelGGL(silu_map_kernel, dim3(ge), dim3(256), 0, s, u2, t2, t3, ne, 1);                      // t3 = du2
  hipLaunchKernelGGL(silu_map_kernel, dim3(ge), dim3(256), 0, s, u1, nullptr, t1, ne, 0);                 // t1 = e1
  MCEDM_LAUNCH_CHECK("silu_map_kernel");
  if ((rc = launch_small_gemm(t3, t1, grads[P.map1_w], ch, ch, n, ch, ch, ch, 1, 0, 0, s))) return rc;    // dW1 = du2^T e1
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(ch, 64)), dim3(64), 0, s, t3, n, ch, ch, grads[P.map1_b]);
  if ((rc = launch_small_gemm(t3, pk + P.w1, t2, n, ch, ch, ch, ch, ch, 0, 0, 0, s))) return rc;          // t2 = de1 = du2 W1
  hipLaunchKernelGGL(silu_map_kernel, dim3(ge), dim3(256), 0, s, u1, t2, t3, ne, 1);                      // t3 = du1
  if ((rc = launch_small_gemm(t3, pe, grads[P.map0_w], ch, ch, n, ch, ch, ch, 1, 0, 0, s))) return rc;    // dW0 = du1^T pe
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(ch, 64)), dim3(64), 0, s, t3, n, ch, ch, grads[P.map0_b]);
  MCEDM_LAUNCH_CHECK("colsum_kernel");
  if (n_buckets > 0) MCEDM_HIP_TRY(hipEventRecord((hipEvent_t)bucket_events[n_buckets - 1], s));
  return MCEDM_OK;
}

extern "C" int mcedm_edm_denoise_backward(const mcedm_plan* plan, const void* packed, const float* const* params,
                                          const float* x, const float* sigma, int n_sigma, const float* cond,
                                          const float* dD, float* const* grads, void* workspace, size_t workspace_bytes,
                                          int B, int H, int W, double sigma_data, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  (void)sigma; (void)sigma_data;
  return denoise_backward_impl(plan, packed, params, x, nullptr, n_sigma, cond, dD, grads, workspace, workspace_bytes, B, H, W, 0,
                               nullptr, nullptr, stream);
}

extern "C" int mcedm_edm_denoise_backward_dx(const mcedm_plan* plan, const void* packed, const float* const* params,
                                             const float* x, const float* dx, const float* sigma, int n_sigma,
                                             const float* cond, const float* dD, float* const* grads, void* workspace,
                                             size_t workspace_bytes, int B, int H, int W, double sigma_data, int n_buckets,
                                             const int32_t* bucket_first_param, void* const* bucket_events, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  (void)sigma; (void)sigma_data;
  MCEDM_REQUIRE(n_buckets >= 0, "denoise_backward_dx: n_buckets must be >= 0");
  return denoise_backward_impl(plan, packed, params, x, dx, n_sigma, cond, dD, grads, workspace, workspace_bytes, B, H, W,
                               n_buckets, bucket_first_param, bucket_events, stream);
}

extern "C" int mcedm_edm_denoise_backward_bucketed(const mcedm_plan* plan, const void* packed, const float* const* params,
                                                   const float* x, const float* sigma, int n_sigma, const float* cond,
                                                   const float* dD, float* const* grads, void* workspace,
                                                   size_t workspace_bytes, int B, int H, int W, double sigma_data,
                                                   int n_buckets, const int32_t* bucket_first_param,
                                                   void* const* bucket_events, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  (void)sigma; (void)sigma_data;
  MCEDM_REQUIRE(n_buckets >= 1, "denoise_backward_bucketed: n_buckets must be >= 1");
  return denoise_backward_impl(plan, packed, params, x, nullptr, n_sigma, cond, dD, grads, workspace, workspace_bytes, B, H, W,
                               n_buckets, bucket_first_param, bucket_events, stream);
}
