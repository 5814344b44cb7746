// ddpm.hip -- SURVEY.md section 8 f1: the DDPM U-Net (models/ddim_blocks.py:222-470) and the RePaint-style EDM sampler
// of PlDdim (models/ddim.py:915-1051) on the kernels of this library.
//
// The network is the ermongroup/ddim U-Net: ResnetBlock = GroupNorm(32, eps 1e-6) -> swish -> conv3x3 -> + temb_proj(swish(temb))
// -> GroupNorm -> swish -> conv3x3 -> + (1x1 nin_shortcut(x) | x); AttnBlock = GroupNorm -> q, k, v 1x1 -> one head over
// all C channels -> proj_out + x; Downsample = pad (0,1,0,1) + conv3x3 stride 2; Upsample = nearest 2x + conv3x3.
// Mapping onto the existing kernels:
//   * GroupNorm + swish are the per-(sample, channel) transform rows a conv applies while staging its input (fused from the
//     producer's epilogue statistics when the 32 groups are whole 4-channel blocks, else one gn_coef_kernel pass);
//   * the timestep term temb_proj(swish(temb)) is constant over the batch while sampling (one sigma per call), so it is
//     folded into conv1's bias vector by ddpm_temb_kernel: no extra pass, and the fused statistics see it;
//   * q, k, v are one packed [3C x C] GEMM whose output is the attention kernel's [3][64][T] layout (C == 64);
//   * the stride-2 conv is conv_s2_mfma_kernel (4-phase LDS tile), nearest-up is the RS_UP staging mode, the channel
//     concat of the decoder is virtual (two source pointers).
// Only what the sampler needs is built: inference, one timestep per call, x_self_cond = None (zeros: a null source).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "edm.hpp"
#include "plan.hpp"
#include "prof.hpp"

namespace mcedm {

struct DConv { int w = -1, b = -1; int cin = 0, cout = 0, taps = 0; size_t wpk = NONE, bias = NONE, wino = NONE; };   // wino: Winograd table (conv_wino.hip)
struct DNorm { int w = -1, b = -1; int C = 0; size_t gamma = NONE, beta = NONE; };
struct DRes {
  std::string key;
  int cin = 0, cout = 0;
  DNorm n1, n2;
  DConv c1, c2, sc;
  bool has_sc = false;
  int tw = -1, tb = -1;      // temb_proj parameters
  int brow = 0;              // first row of this block in the combined (conv1.bias + temb_proj) bias table
};
struct DAttn {
  std::string key;
  int C = 0;
  DNorm n;
  int qw[3] = {-1, -1, -1}, qb[3] = {-1, -1, -1};
  DConv qkv, proj;           // qkv.w / qkv.b unused: packed from the three parameters through qkv_src
  size_t qkv_src = NONE;     // scratch inside the packed buffer: [3C][C] concatenated q, k, v weights
};
struct DLevel {
  std::vector<DRes> blocks;
  std::vector<DAttn> attns;
  bool has_rs = false;
  DConv rs;                  // downsample / upsample conv
};

}  // namespace mcedm

struct mcedm_ddpm_plan {
  mcedm_ddpm_desc desc;
  std::vector<mcedm::ParamInfo> params;
  int d0w = -1, d0b = -1, d1w = -1, d1b = -1;
  mcedm::DConv conv_in, conv_out;
  mcedm::DNorm norm_out;
  std::vector<mcedm::DLevel> down, up;
  mcedm::DRes mid1, mid2;
  mcedm::DAttn mid_attn;
  mcedm::KernelVariants variants;   // per-plan kernel choices (mcedm_ddpm_plan_set_variant); -1 = process default
  int rows = 0;              // total rows of the combined bias table (sum of cout over the ResnetBlocks)
  // packed buffer (float offsets)
  size_t freqs = mcedm::NONE, w0 = mcedm::NONE, b0 = mcedm::NONE, w1 = mcedm::NONE, b1 = mcedm::NONE;
  size_t tproj_w = mcedm::NONE, tproj_b = mcedm::NONE, c1bias = mcedm::NONE;
  size_t packed_floats = 0;
  int in_total = 0;          // conv_in input channels (self-conditioning channels first)
};

namespace mcedm {

static int dadd(mcedm_ddpm_plan& P, const std::string& name, std::initializer_list<int64_t> shape) {
  ParamInfo pi;
  pi.name = name; pi.ndim = (int)shape.size(); pi.numel = 1;
  int i = 0;
  for (int64_t s : shape) { pi.shape[i++] = s; pi.numel *= s; }
  P.params.push_back(pi);
  return (int)P.params.size() - 1;
}
static DNorm dnorm(mcedm_ddpm_plan& P, const std::string& k, int C) {
  DNorm n; n.C = C;
  n.w = dadd(P, k + ".weight", {C}); n.b = dadd(P, k + ".bias", {C});
  return n;
}
static DConv dconv(mcedm_ddpm_plan& P, const std::string& k, int cin, int cout, int ks) {
  DConv c; c.cin = cin; c.cout = cout; c.taps = ks * ks;
  c.w = dadd(P, k + ".weight", {cout, cin, ks, ks}); c.b = dadd(P, k + ".bias", {cout});
  return c;
}
// registration order of ResnetBlock.__init__ (ddim_blocks.py:116-142)
static DRes dres(mcedm_ddpm_plan& P, const std::string& k, int cin, int cout) {
  const int temb = 4 * P.desc.ch;
  DRes r; r.key = k; r.cin = cin; r.cout = cout;
  r.n1 = dnorm(P, k + ".norm1", cin);
  r.c1 = dconv(P, k + ".conv1", cin, cout, 3);
  r.tw = dadd(P, k + ".temb_proj.weight", {cout, temb}); r.tb = dadd(P, k + ".temb_proj.bias", {cout});
  r.n2 = dnorm(P, k + ".norm2", cout);
  r.c2 = dconv(P, k + ".conv2", cout, cout, 3);
  if (cin != cout) { r.has_sc = true; r.sc = dconv(P, k + ".nin_shortcut", cin, cout, 1); }
  r.brow = P.rows; P.rows += cout;
  return r;
}
// AttnBlock.__init__ (ddim_blocks.py:168-192)
static DAttn dattn(mcedm_ddpm_plan& P, const std::string& k, int C) {
  DAttn a; a.key = k; a.C = C;
  a.n = dnorm(P, k + ".norm", C);
  const char* names[3] = {"q", "k", "v"};
  for (int i = 0; i < 3; ++i) {
    a.qw[i] = dadd(P, k + "." + names[i] + ".weight", {C, C, 1, 1});
    a.qb[i] = dadd(P, k + "." + names[i] + ".bias", {C});
  }
  a.qkv.cin = C; a.qkv.cout = 3 * C; a.qkv.taps = 1;
  a.proj = dconv(P, k + ".proj_out", C, C, 1);
  return a;
}
static bool dlist_has(const int32_t* v, int n, int x) {
  for (int i = 0; i < n; ++i) if (v[i] == x) return true;
  return false;
}
struct DTaker {
  size_t cur = 0;
  size_t take(size_t nfloats) { size_t o = cur; cur += align_up(nfloats, 64); return o; }
};
static void dplace(DTaker& t, DConv& c) {
  c.wpk = t.take(conv_packed_floats(c.cout, c.cin, c.taps));
  c.bias = t.take((size_t)(c.cout + 31) / 32 * 32);
  if (c.taps == 9 && c.cout % 64 == 0 && c.cin % 8 == 0) c.wino = t.take(conv_wino_packed_floats(c.cout, c.cin));
}
static void dplace(DTaker& t, DNorm& n) { n.gamma = t.take(n.C); n.beta = t.take(n.C); }
static void dplace(DTaker& t, DRes& r) {
  dplace(t, r.n1); dplace(t, r.c1); dplace(t, r.n2); dplace(t, r.c2);
  if (r.has_sc) dplace(t, r.sc);
}
static void dplace(DTaker& t, DAttn& a) {
  dplace(t, a.n); dplace(t, a.qkv); dplace(t, a.proj);
  a.qkv_src = t.take((size_t)3 * a.C * a.C);
}

}  // namespace mcedm

using namespace mcedm;

extern "C" int mcedm_ddpm_plan_create(const mcedm_ddpm_desc* d, mcedm_ddpm_plan** out) {
  MCEDM_REQUIRE(d && out, "ddpm_plan_create: null argument");
  MCEDM_REQUIRE(d->n_levels >= 1 && d->n_levels <= MCEDM_MAX_LEVELS, "ddpm_plan_create: n_levels=%d out of range", d->n_levels);
  MCEDM_REQUIRE(d->n_attn_resolutions >= 0 && d->n_attn_resolutions <= MCEDM_MAX_LEVELS, "ddpm_plan_create: bad n_attn_resolutions");
  MCEDM_REQUIRE(d->in_channels > 0 && d->out_channels > 0 && d->num_res_blocks >= 1, "ddpm_plan_create: bad channel / block counts");
  MCEDM_REQUIRE(d->ch >= 32 && d->ch % 32 == 0, "ddpm_plan_create: ch=%d must be a multiple of 32 (GroupNorm(32), ddim_blocks.py:62)", d->ch);
  MCEDM_REQUIRE(d->resolution > 0 && d->resolution % (1 << (d->n_levels - 1)) == 0, "ddpm_plan_create: resolution %d not divisible by 2^(levels-1)", d->resolution);
  for (int l = 0; l < d->n_levels; ++l) MCEDM_REQUIRE(d->ch_mult[l] >= 1, "ddpm_plan_create: ch_mult[%d] < 1", l);
  mcedm_ddpm_plan* Pp = new mcedm_ddpm_plan();
  mcedm_ddpm_plan& P = *Pp;
  P.desc = *d;
  const int ch = d->ch, temb = 4 * ch, L = d->n_levels;
  P.in_total = d->in_channels * (d->self_cond ? 2 : 1);
  // registration order of Model.__init__ (ddim_blocks.py:252-362)
  P.d0w = dadd(P, "temb.dense.0.weight", {temb, ch}); P.d0b = dadd(P, "temb.dense.0.bias", {temb});
  P.d1w = dadd(P, "temb.dense.1.weight", {temb, temb}); P.d1b = dadd(P, "temb.dense.1.bias", {temb});
  P.conv_in = dconv(P, "conv_in", P.in_total, ch, 3);
  int res = d->resolution, block_in = ch;
  auto in_mult = [&](int l) { return l == 0 ? 1 : d->ch_mult[l - 1]; };
  P.down.resize(L);
  for (int l = 0; l < L; ++l) {
    DLevel& lv = P.down[l];
    const std::string k = "down." + std::to_string(l);
    block_in = ch * in_mult(l);
    const int block_out = ch * d->ch_mult[l];
    for (int j = 0; j < d->num_res_blocks; ++j) {
      lv.blocks.push_back(dres(P, k + ".block." + std::to_string(j), block_in, block_out));
      block_in = block_out;
    }
    if (dlist_has(d->attn_resolutions, d->n_attn_resolutions, res))
      for (int j = 0; j < d->num_res_blocks; ++j) lv.attns.push_back(dattn(P, k + ".attn." + std::to_string(j), block_in));
    if (l != L - 1) { lv.has_rs = true; lv.rs = dconv(P, k + ".downsample.conv", block_in, block_in, 3); res /= 2; }
  }
  P.mid1 = dres(P, "mid.block_1", block_in, block_in);
  P.mid_attn = dattn(P, "mid.attn_1", block_in);
  P.mid2 = dres(P, "mid.block_2", block_in, block_in);
  // the up path is BUILT from the deepest level (block_in chains that way) but REGISTERED level 0 first
  // (`self.up.insert(0, up)`, ddim_blocks.py:351): build into a temporary parameter list per level, then append in order
  P.up.resize(L);
  std::vector<std::vector<ParamInfo>> up_params(L);
  std::vector<ParamInfo> saved = P.params;
  for (int l = L - 1; l >= 0; --l) {
    P.params.clear();
    DLevel& lv = P.up[l];
    const std::string k = "up." + std::to_string(l);
    const int block_out = ch * d->ch_mult[l];
    int skip_in = ch * d->ch_mult[l];
    for (int j = 0; j < d->num_res_blocks + 1; ++j) {
      if (j == d->num_res_blocks) skip_in = ch * in_mult(l);
      lv.blocks.push_back(dres(P, k + ".block." + std::to_string(j), block_in + skip_in, block_out));
      block_in = block_out;
    }
    if (dlist_has(d->attn_resolutions, d->n_attn_resolutions, res))
      for (int j = 0; j < d->num_res_blocks + 1; ++j) lv.attns.push_back(dattn(P, k + ".attn." + std::to_string(j), block_in));
    if (l != 0) { lv.has_rs = true; lv.rs = dconv(P, k + ".upsample.conv", block_in, block_in, 3); res *= 2; }
    up_params[l] = P.params;
  }
  // re-number: parameter indices inside level l are local to up_params[l]; shift them to their final position
  P.params = saved;
  auto shift_conv = [](DConv& c, int by) { if (c.w >= 0) { c.w += by; c.b += by; } };
  auto shift_norm = [](DNorm& n, int by) { n.w += by; n.b += by; };
  for (int l = 0; l < L; ++l) {
    const int by = (int)P.params.size();
    DLevel& lv = P.up[l];
    for (DRes& r : lv.blocks) {
      shift_norm(r.n1, by); shift_conv(r.c1, by); r.tw += by; r.tb += by; shift_norm(r.n2, by); shift_conv(r.c2, by);
      if (r.has_sc) shift_conv(r.sc, by);
    }
    for (DAttn& a : lv.attns) {
      shift_norm(a.n, by);
      for (int i = 0; i < 3; ++i) { a.qw[i] += by; a.qb[i] += by; }
      shift_conv(a.proj, by);
    }
    if (lv.has_rs) shift_conv(lv.rs, by);
    P.params.insert(P.params.end(), up_params[l].begin(), up_params[l].end());
  }
  P.norm_out = dnorm(P, "norm_out", block_in);
  P.conv_out = dconv(P, "conv_out", block_in, d->out_channels, 3);

  // every width that meets GroupNorm(32) must be a multiple of 32; attention is one head over all C channels and the
  // attention kernel is built for 64
  std::string bad;
  auto chk_res = [&](const DRes& r) { if (r.cin % 32 || r.cout % 32) bad = r.key; };
  auto chk_attn = [&](const DAttn& a) { if (a.C != 64) bad = a.key + " (attention over " + std::to_string(a.C) + " channels; only 64 is built)"; };
  for (auto* v : {&P.down, &P.up}) for (DLevel& lv : *v) { for (DRes& r : lv.blocks) chk_res(r); for (DAttn& a : lv.attns) chk_attn(a); }
  chk_res(P.mid1); chk_res(P.mid2); chk_attn(P.mid_attn);
  if (!bad.empty()) {
    set_error("ddpm_plan_create: unsupported block %s", bad.c_str());
    delete Pp;
    return MCEDM_ERR_UNSUPPORTED;
  }

  DTaker t;
  P.freqs = t.take(ch / 2);
  P.w0 = t.take((size_t)temb * ch); P.b0 = t.take(temb);
  P.w1 = t.take((size_t)temb * temb); P.b1 = t.take(temb);
  P.tproj_w = t.take((size_t)P.rows * temb); P.tproj_b = t.take(P.rows); P.c1bias = t.take(P.rows);
  dplace(t, P.conv_in);
  for (auto* v : {&P.down, &P.up})
    for (DLevel& lv : *v) {
      for (DRes& r : lv.blocks) dplace(t, r);
      for (DAttn& a : lv.attns) dplace(t, a);
      if (lv.has_rs) dplace(t, lv.rs);
    }
  dplace(t, P.mid1); dplace(t, P.mid_attn); dplace(t, P.mid2);
  dplace(t, P.norm_out); dplace(t, P.conv_out);
  P.packed_floats = t.cur;
  *out = Pp;
  return MCEDM_OK;
}

extern "C" int mcedm_ddpm_plan_set_variant(mcedm_ddpm_plan* plan, int which, int value) {
  MCEDM_REQUIRE(plan, "ddpm_plan_set_variant: null plan");
  MCEDM_REQUIRE(which >= 0 && which < KV_COUNT, "ddpm_plan_set_variant: unknown switch %d", which);
  MCEDM_REQUIRE(value >= -1 && value <= 1, "ddpm_plan_set_variant: value must be -1 (process default), 0 or 1");
  plan->variants.v[which] = value;
  return MCEDM_OK;
}
extern "C" void mcedm_ddpm_plan_destroy(mcedm_ddpm_plan* plan) { delete plan; }
extern "C" int mcedm_ddpm_param_count(const mcedm_ddpm_plan* plan) { return plan ? (int)plan->params.size() : MCEDM_ERR_INVALID; }
extern "C" int mcedm_ddpm_param_info(const mcedm_ddpm_plan* plan, int index, const char** name, int64_t* numel, int32_t* ndim,
                                     int64_t shape[4]) {
  MCEDM_REQUIRE(plan && index >= 0 && index < (int)plan->params.size(), "ddpm_param_info: index %d out of range", index);
  const ParamInfo& p = plan->params[index];
  if (name) *name = p.name.c_str();
  if (numel) *numel = p.numel;
  if (ndim) *ndim = p.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  return MCEDM_OK;
}
extern "C" int mcedm_ddpm_packed_bytes(const mcedm_ddpm_plan* plan, size_t* bytes) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && bytes, "ddpm_packed_bytes: null argument");
  *bytes = plan->packed_floats * sizeof(float);
  return MCEDM_OK;
}

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
namespace mcedm {

static int dcopy(float* dst, const float* src, size_t n, hipStream_t s) {
  MCEDM_HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return MCEDM_OK;
}
static int dpack(const DConv& c, const float* const* params, float* pk, hipStream_t s) {
  int rc = launch_pack_conv(params[c.w], pk + c.wpk, c.cout, c.cin, c.taps, 0, 0, s);
  if (rc) return rc;
  if (c.wino != NONE && (rc = launch_pack_conv_wino(params[c.w], pk + c.wino, c.cout, c.cin, 0, s))) return rc;
  return dcopy(pk + c.bias, params[c.b], c.cout, s);
}
static int dpack(const DNorm& n, const float* const* params, float* pk, hipStream_t s) {
  int rc = dcopy(pk + n.gamma, params[n.w], n.C, s);
  if (rc) return rc;
  return dcopy(pk + n.beta, params[n.b], n.C, s);
}
static int dpack(const mcedm_ddpm_plan& P, const DRes& r, const float* const* params, float* pk, hipStream_t s) {
  const int temb = 4 * P.desc.ch;
  int rc;
  if ((rc = dpack(r.n1, params, pk, s)) || (rc = dpack(r.c1, params, pk, s)) || (rc = dpack(r.n2, params, pk, s)) ||
      (rc = dpack(r.c2, params, pk, s))) return rc;
  if (r.has_sc && (rc = dpack(r.sc, params, pk, s))) return rc;
  if ((rc = dcopy(pk + P.tproj_w + (size_t)r.brow * temb, params[r.tw], (size_t)r.cout * temb, s))) return rc;
  if ((rc = dcopy(pk + P.tproj_b + r.brow, params[r.tb], r.cout, s))) return rc;
  return dcopy(pk + P.c1bias + r.brow, params[r.c1.b], r.cout, s);
}
static int dpack(const DAttn& a, const float* const* params, float* pk, hipStream_t s) {
  int rc;
  if ((rc = dpack(a.n, params, pk, s))) return rc;
  const size_t cc = (size_t)a.C * a.C;
  for (int i = 0; i < 3; ++i) {       // rows [q; k; v]: with one head this IS the attention kernel's (which, c) order
    if ((rc = dcopy(pk + a.qkv_src + i * cc, params[a.qw[i]], cc, s))) return rc;
    if ((rc = dcopy(pk + a.qkv.bias + (size_t)i * a.C, params[a.qb[i]], a.C, s))) return rc;
  }
  if ((rc = launch_pack_conv(pk + a.qkv_src, pk + a.qkv.wpk, 3 * a.C, a.C, 1, 0, 0, s))) return rc;
  return dpack(a.proj, params, pk, s);
}

}  // namespace mcedm

extern "C" int mcedm_ddpm_pack_weights(const mcedm_ddpm_plan* plan, const float* const* params, const float* temb_freqs,
                                       void* packed, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && params && temb_freqs && packed, "ddpm_pack_weights: null argument");
  const mcedm_ddpm_plan& P = *plan;
  for (size_t i = 0; i < P.params.size(); ++i)
    MCEDM_REQUIRE(params[i] != nullptr, "ddpm_pack_weights: parameter %zu (%s) is null", i, P.params[i].name.c_str());
  hipStream_t s = (hipStream_t)stream;
  float* pk = (float*)packed;
  const int ch = P.desc.ch, temb = 4 * ch;
  int rc;
  if ((rc = dcopy(pk + P.freqs, temb_freqs, ch / 2, s))) return rc;
  if ((rc = dcopy(pk + P.w0, params[P.d0w], (size_t)temb * ch, s)) || (rc = dcopy(pk + P.b0, params[P.d0b], temb, s)) ||
      (rc = dcopy(pk + P.w1, params[P.d1w], (size_t)temb * temb, s)) || (rc = dcopy(pk + P.b1, params[P.d1b], temb, s))) return rc;
  if ((rc = dpack(P.conv_in, params, pk, s))) return rc;
  for (auto* v : {&P.down, &P.up})
    for (const DLevel& lv : *v) {
      for (const DRes& r : lv.blocks) if ((rc = dpack(P, r, params, pk, s))) return rc;
      for (const DAttn& a : lv.attns) if ((rc = dpack(a, params, pk, s))) return rc;
      if (lv.has_rs && (rc = dpack(lv.rs, params, pk, s))) return rc;
    }
  if ((rc = dpack(P, P.mid1, params, pk, s)) || (rc = dpack(P.mid_attn, params, pk, s)) || (rc = dpack(P, P.mid2, params, pk, s))) return rc;
  if ((rc = dpack(P.norm_out, params, pk, s))) return rc;
  return dpack(P.conv_out, params, pk, s);
}

// ------------------------------------------------------------------------------------------
// timestep embedding -> the combined bias rows of every ResnetBlock's conv1
// ------------------------------------------------------------------------------------------
namespace mcedm {

__device__ __forceinline__ float swish_d(float v) { return v * (1.0f / (1.0f + expf(-v))); }     // x * sigmoid(x), ddim_blocks.py:33-35

// get_timestep_embedding (ddim_blocks.py:12-30: [sin | cos] of t * freqs) -> dense0 -> swish -> dense1 (:413-416), then
// for every block row r: out[r] = temb_proj_r . swish(temb) + temb_proj.bias[r] + conv1.bias[r] (:149-151).
// grid = nsplit workgroups of 16 waves; each recomputes the small MLP (one wave per row, 16 rows in flight: the kernel is a
// chain of three dependent mat-vec products and runs once per U-Net evaluation) and writes its slice of the rows.
constexpr int TEMB_NT = 1024, TEMB_NW = TEMB_NT / 64;
__global__ __launch_bounds__(TEMB_NT) void ddpm_temb_kernel(float t, int ch, const float* __restrict__ freqs,
                                                            const float* __restrict__ w0, const float* __restrict__ b0,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ wp, const float* __restrict__ bp,
                                                            const float* __restrict__ c1b, int rows, float* __restrict__ out) {
  extern __shared__ float sm[];
  const int temb = 4 * ch, half = ch / 2;
  float* e0 = sm;              // [ch]
  float* e1 = sm + ch;         // [temb]
  float* e2 = e1 + temb;       // [temb]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < ch; k += TEMB_NT) {
    const float arg = t * freqs[k < half ? k : k - half];
    e0[k] = (k < half) ? sinf(arg) : cosf(arg);
  }
  __syncthreads();
  for (int j = wave; j < temb; j += TEMB_NW) {
    float s = 0.f;
    for (int k = lane; k < ch; k += 64) s = fmaf(e0[k], w0[(size_t)j * ch + k], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) e1[j] = swish_d(s + b0[j]);
  }
  __syncthreads();
  for (int j = wave; j < temb; j += TEMB_NW) {
    float s = 0.f;
    for (int k = lane; k < temb; k += 64) s = fmaf(e1[k], w1[(size_t)j * temb + k], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) e2[j] = swish_d(s + b1[j]);          // every consumer applies swish to temb first
  }
  __syncthreads();
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  const int r1 = min(rows, (int)(blockIdx.x + 1) * per);
  for (int r = blockIdx.x * per + wave; r < r1; r += TEMB_NW) {
    float s = 0.f;
    for (int k = lane; k < temb; k += 64) s = fmaf(e2[k], wp[(size_t)r * temb + k], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[r] = (s + bp[r]) + c1b[r];
  }
}

// ------------------------------------------------------------------------------------------
// forward schedule, written once and run twice: `dry` (sizes only) and for real
// ------------------------------------------------------------------------------------------
struct DT { int C = 0, H = 0, W = 0; size_t off = NONE, bytes = 0, sums = NONE, sums_bytes = 0; SumTiles st; int rc = 0; };

// Channels per fused-statistics record of a C-channel tensor: GroupNorm(32) consumes it alone with C / 32 channels per
// group (2 at C = 64: pair records; multiples of 4: quad records) or inside a concat whose groups are unions of those.
// 0: no fused statistics for this width (a gn_coef_kernel pass instead).
static int record_width(int C) {
  const int cpg = C / 32;
  return (C % 32) ? 0 : (cpg % 4 == 0) ? 4 : (cpg % 2 == 0) ? 2 : 0;
}

struct DExec {
  const mcedm_ddpm_plan& P;
  bool dry;
  char* base;                 // activation region
  const float* pk;
  hipStream_t s;
  int B;
  Pool pool;
  std::vector<DT> t;

  int act(int C, int H, int W, bool with_sums) {
    DT d; d.C = C; d.H = H; d.W = W;
    d.bytes = (size_t)B * C * H * W * sizeof(float);
    d.off = pool.alloc(d.bytes);
    d.rc = with_sums ? record_width(C) : 0;
    if (d.rc) {
      d.sums_bytes = (size_t)B * conv_max_tiles(H, W) * ceil_div(C, d.rc) * 2 * sizeof(float);
      d.sums = pool.alloc(d.sums_bytes);
    }
    t.push_back(d);
    return (int)t.size() - 1;
  }
  int raw(size_t bytes) { DT d; d.bytes = bytes; d.off = pool.alloc(bytes); t.push_back(d); return (int)t.size() - 1; }
  void release(int id) {
    if (id < 0) return;
    pool.release(t[id].off, t[id].bytes);
    if (t[id].sums != NONE) pool.release(t[id].sums, t[id].sums_bytes);
  }
  float* ptr(int id) const { return id < 0 ? nullptr : reinterpret_cast<float*>(base + t[id].off); }
  float* sums(int id) const { return (id < 0 || t[id].sums == NONE) ? nullptr : reinterpret_cast<float*>(base + t[id].sums); }
};

// conv `c` reads swish(GroupNorm(cat(xa, xb))): fused from the producers' statistics records when every group is a
// whole number of records of each source (pair records on the 64-channel tensors, quads on wider ones), else one pass
// of gn_coef_kernel into a table (returned id, to be released after the conv)
static int gn_into(DExec& E, const DNorm& n, int xa, int xb, ConvArgs& c, int* table_id) {
  const DT& A = E.t[xa];
  const int Ca = A.C, Cb = xb >= 0 ? E.t[xb].C : 0;
  GnArgs g{E.ptr(xa), E.ptr(xb), Ca, Cb, A.H * A.W, E.B, 32, E.pk + n.gamma, E.pk + n.beta, nullptr, 0, 0, E.P.desc.eps,
           nullptr, nullptr, E.sums(xa), E.sums(xb), A.st, xb >= 0 ? E.t[xb].st : SumTiles{}, A.W};
  *table_id = -1;
  c.act = 1; c.coef_batch = 1;
  const int cpg = (Ca + Cb) / 32, ra = A.rc, rb = xb >= 0 ? E.t[xb].rc : A.rc;
  const bool usable = (Ca + Cb) % 32 == 0 && ra > 0 && rb > 0 && cpg % ra == 0 && cpg % rb == 0 && Ca % ra == 0 && Ca % rb == 0;
  if (usable) {
    if (!E.dry && !gn_sums_usable(g)) { set_error("ddpm forward: statistics table of a %d-channel input is unusable", Ca + Cb); return MCEDM_ERR_INVALID; }
    c.gn = g; c.gn_on = 1; c.coef = nullptr;
    return MCEDM_OK;
  }
  *table_id = E.raw((size_t)E.B * (Ca + Cb) * sizeof(Coef));
  g.coef = reinterpret_cast<Coef*>(E.ptr(*table_id));
  c.coef = g.coef;
  return E.dry ? MCEDM_OK : launch_gn_coef(g, E.s);
}

static void src_of(const DExec& E, ConvArgs& c, int xa, int xb) {
  c.xa = E.ptr(xa); c.Ca = E.t[xa].C;
  c.xb = E.ptr(xb); c.Cb = xb >= 0 ? E.t[xb].C : 0;
  c.Hs = E.t[xa].H; c.Ws = E.t[xa].W; c.H = c.Hs; c.W = c.Ws;
}
static void dst_of(DExec& E, ConvArgs& c, int out, const DConv& cv, const float* bias, bool stats) {
  c.wpk = E.pk + cv.wpk; c.bias = bias ? bias : E.pk + cv.bias;
  c.wino = cv.wino != NONE ? E.pk + cv.wino : nullptr;
  c.out = E.ptr(out); c.Cout = cv.cout; c.B = E.B;
  if (stats && E.t[out].rc) { c.gsum = E.sums(out); c.gsum_rc = E.t[out].rc; c.gsum_tiles = &E.t[out].st; }
}
// the tiling a launch WOULD use must be known in the dry run too (the consumer's usability test reads st.tiles):
// the real launch overwrites it with the same values
static int run_conv(DExec& E, ConvArgs& c, int taps, int out) {
  if (E.dry) { if (c.gsum_tiles) *c.gsum_tiles = SumTiles{1, 1, 8, 8, c.gsum_rc == 2 ? 2 : 4}; (void)out; return MCEDM_OK; }
  return launch_conv(c, taps, E.s);
}

// ResnetBlock.forward (ddim_blocks.py:144-164) on cat(xa, xb); returns the output tensor id
static int res_block(DExec& E, const DRes& r, int xa, int xb, const float* bias_table, int* out_id) {
  int rc, tab;
  const int H = E.t[xa].H, W = E.t[xa].W;
  const int h = E.act(r.cout, H, W, true);
  ConvArgs c1{};
  src_of(E, c1, xa, xb);
  if ((rc = gn_into(E, r.n1, xa, xb, c1, &tab))) return rc;
  dst_of(E, c1, h, r.c1, bias_table + r.brow, true);          // bias = conv1.bias + temb_proj(swish(temb))
  if ((rc = run_conv(E, c1, 9, h))) return rc;
  E.release(tab);
  // y = conv2(swish(norm2(h))) + (nin_shortcut(x) | x).  The 1x1 shortcut is a GEMM onto conv2's output tile, so it rides on
  // conv2 as extra K chunks at the centre tap (ConvArgs::sk_*, as the ADM decoder's skip projection does): the projected
  // tensor is neither written nor read back, one launch less per block
  const int y = E.act(r.cout, H, W, true);
  ConvArgs c2{};
  src_of(E, c2, h, -1);
  if ((rc = gn_into(E, r.n2, h, -1, c2, &tab))) return rc;
  dst_of(E, c2, y, r.c2, nullptr, true);
  int scid = -1;
  if (r.has_sc && r.c2.wino != NONE && conv_wino_shape_ok(r.cout, r.cout, H, W)) {
    // conv2 goes to the Winograd kernel, where a 1x1 projection cannot ride along: it runs as its own launch and enters
    // conv2 as the residual
    scid = E.act(r.cout, H, W, false);
    ConvArgs cs{};
    src_of(E, cs, xa, xb);
    dst_of(E, cs, scid, r.sc, nullptr, false);
    if ((rc = run_conv(E, cs, 1, scid))) return rc;
    c2.res = E.ptr(scid); c2.res_mode = RS_NONE;
  } else if (r.has_sc) {
    c2.sk_xa = E.ptr(xa); c2.sk_Ca = E.t[xa].C;
    c2.sk_xb = E.ptr(xb); c2.sk_Cb = xb >= 0 ? E.t[xb].C : 0;
    c2.sk_wpk = E.pk + r.sc.wpk; c2.sk_bias = E.pk + r.sc.bias;
  } else {
    c2.res = E.ptr(xa); c2.res_mode = RS_NONE;
  }
  if ((rc = run_conv(E, c2, 9, y))) return rc;
  E.release(tab);
  E.release(h);
  if (scid >= 0) E.release(scid);
  *out_id = y;
  return MCEDM_OK;
}

// AttnBlock.forward (ddim_blocks.py:194-219); consumes x (released here), returns the output id
static int attn_block(DExec& E, const DAttn& a, int x, int* out_id) {
  int rc, tab;
  const int H = E.t[x].H, W = E.t[x].W;
  const int qkv = E.act(3 * a.C, H, W, false);
  ConvArgs cq{};
  src_of(E, cq, x, -1);
  if ((rc = gn_into(E, a.n, x, -1, cq, &tab))) return rc;
  cq.act = 0;                                               // no nonlinearity between norm and q / k / v
  dst_of(E, cq, qkv, a.qkv, nullptr, false);
  if ((rc = run_conv(E, cq, 1, qkv))) return rc;
  E.release(tab);
  const int av = E.act(a.C, H, W, false);
  if (!E.dry && (rc = launch_attention(E.ptr(qkv), E.ptr(av), E.B, 1, H * W, E.s))) return rc;
  E.release(qkv);
  const int z = E.act(a.C, H, W, true);
  ConvArgs cp{};
  src_of(E, cp, av, -1);
  dst_of(E, cp, z, a.proj, nullptr, true);
  cp.res = E.ptr(x); cp.res_mode = RS_NONE;
  if ((rc = run_conv(E, cp, 1, z))) return rc;
  E.release(av);
  E.release(x);
  *out_id = z;
  return MCEDM_OK;
}

// header in front of the activations
struct DHeader { size_t bias, coef_in, F, total; };
static DHeader dheader(const mcedm_ddpm_plan& P, int B, int H, int W) {
  DHeader h;
  size_t cur = 0;
  auto take = [&](size_t bytes) { size_t o = cur; cur += align_up(bytes, 256); return o; };
  h.bias = take((size_t)P.rows * sizeof(float));
  h.coef_in = take((size_t)P.in_total * sizeof(Coef));
  h.F = take((size_t)B * P.desc.out_channels * H * W * sizeof(float));
  h.total = cur;
  return h;
}

// Model.forward (ddim_blocks.py:410-470) with cond None, dx None, x_self_cond None; x is scaled by the rows of coef_in
// (null = identity) while conv_in stages it.  `act` = start of the activation region.  Returns the peak bytes in *peak.
static int ddpm_forward(const mcedm_ddpm_plan& P, bool dry, const float* pk, const float* x, const Coef* coef_in, float t,
                        float* bias_table, float* out, char* act, int B, hipStream_t s, size_t* peak,
                        const float* x_self_cond = nullptr) {
  const mcedm_ddpm_desc& d = P.desc;
  const int R = d.resolution, L = d.n_levels;
  int rc;
  if (!dry) {
    int nsplit = P.rows / 32; if (nsplit < 1) nsplit = 1; if (nsplit > 64) nsplit = 64;
    const int ch = d.ch;
    hipLaunchKernelGGL(ddpm_temb_kernel, dim3(nsplit), dim3(TEMB_NT), (size_t)(ch + 8 * ch) * sizeof(float), s, t, ch, pk + P.freqs,
                       pk + P.w0, pk + P.b0, pk + P.w1, pk + P.b1, pk + P.tproj_w, pk + P.tproj_b, pk + P.c1bias, P.rows, bias_table);
    MCEDM_LAUNCH_CHECK("ddpm_temb_kernel");
  }
  DExec E{P, dry, act, pk, s, B, Pool(), {}};
  E.t.reserve(4096);        // ConvArgs::gsum_tiles points into this vector during a launch
  // conv_in on cat(x_self_cond = zeros, x): the self-conditioning half is a null source (reads as zeros)
  const int n_self = P.in_total - d.in_channels;
  std::vector<int> hs;
  {
    const int h0 = E.act(d.ch, R, R, true);
    ConvArgs ci{};
    ci.xa = x_self_cond; ci.Ca = n_self; ci.xb = x; ci.Cb = d.in_channels;       // a null self-conditioning source reads as zeros
    if (n_self == 0) { ci.xa = x; ci.Ca = d.in_channels; ci.xb = nullptr; ci.Cb = 0; }
    ci.coef = coef_in; ci.coef_batch = 0; ci.act = 0;
    ci.Hs = R; ci.Ws = R; ci.H = R; ci.W = R;
    dst_of(E, ci, h0, P.conv_in, nullptr, true);
    if ((rc = run_conv(E, ci, 9, h0))) return rc;
    hs.push_back(h0);
  }
  for (int l = 0; l < L; ++l) {
    const DLevel& lv = P.down[l];
    for (size_t j = 0; j < lv.blocks.size(); ++j) {
      int h;
      if ((rc = res_block(E, lv.blocks[j], hs.back(), -1, bias_table, &h))) return rc;
      if (!lv.attns.empty() && (rc = attn_block(E, lv.attns[j], h, &h))) return rc;
      hs.push_back(h);
    }
    if (lv.has_rs) {                                           // Downsample: stride-2 conv, no norm / activation
      const int src = hs.back();
      const int o = E.act(lv.rs.cout, E.t[src].H / 2, E.t[src].W / 2, true);
      ConvArgs cd{};
      src_of(E, cd, src, -1);
      cd.resample = RS_S2; cd.H = cd.Hs / 2; cd.W = cd.Ws / 2;
      dst_of(E, cd, o, lv.rs, nullptr, true);
      if ((rc = run_conv(E, cd, 9, o))) return rc;
      hs.push_back(o);
    }
  }
  // middle: hs[-1] stays on the stack (it is popped by the first up block)
  int h;
  if ((rc = res_block(E, P.mid1, hs.back(), -1, bias_table, &h))) return rc;
  if ((rc = attn_block(E, P.mid_attn, h, &h))) return rc;
  {
    int h2;
    if ((rc = res_block(E, P.mid2, h, -1, bias_table, &h2))) return rc;
    E.release(h);
    h = h2;
  }
  for (int l = L - 1; l >= 0; --l) {
    const DLevel& lv = P.up[l];
    for (size_t j = 0; j < lv.blocks.size(); ++j) {
      const int skip = hs.back(); hs.pop_back();
      int y;
      if ((rc = res_block(E, lv.blocks[j], h, skip, bias_table, &y))) return rc;     // cat([h, hs.pop()], dim=1)
      E.release(h); E.release(skip);
      h = y;
      if (!lv.attns.empty() && (rc = attn_block(E, lv.attns[j], h, &h))) return rc;
    }
    if (lv.has_rs) {                                           // Upsample: nearest 2x while staging, then 3x3
      const int o = E.act(lv.rs.cout, E.t[h].H * 2, E.t[h].W * 2, true);
      ConvArgs cu{};
      src_of(E, cu, h, -1);
      cu.resample = RS_UP; cu.H = cu.Hs * 2; cu.W = cu.Ws * 2;
      dst_of(E, cu, o, lv.rs, nullptr, true);
      if ((rc = run_conv(E, cu, 9, o))) return rc;
      E.release(h);
      h = o;
    }
  }
  MCEDM_REQUIRE(hs.empty(), "ddpm forward: %zu skip tensors left over", hs.size());
  // out = conv_out(swish(norm_out(h)))
  {
    int tab;
    ConvArgs co{};
    src_of(E, co, h, -1);
    if ((rc = gn_into(E, P.norm_out, h, -1, co, &tab))) return rc;
    co.wpk = pk + P.conv_out.wpk; co.bias = pk + P.conv_out.bias;
    co.out = out; co.Cout = d.out_channels; co.B = B;
    if (!dry && (rc = launch_conv(co, 9, s))) return rc;
    E.release(tab);
    E.release(h);
  }
  if (peak) *peak = E.pool.peak;
  return MCEDM_OK;
}

static int ddpm_sizes(const mcedm_ddpm_plan& P, int B, DHeader* hd, size_t* act_bytes) {
  MCEDM_REQUIRE(B > 0, "ddpm: empty batch");
  *hd = dheader(P, B, P.desc.resolution, P.desc.resolution);
  return ddpm_forward(P, true, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, B, nullptr, act_bytes);
}

// get_denoised (ddim.py:915-947): D = x + (-sigma) F(c_in x, c_noise), sigma and c_noise host scalars
static int ddpm_denoise_impl(const mcedm_ddpm_plan& P, const DHeader& hd, const float* pk, const float* x, float sigma,
                             float c_noise, float* D_out, float* F_out, void* ws, int B, hipStream_t s) {
  int rc;
  const float c_in = 1.0f / sqrtf(sigma * sigma + 1.0f);                 // 1 / (sigma ** 2 + 1).sqrt(), fp32
  Coef* coef_in = at<Coef>(ws, hd.coef_in);
  const int n_self = P.in_total - P.desc.in_channels;
  if ((rc = launch_vp_coef(c_in, n_self, P.desc.in_channels, coef_in, s))) return rc;
  float* F = F_out ? F_out : at<float>(ws, hd.F);
  if ((rc = ddpm_forward(P, false, pk, x, coef_in, c_noise, at<float>(ws, hd.bias), F, at<char>(ws, hd.total), B, s, nullptr))) return rc;
  const size_t total = (size_t)B * P.desc.out_channels * P.desc.resolution * P.desc.resolution;
  return launch_vp_finish(x, F, sigma, total, D_out, s);
}

// round_sigma (ddim.py:949-957): nearest entry of edm_steps in fp32; ties resolve to the lower index like argmin
static int nearest_step(const float* steps, int n, float sigma) {
  int best = 0;
  float bd = fabsf(sigma - steps[0]);
  for (int j = 1; j < n; ++j) {
    const float dd = fabsf(sigma - steps[j]);
    if (dd < bd) { bd = dd; best = j; }
  }
  return best;
}

}  // namespace mcedm

extern "C" int mcedm_ddpm_workspace_bytes(const mcedm_ddpm_plan* plan, int B, size_t* bytes) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && bytes, "ddpm_workspace_bytes: null argument");
  DHeader hd; size_t act = 0;
  int rc = ddpm_sizes(*plan, B, &hd, &act);
  if (rc) return rc;
  *bytes = hd.total + act;
  return MCEDM_OK;
}

extern "C" int mcedm_ddpm_forward(const mcedm_ddpm_plan* plan, const void* packed, const float* x, float t, float* out,
                                  void* workspace, size_t workspace_bytes, int B, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && packed && x && out && workspace, "ddpm_forward: null argument");
  DHeader hd; size_t act = 0;
  int rc = ddpm_sizes(*plan, B, &hd, &act);
  if (rc) return rc;
  if (hd.total + act > workspace_bytes) { set_error("ddpm_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, hd.total + act); return MCEDM_ERR_WORKSPACE; }
  return ddpm_forward(*plan, false, (const float*)packed, x, nullptr, t, at<float>(workspace, hd.bias), out,
                      at<char>(workspace, hd.total), B, (hipStream_t)stream, nullptr);
}

extern "C" int mcedm_ddpm_forward_sc(const mcedm_ddpm_plan* plan, const void* packed, const float* x, const float* x_self_cond,
                                     float t, float* out, void* workspace, size_t workspace_bytes, int B, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && packed && x && out && workspace, "ddpm_forward_sc: null argument");
  MCEDM_REQUIRE(!x_self_cond || plan->desc.self_cond, "ddpm_forward_sc: the network was built without self-conditioning channels");
  DHeader hd; size_t act = 0;
  int rc = ddpm_sizes(*plan, B, &hd, &act);
  if (rc) return rc;
  if (hd.total + act > workspace_bytes) { set_error("ddpm_forward_sc: workspace too small (%zu < %zu bytes)", workspace_bytes, hd.total + act); return MCEDM_ERR_WORKSPACE; }
  return ddpm_forward(*plan, false, (const float*)packed, x, nullptr, t, at<float>(workspace, hd.bias), out,
                      at<char>(workspace, hd.total), B, (hipStream_t)stream, nullptr, x_self_cond);
}

extern "C" int mcedm_ddpm_denoise(const mcedm_ddpm_plan* plan, const void* packed, const float* x, float sigma, float c_noise,
                                  float* D_out, float* F_out, void* workspace, size_t workspace_bytes, int B, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && packed && x && D_out && workspace, "ddpm_denoise: null argument");
  DHeader hd; size_t act = 0;
  int rc = ddpm_sizes(*plan, B, &hd, &act);
  if (rc) return rc;
  if (hd.total + act > workspace_bytes) { set_error("ddpm_denoise: workspace too small (%zu < %zu bytes)", workspace_bytes, hd.total + act); return MCEDM_ERR_WORKSPACE; }
  return ddpm_denoise_impl(*plan, hd, (const float*)packed, x, sigma, c_noise, D_out, F_out, workspace, B, (hipStream_t)stream);
}

namespace mcedm {
struct RBufs { size_t x, xn, d, x32, D, mask, total; };
static RBufs rbufs(const mcedm_ddpm_plan& P, int B) {
  RBufs r; size_t cur = 0;
  auto take = [&](size_t bytes) { size_t o = cur; cur += align_up(bytes, 256); return o; };
  const size_t n = (size_t)B * P.desc.in_channels * P.desc.resolution * P.desc.resolution;
  r.x = take(n * 8); r.xn = take(n * 8); r.d = take(n * 8); r.x32 = take(n * 4); r.D = take(n * 4); r.mask = take(n * 4);
  r.total = cur;
  return r;
}
// hu_mask: 1 = known: rows [0, n_time_h) of the h channels and [0, n_time_u) of the u channels (ddim.py:970-972)
__global__ void repaint_mask_kernel(float* __restrict__ m, int C, int H, int W, int h_ch, int u_ch, int n_time_h, int n_time_u,
                                    size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)((i / W) % H), c = (int)((i / ((size_t)W * H)) % C);
    float v = 1.0f;
    if (c < h_ch && row >= n_time_h) v = 0.0f;
    if (c >= h_ch && c < h_ch + u_ch && row >= n_time_u) v = 0.0f;
    m[i] = v;
  }
}
}  // namespace mcedm

extern "C" int mcedm_repaint_workspace_bytes(const mcedm_ddpm_plan* plan, int B, size_t* bytes) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && bytes, "repaint_workspace_bytes: null argument");
  size_t u = 0;
  int rc = mcedm_ddpm_workspace_bytes(plan, B, &u);
  if (rc) return rc;
  *bytes = rbufs(*plan, B).total + u;
  return MCEDM_OK;
}

extern "C" int mcedm_repaint_schedule(const mcedm_repaint_desc* sp, double* t_steps) {
  MCEDM_REQUIRE(sp && t_steps && sp->edm_steps && sp->num_diffusion_timesteps >= 2, "repaint_schedule: bad argument");
  MCEDM_REQUIRE(sp->timesteps >= 2, "repaint_schedule: timesteps=%d (the reference divides by timesteps-1)", sp->timesteps);
  const int n = sp->num_diffusion_timesteps, N = sp->timesteps;
  const double smin = std::max(sp->sigma_min, (double)sp->edm_steps[n - 1]);      // ddim.py:977-978 with :128-129
  const double smax = std::min(sp->sigma_max, (double)sp->edm_steps[0]);
  const double a = std::pow(smax, 1.0 / sp->rho), b = std::pow(smin, 1.0 / sp->rho) - std::pow(smax, 1.0 / sp->rho);
  for (int i = 0; i < N; ++i) {
    const double ts = std::pow(a + (double)i / (double)(N - 1) * b, sp->rho);
    t_steps[i] = (double)sp->edm_steps[nearest_step(sp->edm_steps, n, (float)ts)];      // round_sigma, ddim.py:986
  }
  t_steps[N] = 0.0;
  return MCEDM_OK;
}

namespace mcedm {
static int repaint_impl(const mcedm_ddpm_plan* plan, const void* packed, const mcedm_repaint_desc* sp, const float* hu,
                        const float* init_noise, const double* step_noise, const double* repeat_noise,
                        const unsigned long long* rng_seed, double* out, int return_last, void* workspace,
                        size_t workspace_bytes, int B, void* stream);
}
extern "C" int mcedm_repaint_sample(const mcedm_ddpm_plan* plan, const void* packed, const mcedm_repaint_desc* sp,
                                    const float* hu, const float* init_noise, const double* step_noise,
                                    const double* repeat_noise, double* out, int return_last, void* workspace,
                                    size_t workspace_bytes, int B, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  return repaint_impl(plan, packed, sp, hu, init_noise, step_noise, repeat_noise, nullptr, out, return_last, workspace,
                      workspace_bytes, B, stream);
}
extern "C" int mcedm_repaint_sample_rng(const mcedm_ddpm_plan* plan, const void* packed, const mcedm_repaint_desc* sp,
                                        const float* hu, const float* init_noise, const uint64_t* rng_seed, double* out,
                                        int return_last, void* workspace, size_t workspace_bytes, int B, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(rng_seed != nullptr, "repaint_sample_rng: rng_seed is null");
  return repaint_impl(plan, packed, sp, hu, init_noise, nullptr, nullptr, reinterpret_cast<const unsigned long long*>(rng_seed),
                      out, return_last, workspace, workspace_bytes, B, stream);
}
extern "C" int mcedm_normal_fill(double* out, size_t n, const uint64_t* rng_seed, uint64_t draw, void* stream) {
  MCEDM_REQUIRE(out && rng_seed, "normal_fill: null argument");
  if (n == 0) return MCEDM_OK;
  return launch_normal_fill(out, reinterpret_cast<const unsigned long long*>(rng_seed), draw, n, (hipStream_t)stream);
}

static int mcedm::repaint_impl(const mcedm_ddpm_plan* plan, const void* packed, const mcedm_repaint_desc* sp, const float* hu,
                               const float* init_noise, const double* step_noise, const double* repeat_noise,
                               const unsigned long long* rng_seed, double* out, int return_last, void* workspace,
                               size_t workspace_bytes, int B, void* stream) {
  MCEDM_REQUIRE(plan && packed && sp && hu && init_noise && out && workspace, "repaint_sample: null argument");
  const mcedm_ddpm_plan& P = *plan;
  MCEDM_REQUIRE(P.desc.in_channels == P.desc.out_channels, "repaint_sample: in_channels != out_channels");
  MCEDM_REQUIRE(sp->edm_steps && sp->alphas_cumprod_ext && sp->num_diffusion_timesteps >= 2, "repaint_sample: missing schedule tables");
  MCEDM_REQUIRE(sp->timesteps >= 2 && sp->timesteps <= 4096 && sp->n_repeat >= 1, "repaint_sample: timesteps=%d n_repeat=%d out of range", sp->timesteps, sp->n_repeat);
  MCEDM_REQUIRE(sp->h_ch >= 0 && sp->u_ch >= 0 && sp->h_ch + sp->u_ch <= P.desc.in_channels, "repaint_sample: h_ch + u_ch exceeds the state channels");
  MCEDM_REQUIRE(std::fabs(sp->w) < 0.001, "repaint_sample: classifier-free guidance needs a conditional network (cond is None on this path, ddim.py:935)");
  MCEDM_REQUIRE(sp->n_repeat == 1 || repeat_noise != nullptr || rng_seed != nullptr, "repaint_sample: n_repeat > 1 needs repeat_noise");
  const int N = sp->timesteps, R = sp->n_repeat, n = sp->num_diffusion_timesteps;
  const int S = P.desc.resolution, C = P.desc.in_channels;
  std::vector<double> t(N + 1);
  int rc = mcedm_repaint_schedule(sp, t.data());
  if (rc) return rc;
  DHeader hd; size_t act = 0;
  if ((rc = ddpm_sizes(P, B, &hd, &act))) return rc;
  const RBufs rb = rbufs(P, B);
  if (rb.total + hd.total + act > workspace_bytes) {
    set_error("repaint_sample: workspace too small (%zu < %zu bytes)", workspace_bytes, rb.total + hd.total + act);
    return MCEDM_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const float* pk = (const float*)packed;
  double* x = at<double>(workspace, rb.x);
  double* xn = at<double>(workspace, rb.xn);
  double* dcur = at<double>(workspace, rb.d);
  float* x32 = at<float>(workspace, rb.x32);
  float* D = at<float>(workspace, rb.D);
  float* mask = at<float>(workspace, rb.mask);
  void* uws = at<char>(workspace, rb.total);
  const size_t hw = (size_t)S * S, total = (size_t)B * C * hw;
  const int Tout = return_last ? 1 : N + 1;

  auto alpha = [&](double tt) -> float {          // compute_alpha(t.long()) (ddim.py:700-704): the SIGMA is the index
    long idx = (long)tt + 1;
    if (idx < 0) idx = 0;
    if (idx > n) idx = n;
    return sp->alphas_cumprod_ext[idx];
  };
  auto round_sigma = [&](double sg) -> double { return (double)sp->edm_steps[nearest_step(sp->edm_steps, n, (float)sg)]; };
  auto denoise = [&](double sg) -> int {          // get_denoised at a scalar sigma: c_noise = n - 1 - index(sigma)
    const float s32 = (float)sg;
    const float cn = (float)(n - 1 - nearest_step(sp->edm_steps, n, s32));
    return ddpm_denoise_impl(P, hd, pk, x32, s32, cn, D, nullptr, uws, B, s);
  };

  hipLaunchKernelGGL(repaint_mask_kernel, dim3(2048 < (total + 255) / 256 ? 2048 : (unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     mask, C, S, S, sp->h_ch, sp->u_ch, sp->n_time_h, sp->n_time_u, total);
  MCEDM_LAUNCH_CHECK("repaint_mask_kernel");
  {
    const float aT = alpha(t[0]);
    if ((rc = launch_repaint_init(hu, init_noise, mask, sqrtf(aT), sqrtf(1.0f - aT), t[0], total, x, x32, s))) return rc;
  }
  if (!return_last && (rc = launch_heun_store(x, C, hw, 0, Tout, total, out, s))) return rc;
  for (int i = 0; i < N; ++i) {
    const double t_cur = t[i], t_next = t[i + 1];
    const bool in_range = sp->S_min <= t_cur && t_cur <= sp->S_max;
    const double gamma = in_range ? std::min(sp->S_churn / N, std::sqrt(2.0) - 1.0) : 0.0;
    double t_hat = round_sigma(t_cur + gamma * t_cur);
    {
      const double c = std::sqrt(t_hat * t_hat - t_cur * t_cur) * sp->S_noise;          // ddim.py:1004
      if (c != 0.0) {
        if (rng_seed) {
          if ((rc = launch_heun_churn_rng(x, rng_seed, (unsigned long long)i * R, c, total, x32, s))) return rc;
        } else {
          MCEDM_REQUIRE(step_noise != nullptr, "repaint_sample: step %d adds noise (t_hat > t_cur) but step_noise is NULL", i);
          if ((rc = launch_heun_churn(x, step_noise + (size_t)i * total, nullptr, c, total, x32, s))) return rc;
        }
      }
    }
    for (int k = 0; k < R; ++k) {
      // Euler step (ddim.py:1008-1015); x holds x_hat
      if ((rc = denoise(t_hat))) return rc;
      if ((rc = launch_heun_euler(x, D, nullptr, t_hat, t_next - t_hat, total, dcur, xn, x32, s))) return rc;
      if (i < N - 1) {                                                                   // 2nd order correction (:1018-1026)
        if ((rc = denoise(t_next))) return rc;
        if ((rc = launch_heun_correct(x, dcur, D, nullptr, t_next, t_next - t_hat, total, xn, x32, s))) return rc;
      }
      std::swap(x, xn);                                                                  // x = x_next
      // replace the known part with the data noised to t_next (:1028-1031)
      const float at = alpha(t_next);
      if ((rc = launch_repaint_known(x, hu, init_noise, mask, sqrtf(at), sqrtf(1.0f - at), 0, total, x32, s))) return rc;
      if (k < R - 1) {                                                                   // back up from t_next to a new t_hat (:1033-1037)
        t_hat = round_sigma(t_next + (std::sqrt(2.0) - 1.0) * t_next);
        const double c = std::sqrt(t_hat * t_hat - t_next * t_next) * sp->S_noise;
        if (rng_seed) rc = launch_heun_churn_rng(x, rng_seed, (unsigned long long)i * R + 1 + k, c, total, x32, s);
        else rc = launch_heun_churn(x, repeat_noise + ((size_t)i * (R - 1) + k) * total, nullptr, c, total, x32, s);
        if (rc) return rc;
      }
    }
    if (i == N - 1 && (rc = launch_repaint_known(x, hu, init_noise, mask, 0.f, 0.f, 1, total, x32, s))) return rc;   // :1041-1043
    if (!return_last && (rc = launch_heun_store(x, C, hw, i + 1, Tout, total, out, s))) return rc;
  }
  if (return_last && (rc = launch_heun_store(x, C, hw, 0, 1, total, out, s))) return rc;
  return MCEDM_OK;
}


// ------------------------------------------------------------------------------------------
// PlDdim.sample_with_repeat (models/ddim.py:808-913): DDIM steps with RePaint inner loops, fp32 throughout
// ------------------------------------------------------------------------------------------
namespace mcedm {
struct DdimBufs { size_t xt, x0, et, mask, total; };
static DdimBufs ddim_bufs(const mcedm_ddpm_plan& P, int B) {
  DdimBufs r; size_t cur = 0;
  auto take = [&](size_t bytes) { size_t o = cur; cur += align_up(bytes, 256); return o; };
  const size_t n = (size_t)B * P.desc.in_channels * P.desc.resolution * P.desc.resolution;
  r.xt = take(n * 4); r.x0 = take(n * 4); r.et = take(n * 4); r.mask = take(n * 4);
  r.total = cur;
  return r;
}
}  // namespace mcedm

extern "C" int mcedm_ddim_workspace_bytes(const mcedm_ddpm_plan* plan, int B, size_t* bytes) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && bytes, "ddim_workspace_bytes: null argument");
  size_t u = 0;
  int rc = mcedm_ddpm_workspace_bytes(plan, B, &u);
  if (rc) return rc;
  *bytes = ddim_bufs(*plan, B).total + u;
  return MCEDM_OK;
}

// The DDIM timestep sequence of PlDdim.sample_with_repeat (models/ddim.py:823-830).
//  uniform: range(0, n, n // N) -- may hold more than N entries;
//  quad:    [int(s) for s in np.linspace(0, sqrt(0.8 n), N) ** 2].  numpy.linspace evaluates arange(N) * step with
//           step = stop / (N - 1) and then PINS the last sample to `stop` itself; hi * i / (N - 1) is a different rounding
//           and lands on the other side of an integer for 85 of ~1200 (n, N) pairs (n = 1000, N = 100: first sampled step
//           799 instead of 800), so the two operations are reproduced in numpy's order.
static std::vector<int> ddim_timestep_seq(int n, int N, int skip_type) {
  std::vector<int> seq;
  if (skip_type == 0) {
    const int skip = n / N;
    for (int v = 0; v < n; v += skip) seq.push_back(v);
  } else {
    const double hi = std::sqrt(n * 0.8);
    const double step = N > 1 ? hi / (double)(N - 1) : 0.0;
    for (int i = 0; i < N; ++i) {
      const double v = (N > 1 && i == N - 1) ? hi : (double)i * step;
      seq.push_back((int)(v * v));
    }
  }
  return seq;
}

extern "C" int mcedm_ddim_timesteps(int num_diffusion_timesteps, int timesteps, int skip_type, int* seq, int capacity, int* count) {
  MCEDM_REQUIRE(count && num_diffusion_timesteps >= 2 && timesteps >= 1 && timesteps <= num_diffusion_timesteps,
                "ddim_timesteps: bad schedule (timesteps=%d of %d)", timesteps, num_diffusion_timesteps);
  MCEDM_REQUIRE(skip_type == 0 || skip_type == 1, "ddim_timesteps: skip_type must be 0 (uniform) or 1 (quad)");
  const std::vector<int> s = ddim_timestep_seq(num_diffusion_timesteps, timesteps, skip_type);
  *count = (int)s.size();
  if (seq) {
    MCEDM_REQUIRE(capacity >= (int)s.size(), "ddim_timesteps: capacity %d < %d entries", capacity, (int)s.size());
    for (size_t i = 0; i < s.size(); ++i) seq[i] = s[i];
  }
  return MCEDM_OK;
}

extern "C" int mcedm_ddim_repaint_sample(const mcedm_ddpm_plan* plan, const void* packed, const mcedm_ddim_desc* sp,
                                         const float* hu, const float* init_noise, const float* eta_noise, float* xs_out,
                                         float* x0_out, int return_last, void* workspace, size_t workspace_bytes, int B,
                                         void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && packed && sp && hu && init_noise && xs_out && workspace, "ddim_repaint_sample: null argument");
  const mcedm_ddpm_plan& P = *plan;
  MCEDM_REQUIRE(P.desc.in_channels == P.desc.out_channels, "ddim_repaint_sample: in_channels != out_channels");
  const int n = sp->num_diffusion_timesteps, N = sp->timesteps, R = sp->n_repeat;
  MCEDM_REQUIRE(sp->alphas_cumprod_ext && n >= 2 && N >= 1 && N <= n && R >= 1, "ddim_repaint_sample: bad schedule (timesteps=%d of %d, n_repeat=%d)", N, n, R);
  MCEDM_REQUIRE(sp->skip_type == 0 || sp->skip_type == 1, "ddim_repaint_sample: skip_type must be 0 (uniform) or 1 (quad)");
  MCEDM_REQUIRE(sp->h_ch >= 0 && sp->u_ch >= 0 && sp->h_ch + sp->u_ch <= P.desc.in_channels, "ddim_repaint_sample: h_ch + u_ch exceeds the state channels");
  MCEDM_REQUIRE(!sp->self_cond || P.desc.self_cond, "ddim_repaint_sample: self-conditioning asked of a network built without it");
  const bool stochastic = std::fabs(sp->eta) > 1e-10;                   // ddim.py:884
  MCEDM_REQUIRE(!stochastic || eta_noise != nullptr, "ddim_repaint_sample: eta != 0 needs eta_noise");
  // the timestep sequence (ddim.py:823-830) and its predecessor list (:845)
  std::vector<int> seq = ddim_timestep_seq(n, N, sp->skip_type);
  const int S = (int)seq.size();
  DHeader hd; size_t act = 0;
  int rc;
  if ((rc = ddpm_sizes(P, B, &hd, &act))) return rc;
  const DdimBufs db = ddim_bufs(P, B);
  if (db.total + hd.total + act > workspace_bytes) {
    set_error("ddim_repaint_sample: workspace too small (%zu < %zu bytes)", workspace_bytes, db.total + hd.total + act);
    return MCEDM_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const float* pk = (const float*)packed;
  float* xt = at<float>(workspace, db.xt);
  float* x0 = at<float>(workspace, db.x0);
  float* et = at<float>(workspace, db.et);
  float* mask = at<float>(workspace, db.mask);
  void* uws = at<char>(workspace, db.total);
  const int res = P.desc.resolution, C = P.desc.in_channels;
  const size_t hw = (size_t)res * res, total = (size_t)B * C * hw;
  const int Txs = return_last ? 1 : S + 1, Tx0 = return_last ? 1 : S;
  auto alpha = [&](int t) -> float { return sp->alphas_cumprod_ext[t + 1]; };      // compute_alpha(t): index t + 1 (ddim.py:700-704)

  hipLaunchKernelGGL(repaint_mask_kernel, dim3(2048 < (total + 255) / 256 ? 2048 : (unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     mask, C, res, res, sp->h_ch, sp->u_ch, sp->n_time_h, sp->n_time_u, total);
  MCEDM_LAUNCH_CHECK("repaint_mask_kernel");
  {   // x = (hu * a[-1].sqrt() + hu_noise * (1 - a[-1]).sqrt()) * mask + hu_noise * (1 - mask)   (:837-838); a[-1] = alpha(n - 1)
    const float aT = alpha(n - 1);
    if ((rc = launch_ddim_init(hu, init_noise, mask, sqrtf(aT), sqrtf(1.0f - aT), total, xt, s))) return rc;
  }
  if (!return_last && (rc = launch_store_f32(xt, C, hw, 0, Txs, total, xs_out, s))) return rc;
  bool have_x0 = false;
  for (int step = 0; step < S; ++step) {
    const int i = seq[S - 1 - step], j = (S - 1 - step) > 0 ? seq[S - 2 - step] : -1;
    const float a_t = alpha(i), at_next = alpha(j);
    const float s0 = sqrtf(a_t), s1 = sqrtf(1.0f - a_t);
    for (int k = 0; k < R; ++k) {
      const float* sc = (sp->self_cond && have_x0) ? x0 : nullptr;      // x_self_cond = x0_t if self_condition else None (:863)
      if ((rc = ddpm_forward(P, false, pk, xt, nullptr, (float)i, at<float>(uws, hd.bias), et, at<char>(uws, hd.total), B, s, nullptr, sc))) return rc;
      if ((rc = launch_ddim_x0(xt, et, hu, mask, s0, s1, k < R - 1 ? 1 : 0, total, x0, s))) return rc;
      have_x0 = true;
    }
    float c1 = 0.f, c2;
    if (stochastic) {        // c1 = eta * sqrt((1 - at / at_next) * (1 - at_next) / (1 - at)); c2 = sqrt((1 - at_next) - c1^2), fp32 like the tensors
      c1 = (float)sp->eta * sqrtf((1.0f - a_t / at_next) * (1.0f - at_next) / (1.0f - a_t));
      c2 = sqrtf((1.0f - at_next) - c1 * c1);
    } else {
      c2 = sqrtf(1.0f - at_next);
    }
    if ((rc = launch_ddim_next(x0, et, hu, init_noise, mask, stochastic ? eta_noise + (size_t)step * total : nullptr, sqrtf(at_next), c1,
                               c2, total, xt, s))) return rc;
    if (!return_last) {
      if ((rc = launch_store_f32(xt, C, hw, step + 1, Txs, total, xs_out, s))) return rc;
      if (x0_out && (rc = launch_store_f32(x0, C, hw, step, Tx0, total, x0_out, s))) return rc;
    }
  }
  if (return_last) {
    if ((rc = launch_store_f32(xt, C, hw, 0, 1, total, xs_out, s))) return rc;
    if (x0_out && (rc = launch_store_f32(x0, C, hw, 0, 1, total, x0_out, s))) return rc;
  }
  return MCEDM_OK;
}
