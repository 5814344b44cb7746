// plan.hip -- the C ABI: plan construction, weight packing, workspace layout, the U-Net forward
// schedule, EDM denoise and the Heun sampler loop.  Host code only launches the kernels of
// conv_mfma.hip / norm_emb_attn.hip / edm.hip on the caller's stream; it never allocates device
// memory and never synchronises.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "edm.hpp"
#include "pack.hpp"
#include "plan.hpp"
#include "prof.hpp"
#include "bwd.hpp"

namespace mcedm {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------------------------------
// plan construction: restates DhariwalUNet.__init__ (models/adm_blocks.py:203-317)
// ------------------------------------------------------------------------------------------
static int add_param(mcedm_plan& P, const std::string& name, std::initializer_list<int64_t> shape) {
  ParamInfo pi;
  pi.name = name;
  pi.ndim = (int)shape.size();
  pi.numel = 1;
  int i = 0;
  for (int64_t s : shape) { pi.shape[i++] = s; pi.numel *= s; }
  P.params.push_back(pi);
  return (int)P.params.size() - 1;
}

static NormP make_norm(mcedm_plan& P, const std::string& key, int C) {
  NormP n;
  n.C = C;
  n.groups = std::min(32, C / 4);          // adm_blocks.py:89
  n.w = add_param(P, key + ".weight", {C});
  n.b = add_param(P, key + ".bias", {C});
  return n;
}

static ConvP make_conv(mcedm_plan& P, const std::string& key, int cin, int cout, int k, int qkv_heads = 0) {
  ConvP c;
  c.cin = cin; c.cout = cout; c.taps = k * k; c.qkv_heads = qkv_heads;
  c.w = add_param(P, key + ".weight", {cout, cin, k, k});
  c.b = add_param(P, key + ".bias", {cout});
  return c;
}

static BlockP make_block(mcedm_plan& P, const std::string& key, int cin, int cout, bool up, bool down, bool attn) {
  const int emb = P.desc.ch;
  BlockP b;
  b.key = key; b.cin = cin; b.cout = cout; b.up = up; b.down = down;
  b.heads = attn ? cout / P.desc.channels_per_head : 0;      // adm_blocks.py:135
  b.attn = b.heads > 0;
  b.norm0 = make_norm(P, key + ".norm0", cin);
  b.conv0 = make_conv(P, key + ".conv0", cin, cout, 3);
  b.aff_w = add_param(P, key + ".affine.weight", {2 * cout, emb});
  b.aff_b = add_param(P, key + ".affine.bias", {2 * cout});
  b.norm1 = make_norm(P, key + ".norm1", cout);
  b.conv1 = make_conv(P, key + ".conv1", cout, cout, 3);
  b.skip_kernel = -1;
  if (cout != cin || up || down) {                           // adm_blocks.py:148-151
    b.skip_kernel = (cout != cin) ? 1 : 0;
    if (b.skip_kernel == 1) b.skip = make_conv(P, key + ".skip", cin, cout, 1);
  }
  if (b.attn) {
    b.norm2 = make_norm(P, key + ".norm2", cout);
    b.qkv = make_conv(P, key + ".qkv", cout, 3 * cout, 1, b.heads);
    b.proj = make_conv(P, key + ".proj", cout, cout, 1);
  }
  return b;
}

static bool in_list(const int32_t* v, int n, int x) {
  for (int i = 0; i < n; ++i) if (v[i] == x) return true;
  return false;
}

struct Taker {
  size_t cur = 0;
  size_t take(size_t nfloats) { size_t o = cur; cur += align_up(nfloats, 64); return o; }
};

static void place_conv(Taker& t, ConvP& c, bool dgrad) {
  c.wpk = t.take(conv_packed_floats(c.cout, c.cin, c.taps));
  c.bias = t.take((size_t)(c.cout + 31) / 32 * 32);
  if (dgrad) c.wpk_dgrad = t.take(conv_packed_floats(c.cin, c.cout, c.taps));
  if (c.taps == 9 && c.qkv_heads == 0) {       // Winograd tables where the kernel's channel constraints can be met
    if (c.cout % 64 == 0 && c.cin % 8 == 0) c.wino = t.take(conv_wino_packed_floats(c.cout, c.cin));
    if (dgrad && c.cin % 64 == 0 && c.cout % 8 == 0) c.wino_dgrad = t.take(conv_wino_packed_floats(c.cin, c.cout));
  }
}
static void place_norm(Taker& t, NormP& n) { n.gamma = t.take(n.C); n.beta = t.take(n.C); }

}  // namespace mcedm

using namespace mcedm;

extern "C" int mcedm_version(void) { return MCEDM_ABI_VERSION; }

extern "C" int mcedm_unet_plan_set_variant(mcedm_plan* plan, int which, int value) {
  MCEDM_REQUIRE(plan, "plan_set_variant: null plan");
  MCEDM_REQUIRE(which >= 0 && which < KV_COUNT, "plan_set_variant: unknown switch %d", which);
  MCEDM_REQUIRE(value >= -1 && value <= 1, "plan_set_variant: value must be -1 (process default), 0 or 1");
  plan->variants.v[which] = value;
  return MCEDM_OK;
}
extern "C" const char* mcedm_last_error(void) { return g_err; }

extern "C" int mcedm_unet_plan_create(const mcedm_unet_desc* d, mcedm_plan** out) {
  MCEDM_REQUIRE(d && out, "plan_create: null argument");
  MCEDM_REQUIRE(d->n_levels >= 1 && d->n_levels <= MCEDM_MAX_LEVELS, "plan_create: n_levels=%d out of range", d->n_levels);
  MCEDM_REQUIRE(d->n_attn_resolutions >= 0 && d->n_attn_resolutions <= MCEDM_MAX_LEVELS, "plan_create: bad n_attn_resolutions");
  MCEDM_REQUIRE(d->in_channels > 0 && d->out_channels > 0 && d->cond_channels >= 0, "plan_create: bad channel counts");
  MCEDM_REQUIRE(d->ch > 0 && d->ch % 8 == 0, "plan_create: ch=%d must be a positive multiple of 8", d->ch);
  MCEDM_REQUIRE(d->num_res_blocks >= 1, "plan_create: num_res_blocks must be >= 1");
  if (d->channels_per_head != 64) {
    set_error("plan_create: channels_per_head=%d unsupported (the hot path uses 64, adm_blocks.py:223)", d->channels_per_head);
    return MCEDM_ERR_UNSUPPORTED;
  }
  for (int l = 0; l < d->n_levels; ++l) {
    const int c = d->ch * d->ch_mult[l];
    MCEDM_REQUIRE(d->ch_mult[l] >= 1 && c % 8 == 0, "plan_create: level %d width %d must be a multiple of 8", l, c);
  }
  MCEDM_REQUIRE(d->dx_mode == MCEDM_DX_NONE || d->dx_mode == MCEDM_DX_CAT || d->dx_mode == MCEDM_DX_ENC, "plan_create: dx_mode=%d", d->dx_mode);
  MCEDM_REQUIRE((d->dx_mode == MCEDM_DX_NONE) == (d->dx_channels == 0) && d->dx_channels >= 0,
                "plan_create: dx_channels=%d does not go with dx_mode=%d", d->dx_channels, d->dx_mode);
  mcedm_plan* Pp = new mcedm_plan();
  mcedm_plan& P = *Pp;
  P.desc = *d;
  P.levels_div = 1 << (d->n_levels - 1);
  const int ch = d->ch;
  auto key = [&](const char* side, int res, const std::string& tail) {
    return std::string(side) + "." + std::to_string(res) + "x" + std::to_string(res) + "_" + tail;
  };
  P.map0_w = add_param(P, "map_layer0.weight", {ch, ch});
  P.map0_b = add_param(P, "map_layer0.bias", {ch});
  P.map1_w = add_param(P, "map_layer1.weight", {ch, ch});
  P.map1_b = add_param(P, "map_layer1.bias", {ch});
  if (d->dx_mode == MCEDM_DX_ENC) {        // registered before self.enc (adm_blocks.py:266-280): state_dict order
    const int c0 = ch * d->ch_mult[0];
    P.dx_enc0 = make_conv(P, "dx_enc.0", d->dx_channels, c0, 3);
    P.dx_enc2 = make_conv(P, "dx_enc.2", c0, c0, 3);
    P.combine = make_conv(P, "combine_enc", 2 * c0, c0, 3);
  }
  std::vector<int> skips;
  int cout = d->in_channels + d->cond_channels + (d->dx_mode == MCEDM_DX_CAT ? d->dx_channels : 0);   // adm_blocks.py:236-238
  for (int level = 0; level < d->n_levels; ++level) {
    const int res = d->resolution >> level, mult = d->ch_mult[level];
    if (level == 0) {
      const int cin = cout;
      cout = ch * mult;
      P.conv_in = make_conv(P, key("enc", res, "conv"), cin, cout, 3);
      skips.push_back(cout);
    } else {
      P.enc.push_back(make_block(P, key("enc", res, "down"), cout, cout, false, true, false));
      skips.push_back(cout);
    }
    for (int idx = 0; idx < d->num_res_blocks; ++idx) {
      const int cin = cout;
      cout = ch * mult;
      const bool attn = in_list(d->attn_resolutions, d->n_attn_resolutions, res);
      P.enc.push_back(make_block(P, key("enc", res, "block" + std::to_string(idx)), cin, cout, false, false, attn));
      skips.push_back(cout);
    }
  }
  for (int level = d->n_levels - 1; level >= 0; --level) {
    const int res = d->resolution >> level, mult = d->ch_mult[level];
    if (level == d->n_levels - 1) {
      P.dec.push_back(make_block(P, key("dec", res, "in0"), cout, cout, false, false, true));
      P.dec.push_back(make_block(P, key("dec", res, "in1"), cout, cout, false, false, false));
    } else {
      P.dec.push_back(make_block(P, key("dec", res, "up"), cout, cout, true, false, false));
    }
    for (int idx = 0; idx < d->num_res_blocks + 1; ++idx) {
      const int cin = cout + skips.back();
      skips.pop_back();
      cout = ch * mult;
      const bool attn = in_list(d->attn_resolutions, d->n_attn_resolutions, res);
      P.dec.push_back(make_block(P, key("dec", res, "block" + std::to_string(idx)), cin, cout, false, false, attn));
    }
  }
  P.out_norm = make_norm(P, "out_norm", cout);
  P.conv_out = make_conv(P, "out_conv", cout, d->out_channels, 3);
  // the attention kernels are built for head_dim == 64; the reference's head_dim is cout / (cout // 64)
  // (adm_blocks.py:135,175), which differs as soon as cout is not a multiple of 64 (e.g. 96 -> one head of 96)
  for (auto* v : {&P.enc, &P.dec})
    for (const BlockP& b : *v)
      if (b.attn && b.cout % d->channels_per_head != 0) {
        set_error("plan_create: block %s has attention with %d channels, not a multiple of channels_per_head=%d "
                  "(head_dim %d is not built)", b.key.c_str(), b.cout, d->channels_per_head, b.cout / b.heads);
        delete Pp;
        return MCEDM_ERR_UNSUPPORTED;
      }

  // packed-buffer layout
  Taker t;
  P.freqs = t.take(ch / 2);
  P.w0 = t.take((size_t)ch * ch); P.b0 = t.take(ch);
  P.w1 = t.take((size_t)ch * ch); P.b1 = t.take(ch);
  int rows = 0;
  for (auto* v : {&P.enc, &P.dec})
    for (BlockP& b : *v) { b.film_row0 = rows; rows += 2 * b.cout; }
  P.film_rows = rows;
  P.waff = t.take((size_t)rows * ch);
  P.baff = t.take(rows);
  place_conv(t, P.conv_in, false);
  if (d->dx_mode == MCEDM_DX_ENC) { place_conv(t, P.dx_enc0, false); place_conv(t, P.dx_enc2, true); place_conv(t, P.combine, true); }
  for (auto* v : {&P.enc, &P.dec})
    for (BlockP& b : *v) {
      place_norm(t, b.norm0); place_conv(t, b.conv0, true);
      place_norm(t, b.norm1); place_conv(t, b.conv1, true);
      if (b.skip_kernel == 1) place_conv(t, b.skip, true);
      if (b.attn) { place_norm(t, b.norm2); place_conv(t, b.qkv, true); place_conv(t, b.proj, true); }
    }
  place_norm(t, P.out_norm);
  place_conv(t, P.conv_out, true);
  P.packed_floats = t.cur;
  *out = Pp;
  return MCEDM_OK;
}

extern "C" void mcedm_unet_plan_destroy(mcedm_plan* plan) { delete plan; }

extern "C" int mcedm_unet_param_count(const mcedm_plan* plan) { return plan ? (int)plan->params.size() : MCEDM_ERR_INVALID; }

extern "C" int mcedm_unet_param_info(const mcedm_plan* plan, int index, const char** name, int64_t* numel, int32_t* ndim,
                                     int64_t shape[4]) {
  MCEDM_REQUIRE(plan && index >= 0 && index < (int)plan->params.size(), "param_info: index %d out of range", index);
  const ParamInfo& p = plan->params[index];
  if (name) *name = p.name.c_str();
  if (numel) *numel = p.numel;
  if (ndim) *ndim = p.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  return MCEDM_OK;
}

extern "C" int mcedm_unet_packed_bytes(const mcedm_plan* plan, size_t* bytes) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && bytes, "packed_bytes: null argument");
  *bytes = plan->packed_floats * sizeof(float);
  return MCEDM_OK;
}

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
namespace mcedm {

constexpr int COPY_MAX = 40;
struct CopyBatch {
  const float* src[COPY_MAX];
  float* dst[COPY_MAX];
  int n[COPY_MAX];
};

__global__ void batched_copy_kernel(CopyBatch b) {
  const int j = blockIdx.y;
  const float* s = b.src[j];
  float* d = b.dst[j];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < b.n[j]; i += gridDim.x * blockDim.x) d[i] = s[i];
}

// freqs[k] = (1/10000)^(k/half)   (adm_blocks.py:193-196, endpoint=False), fp32
__global__ void freqs_kernel(float* f, int half) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < half) f[k] = powf(1.0f / 10000.0f, (float)k / (float)half);
}

// Every conv weight of a plan is re-laid per optimisation step (direct form, its data-gradient mirror, both again in
// Winograd form): as one launch per table that was ~140 kernels of 4-5 us per training step (profiles/r3_train_kernel_stats.csv:
// 805 + 560 calls in six steps).  The jobs are batched instead: one launch packs up to PACK_MAX tables, a fixed number of
// workgroups per table walking its elements (pack.hpp holds the per-element bodies, shared with the one-table kernels).
constexpr int PACK_MAX = 48;
struct PackJob {
  const float* w; float* dst;
  int Cout, Cin, taps, KC, coutp, qkv_heads, tflip;
  int kind;                     // 0: direct form (pack_conv_value), 1: Winograd F(2x2, 3x3) form (wino_pack_elem)
  unsigned total;               // elements to walk: packed floats (direct) or coutp * nch * WKC (Winograd)
};
struct PackBatch { PackJob j[PACK_MAX]; };
static_assert(sizeof(PackBatch) <= 4000, "kernel argument budget");

__global__ void pack_batch_kernel(PackBatch b) {
  const PackJob& J = b.j[blockIdx.y];
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < J.total; i += gridDim.x * blockDim.x) {
    if (J.kind == 0) J.dst[i] = pack_conv_value(J.w, i, J.Cout, J.Cin, J.taps, J.KC, J.coutp, J.qkv_heads, J.tflip);
    else wino_pack_elem(J.w, J.dst, (int)i, J.Cout, J.Cin, J.coutp, J.tflip);
  }
}

struct Packer {
  PackBatch b;
  int count = 0;
  hipStream_t s;
  int status = MCEDM_OK;
  double bytes = 0.0;           // read + written by the batch (profiler row: an HBM-bound kernel)
  void flush() {
    if (count == 0 || status != MCEDM_OK) { count = 0; bytes = 0.0; return; }
    {
      ProfScope ps("pack_batch_kernel", 0.0, bytes, s);
      hipLaunchKernelGGL(pack_batch_kernel, dim3(48, count), dim3(256), 0, s, b);
      if (hipGetLastError() != hipSuccess) { set_error("pack_batch_kernel launch failed"); status = MCEDM_ERR_HIP; }
    }
    count = 0; bytes = 0.0;
  }
  void add(const PackJob& j) {
    b.j[count] = j;
    // direct tables: one float read per float written; Winograd: 9 taps read, 16 positions written per (cout, cin)
    bytes += j.kind == 0 ? 8.0 * j.total : 4.0 * (9.0 + 16.0) * j.total;
    if (++count == PACK_MAX) flush();
  }
  void conv(const float* w, float* dst, int Cout, int Cin, int taps, int qkv_heads, int tflip) {
    const size_t total = conv_packed_floats(Cout, Cin, taps);
    if (total >= (1ull << 32)) { set_error("pack: table too large"); status = MCEDM_ERR_INVALID; return; }
    add(PackJob{w, dst, Cout, Cin, taps, conv_kc_for(taps), cout_padded(Cout), qkv_heads, tflip, 0, (unsigned)total});
  }
  void wino(const float* w, float* dst, int Cout, int Cin, int tflip) {
    const int coutp = cout_padded(Cout), nch = ceil_div(Cin, WKC);
    add(PackJob{w, dst, Cout, Cin, 9, WKC, coutp, 0, tflip, 1, (unsigned)(coutp * nch * WKC)});
  }
};

struct Copier {
  CopyBatch b;
  int count = 0;
  hipStream_t s;
  int status = MCEDM_OK;
  void flush() {
    if (count == 0 || status != MCEDM_OK) { count = 0; return; }
    hipLaunchKernelGGL(batched_copy_kernel, dim3(8, count), dim3(256), 0, s, b);
    if (hipGetLastError() != hipSuccess) { set_error("batched_copy_kernel launch failed"); status = MCEDM_ERR_HIP; }
    count = 0;
  }
  void add(const float* src, float* dst, size_t n) {
    b.src[count] = src; b.dst[count] = dst; b.n[count] = (int)n;
    if (++count == COPY_MAX) flush();
  }
};

static int pack_conv(const ConvP& c, const float* const* params, float* pk, Copier& cp, Packer& pp, hipStream_t s) {
  pp.conv(params[c.w], pk + c.wpk, c.cout, c.cin, c.taps, c.qkv_heads, 0);
  int rc = MCEDM_OK;
  if (c.qkv_heads > 0) rc = launch_pack_bias(params[c.b], pk + c.bias, c.cout, c.qkv_heads, s);
  else cp.add(params[c.b], pk + c.bias, c.cout);
  if (rc) return rc;
  // data-gradient GEMM: output channels = conv input channels; for the qkv conv the K index follows the packed
  // (head, which, c) order of the incoming gradient rows
  if (c.wpk_dgrad != NONE) pp.conv(params[c.w], pk + c.wpk_dgrad, c.cin, c.cout, c.taps, c.qkv_heads, 1);
  if (c.wino != NONE) pp.wino(params[c.w], pk + c.wino, c.cout, c.cin, 0);
  if (c.wino_dgrad != NONE) pp.wino(params[c.w], pk + c.wino_dgrad, c.cin, c.cout, 1);
  return pp.status;
}

static void pack_norm(const NormP& n, const float* const* params, float* pk, Copier& cp) {
  cp.add(params[n.w], pk + n.gamma, n.C);
  cp.add(params[n.b], pk + n.beta, n.C);
}

}  // namespace mcedm

extern "C" int mcedm_unet_pack_weights(const mcedm_plan* plan, const float* const* params, void* packed, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && params && packed, "pack_weights: null argument");
  const mcedm_plan& P = *plan;
  for (size_t i = 0; i < P.params.size(); ++i)
    MCEDM_REQUIRE(params[i] != nullptr, "pack_weights: parameter %zu (%s) is null", i, P.params[i].name.c_str());
  hipStream_t s = (hipStream_t)stream;
  float* pk = (float*)packed;
  const int ch = P.desc.ch;
  hipLaunchKernelGGL(freqs_kernel, dim3(ceil_div(ch / 2, 64)), dim3(64), 0, s, pk + P.freqs, ch / 2);
  MCEDM_LAUNCH_CHECK("freqs_kernel");
  Copier cp;
  cp.s = s;
  Packer pp;
  pp.s = s;
  cp.add(params[P.map0_w], pk + P.w0, (size_t)ch * ch); cp.add(params[P.map0_b], pk + P.b0, ch);
  cp.add(params[P.map1_w], pk + P.w1, (size_t)ch * ch); cp.add(params[P.map1_b], pk + P.b1, ch);
  int rc = pack_conv(P.conv_in, params, pk, cp, pp, s);
  if (rc) return rc;
  if (P.desc.dx_mode == MCEDM_DX_ENC) {
    if ((rc = pack_conv(P.dx_enc0, params, pk, cp, pp, s))) return rc;
    if ((rc = pack_conv(P.dx_enc2, params, pk, cp, pp, s))) return rc;
    if ((rc = pack_conv(P.combine, params, pk, cp, pp, s))) return rc;
  }
  for (auto* v : {&P.enc, &P.dec})
    for (const BlockP& b : *v) {
      cp.add(params[b.aff_w], pk + P.waff + (size_t)b.film_row0 * ch, (size_t)2 * b.cout * ch);
      cp.add(params[b.aff_b], pk + P.baff + b.film_row0, (size_t)2 * b.cout);
      pack_norm(b.norm0, params, pk, cp);
      pack_norm(b.norm1, params, pk, cp);
      if ((rc = pack_conv(b.conv0, params, pk, cp, pp, s))) return rc;
      if ((rc = pack_conv(b.conv1, params, pk, cp, pp, s))) return rc;
      if (b.skip_kernel == 1 && (rc = pack_conv(b.skip, params, pk, cp, pp, s))) return rc;
      if (b.attn) {
        pack_norm(b.norm2, params, pk, cp);
        if ((rc = pack_conv(b.qkv, params, pk, cp, pp, s))) return rc;
        if ((rc = pack_conv(b.proj, params, pk, cp, pp, s))) return rc;
      }
    }
  pack_norm(P.out_norm, params, pk, cp);
  if ((rc = pack_conv(P.conv_out, params, pk, cp, pp, s))) return rc;
  cp.flush();
  pp.flush();
  return cp.status != MCEDM_OK ? cp.status : pp.status;
}

// ------------------------------------------------------------------------------------------
// activation layout inside the workspace (first-fit pool, simulated on the host)
// ------------------------------------------------------------------------------------------
namespace mcedm {

struct LayoutBuilder {
  Layout& L;
  Pool pool;
  int B;
  int make(int C, int H, int W, size_t bytes) {
    TRef t;
    t.C = C; t.H = H; t.W = W; t.bytes = bytes;
    t.off = pool.alloc(bytes);
    L.t.push_back(t);
    return (int)L.t.size() - 1;
  }
  int act(int C, int H, int W) { return make(C, H, W, (size_t)B * C * H * W * sizeof(float)); }
  int coef(int C) { return make(C, 1, 1, (size_t)B * C * sizeof(Coef)); }
  int stats(int groups) { return make(groups, 1, 1, (size_t)B * groups * 2 * sizeof(float)); }
  void retain(int id) { if (id >= 0) L.t[id].ref++; }
  void release(int id) {
    if (id < 0) return;
    if (--L.t[id].ref == 0) pool.release(L.t[id].off, L.t[id].bytes);
  }
  void drop(int id) { if (id >= 0) pool.release(L.t[id].off, L.t[id].bytes); }   // un-refcounted temporaries
};

int build_layout(const mcedm_plan& P, int B, int H, int W, int training, int n_noise, Layout* out) {
  MCEDM_REQUIRE(B > 0 && H > 0 && W > 0, "layout: empty shape B=%d H=%d W=%d", B, H, W);
  MCEDM_REQUIRE(H % P.levels_div == 0 && W % P.levels_div == 0,
                "layout: H=%d W=%d must be multiples of %d (2^(levels-1))", H, W, P.levels_div);
  MCEDM_REQUIRE(n_noise == 1 || n_noise == B, "layout: n_noise=%d must be 1 or B=%d", n_noise, B);
  Layout& L = *out;
  L = Layout();
  LayoutBuilder lb{L, Pool(), B};
  lb.pool.keep_all = training != 0;
  L.film = lb.make(P.film_rows, 1, 1, (size_t)n_noise * P.film_rows * sizeof(float));
  // arena of fused GroupNorm statistics: one [B][tiles][C/4][2] fp32 table per conv output that feeds a GroupNorm
  // (conv_in output, and h / y / z of every block), sized for the smallest pixel tile
  auto sums_sz = [&](int C, int h, int w) { return align_up((size_t)B * conv_max_tiles(h, w) * ceil_div(C, 4) * 2 * sizeof(float), 256); };
  {
    size_t need = sums_sz(P.conv_in.cout, H, W);
    int h = H, w = W;
    for (auto* v : {&P.enc, &P.dec})
      for (const BlockP& b : *v) {
        if (b.up) { h *= 2; w *= 2; } else if (b.down) { h /= 2; w /= 2; }
        need += (size_t)(b.attn ? 3 : 2) * sums_sz(b.cout, h, w);
      }
    L.sums_bytes = need;
    const int id = lb.make(1, 1, 1, need);
    L.sums_base = L.t[id].off;
  }
  size_t sums_cur = L.sums_base;
  auto give_sums = [&](int id) {
    L.t[id].sums = sums_cur;
    sums_cur += sums_sz(L.t[id].C, L.t[id].H, L.t[id].W);
  };
  if (P.desc.dx_mode == MCEDM_DX_CAT) L.xdx = lb.act(P.desc.in_channels + P.desc.dx_channels, H, W);
  if (P.desc.dx_mode == MCEDM_DX_ENC) {
    L.xf = lb.act(P.conv_in.cout, H, W);
    L.d1 = lb.act(P.conv_in.cout, H, W);
    L.g1 = lb.act(P.conv_in.cout, H, W);
    L.d2 = lb.act(P.conv_in.cout, H, W);
  }
  L.t0 = lb.act(P.conv_in.cout, H, W);
  give_sums(L.t0);
  lb.drop(L.xdx); lb.drop(L.xf); lb.drop(L.d1); lb.drop(L.g1); lb.drop(L.d2);      // the head's temporaries (kept in training)
  std::vector<int> skips;
  int cur = L.t0;
  lb.retain(cur);                 // chain reference
  lb.retain(cur); skips.push_back(cur);

  auto do_block = [&](const BlockP& b, int xa, int xb) {
    BlockLayout bl;
    bl.xa = xa; bl.xb = xb;
    bl.Hin = L.t[xa].H; bl.Win = L.t[xa].W;
    bl.H = b.up ? bl.Hin * 2 : (b.down ? bl.Hin / 2 : bl.Hin);
    bl.W = b.up ? bl.Win * 2 : (b.down ? bl.Win / 2 : bl.Win);
    bl.coef0 = lb.coef(b.cin);
    if (training) bl.stats0 = lb.stats(b.norm0.groups);
    // a down block's conv0 reads avgpool2x2(silu(norm(x))): four SiLUs per staged element is paid in matrix time
    // inside the fused kernel (62 vs 120 TFLOP/s), so that one input is materialised by an HBM-speed pass instead
    if (b.down) bl.xd = lb.act(b.cin, bl.H, bl.W);
    bl.h = lb.act(b.cout, bl.H, bl.W);
    give_sums(bl.h);
    lb.drop(bl.coef0); lb.drop(bl.xd);
    bl.coef1 = lb.coef(b.cout);
    if (training) bl.stats1 = lb.stats(b.norm1.groups);
    // the 1x1 skip projection is folded into conv1 (ConvArgs::sk_*) in inference AND in training: the backward never reads
    // the projected tensor (its weight gradient takes dy and the block input, its data gradient dy alone), so it only
    // exists for the resampling blocks, which the fold does not serve
    // ... and for blocks whose conv1 the Winograd kernel serves (conv_wino.hip): a 1x1 projection has nothing to gain
    // from that transform (it would cost 16 multiplies per output instead of 1), so it runs as its own launch there
    // and enters conv1 as a residual
    (void)training;
    const bool wino1 = b.conv1.wino != NONE && conv_wino_shape_ok(b.cout, b.cout, bl.H, bl.W);
    if (b.skip_kernel == 1 && (b.up || b.down || wino1)) bl.sk = lb.act(b.cout, bl.H, bl.W);
    bl.y = lb.act(b.cout, bl.H, bl.W);
    give_sums(bl.y);
    lb.drop(bl.h); lb.drop(bl.coef1); lb.drop(bl.sk);
    bl.out = bl.y;
    if (b.attn) {
      bl.coef2 = lb.coef(b.cout);
      if (training) bl.stats2 = lb.stats(b.norm2.groups);
      bl.qkv = lb.act(3 * b.cout, bl.H, bl.W);
      lb.drop(bl.coef2);
      bl.a = lb.act(b.cout, bl.H, bl.W);
      lb.drop(bl.qkv);
      bl.z = lb.act(b.cout, bl.H, bl.W);
      give_sums(bl.z);
      lb.drop(bl.a); lb.drop(bl.y);
      bl.out = bl.z;
    }
    L.blocks.push_back(bl);
    return bl.out;
  };

  for (const BlockP& b : P.enc) {
    const int o = do_block(b, cur, -1);
    lb.retain(o);                 // chain
    lb.retain(o); skips.push_back(o);
    lb.release(cur);
    cur = o;
  }
  for (const BlockP& b : P.dec) {
    int xb = -1;
    if (L.t[cur].C != b.cin) {    // adm_blocks.py:400-401
      MCEDM_REQUIRE(!skips.empty(), "layout: skip stack underflow at %s", b.key.c_str());
      xb = skips.back(); skips.pop_back();
      MCEDM_REQUIRE(L.t[cur].C + L.t[xb].C == b.cin && L.t[cur].H == L.t[xb].H && L.t[cur].W == L.t[xb].W,
                    "layout: concat mismatch at %s", b.key.c_str());
    }
    const int o = do_block(b, cur, xb);
    lb.retain(o);
    lb.release(cur);
    lb.release(xb);
    cur = o;
  }
  L.last = cur;
  L.coef_out = lb.coef(P.out_norm.C);
  if (training) L.stats_out = lb.stats(P.out_norm.groups);
  L.total_bytes = lb.pool.peak;
  return MCEDM_OK;
}

// header in front of the U-Net activations: EDM coefficient rows, conv_in transform, F / F_uncond
Header header_for(const mcedm_plan& P, int B, int H, int W) {
  Header h;
  size_t cur = 0;
  auto take = [&](size_t bytes) { size_t o = cur; cur += align_up(bytes, 256); return o; };
  const int Ct = P.desc.in_channels + P.desc.cond_channels + P.desc.dx_channels;
  h.coefs4 = take((size_t)B * 4 * sizeof(float));
  h.c_noise = take((size_t)B * sizeof(float));
  h.coef_in = take((size_t)B * Ct * sizeof(Coef));
  h.F = take((size_t)B * P.desc.out_channels * H * W * sizeof(float));
  h.Fu = take((size_t)B * P.desc.out_channels * H * W * sizeof(float));
  h.total = cur;
  return h;
}


// ------------------------------------------------------------------------------------------
// forward schedule (adm_blocks.py:364-404 and :159-181)
// ------------------------------------------------------------------------------------------
// Inference: the consuming conv derives the GroupNorm(+FiLM) rows itself from the producers' partial sums
// (ConvArgs::gn_on), so no GroupNorm kernel is launched.  Training keeps the table (wgrad / GN backward read it and
// the saved statistics), as does any consumer that needs the table in memory (need_table) or a shape without
// usable partial sums.
static int gn_for_conv(const GnArgs& g, ConvArgs& c, bool need_table, hipStream_t s) {
  if (!need_table && g.stats == nullptr && gn_sums_usable(g)) {
    c.gn = g; c.gn_on = 1; c.coef = nullptr;
    return MCEDM_OK;
  }
  return launch_gn_coef_from_sums(g, s);
}

static int run_block(const mcedm_plan& P, const BlockP& b, const BlockLayout& bl, const Layout& L, void* act,
                     const float* pk, int B, int n_noise, hipStream_t s, std::vector<SumTiles>& st) {
  auto T = [&](int id) -> float* { return id < 0 ? nullptr : at<float>(act, L.t[id].off); };
  auto CF = [&](int id) -> Coef* { return id < 0 ? nullptr : at<Coef>(act, L.t[id].off); };
  auto SUMS = [&](int id) -> float* { return (id < 0 || L.t[id].sums == NONE) ? nullptr : at<float>(act, L.t[id].sums); };
  auto TL = [&](int id) -> SumTiles { return id < 0 ? SumTiles{} : st[id]; };      // tiling of that tensor's statistics table
  const float* xa = T(bl.xa);
  const float* xb = T(bl.xb);
  const int Ca = L.t[bl.xa].C, Cb = bl.xb >= 0 ? L.t[bl.xb].C : 0;
  const float eps = P.desc.eps;
  int rc;
  // norm0 -> transform table for conv0
  GnArgs g0{xa, xb, Ca, Cb, bl.Hin * bl.Win, B, b.norm0.groups, pk + b.norm0.gamma, pk + b.norm0.beta,
            nullptr, 0, 0, eps, CF(bl.coef0), T(bl.stats0), SUMS(bl.xa), SUMS(bl.xb), TL(bl.xa), TL(bl.xb), bl.Win};
  // h = conv0(resample(silu(norm0(x))))
  ConvArgs c0{};
  c0.xa = xa; c0.xb = xb; c0.Ca = Ca; c0.Cb = Cb;
  c0.coef = CF(bl.coef0); c0.coef_batch = 1; c0.act = 1;
  const int rs = b.up ? RS_UP : (b.down ? RS_DOWN : RS_NONE);
  c0.resample = rs;
  c0.Hs = bl.Hin; c0.Ws = bl.Win; c0.H = bl.H; c0.W = bl.W;
  c0.wpk = pk + b.conv0.wpk; c0.bias = pk + b.conv0.bias;
  c0.wino = b.conv0.wino != NONE ? pk + b.conv0.wino : nullptr;
  c0.out = T(bl.h); c0.Cout = b.cout; c0.B = B; c0.gsum = SUMS(bl.h); c0.gsum_tiles = &st[bl.h];
  if ((rc = gn_for_conv(g0, c0, /*need_table=*/bl.xd >= 0, s))) return rc;
  if (bl.xd >= 0) {
    WgradArgs m{};
    m.xa = xa; m.xb = xb; m.Ca = Ca; m.Cb = Cb; m.coef = c0.coef; m.coef_batch = 1; m.act = 1; m.resample = RS_DOWN;
    m.Hs = bl.Hin; m.Ws = bl.Win; m.H = bl.H; m.W = bl.W; m.B = B;
    if ((rc = launch_act_materialize(m, T(bl.xd), s))) return rc;
    c0.xa = T(bl.xd); c0.xb = nullptr; c0.Ca = Ca + Cb; c0.Cb = 0;
    c0.coef = nullptr; c0.act = 0; c0.resample = RS_NONE; c0.Hs = bl.H; c0.Ws = bl.W;
  }
  if ((rc = launch_conv(c0, 9, s))) return rc;
  // norm1 + FiLM -> transform table for conv1
  const float* film = at<float>(act, L.t[L.film].off) + b.film_row0;
  GnArgs g1{T(bl.h), nullptr, b.cout, 0, bl.H * bl.W, B, b.norm1.groups, pk + b.norm1.gamma, pk + b.norm1.beta,
            film, n_noise > 1 ? 1 : 0, P.film_rows, eps, CF(bl.coef1), T(bl.stats1), SUMS(bl.h), nullptr, TL(bl.h), SumTiles{}, bl.W};
  // skip path
  const float* res = xa;
  int res_mode = RS_NONE;
  const bool fold_skip = b.skip_kernel == 1 && bl.sk < 0;      // see build_layout
  if (b.skip_kernel == 1 && !fold_skip) {
    ConvArgs cs{};
    cs.xa = xa; cs.xb = xb; cs.Ca = Ca; cs.Cb = Cb;
    cs.resample = rs; cs.Hs = bl.Hin; cs.Ws = bl.Win; cs.H = bl.H; cs.W = bl.W;
    cs.wpk = pk + b.skip.wpk; cs.bias = pk + b.skip.bias;
    cs.out = T(bl.sk); cs.Cout = b.cout; cs.B = B;
    if ((rc = launch_conv(cs, 1, s))) return rc;
    res = T(bl.sk);
  } else if (b.skip_kernel == 0) {
    res_mode = rs;
  }
  // y = conv1(silu(film(norm1(h)))) + skip
  ConvArgs c1{};
  c1.xa = T(bl.h); c1.Ca = b.cout;
  c1.coef = CF(bl.coef1); c1.coef_batch = 1; c1.act = 1;
  c1.Hs = bl.H; c1.Ws = bl.W; c1.H = bl.H; c1.W = bl.W;
  c1.wpk = pk + b.conv1.wpk; c1.bias = pk + b.conv1.bias;
  c1.wino = b.conv1.wino != NONE ? pk + b.conv1.wino : nullptr;
  c1.res = res; c1.res_mode = res_mode;
  if (fold_skip) {            // y = conv1(...) + skip(orig): the projection rides on conv1 as extra K chunks
    c1.res = nullptr; c1.res_mode = RS_NONE;
    c1.sk_xa = xa; c1.sk_xb = xb; c1.sk_Ca = Ca; c1.sk_Cb = Cb;
    c1.sk_wpk = pk + b.skip.wpk; c1.sk_bias = pk + b.skip.bias;
  }
  c1.out = T(bl.y); c1.Cout = b.cout; c1.B = B; c1.gsum = SUMS(bl.y); c1.gsum_tiles = &st[bl.y];
  if ((rc = gn_for_conv(g1, c1, false, s))) return rc;
  if ((rc = launch_conv(c1, 9, s))) return rc;
  if (!b.attn) return MCEDM_OK;
  // attention: z = proj(attn(qkv(norm2(y)))) + y
  if (bl.stats2 < 0 && attn_block_fused_applicable(b.cout, b.heads, bl.H, bl.W, b.norm2.groups))     // inference, 8 x 8 x 64: one launch
    return launch_attn_block64(T(bl.y), T(bl.z), pk + b.norm2.gamma, pk + b.norm2.beta, eps, b.norm2.groups, pk + b.qkv.wpk,
                               pk + b.qkv.bias, pk + b.proj.wpk, pk + b.proj.bias, SUMS(bl.z), &st[bl.z], B, s);
  GnArgs g2{T(bl.y), nullptr, b.cout, 0, bl.H * bl.W, B, b.norm2.groups, pk + b.norm2.gamma, pk + b.norm2.beta,
            nullptr, 0, 0, eps, CF(bl.coef2), T(bl.stats2), SUMS(bl.y), nullptr, TL(bl.y), SumTiles{}, bl.W};
  ConvArgs cq{};
  cq.xa = T(bl.y); cq.Ca = b.cout;
  cq.coef = CF(bl.coef2); cq.coef_batch = 1; cq.act = 0;
  cq.Hs = bl.H; cq.Ws = bl.W; cq.H = bl.H; cq.W = bl.W;
  cq.wpk = pk + b.qkv.wpk; cq.bias = pk + b.qkv.bias;
  cq.out = T(bl.qkv); cq.Cout = 3 * b.cout; cq.B = B;
  if ((rc = gn_for_conv(g2, cq, false, s))) return rc;
  if ((rc = launch_conv(cq, 1, s))) return rc;
  if ((rc = launch_attention(T(bl.qkv), T(bl.a), B, b.heads, bl.H * bl.W, s))) return rc;
  ConvArgs cp{};
  cp.xa = T(bl.a); cp.Ca = b.cout;
  cp.Hs = bl.H; cp.Ws = bl.W; cp.H = bl.H; cp.W = bl.W;
  cp.wpk = pk + b.proj.wpk; cp.bias = pk + b.proj.bias;
  cp.res = T(bl.y); cp.res_mode = RS_NONE;
  cp.out = T(bl.z); cp.Cout = b.cout; cp.B = B; cp.gsum = SUMS(bl.z); cp.gsum_tiles = &st[bl.z];
  return launch_conv(cp, 1, s);
}

// ---- dx_cond head (adm_blocks.py:334-362) ------------------------------------------------------------------
// out[B, cx + cd, HW] = cat(x[B, cx, HW], dx[B, cd, HW] or zeros)   (cat_dx: adm_blocks.py:335-339)
__global__ void concat_x_dx_kernel(const float* __restrict__ x, const float* __restrict__ dx, int cx, int cd, size_t hw,
                                   size_t total, float* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t per = (size_t)(cx + cd) * hw;
    const size_t n = i / per, r = i - n * per;
    out[i] = r < (size_t)cx * hw ? x[n * cx * hw + r] : (dx ? dx[n * cd * hw + (r - (size_t)cx * hw)] : 0.f);
  }
}
// torch.nn.GELU() (approximate='none'): y = 0.5 v (1 + erf(v / sqrt(2)));  mode 1: out = g * dGELU(v)
__global__ void gelu_kernel(const float* __restrict__ v, const float* g, float* out, size_t n, int mode) {      // out may be g
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float a = v[i];
    const float cdf = 0.5f * (1.0f + erff(a * 0.70710678118654752440f));
    out[i] = mode ? g[i] * (cdf + a * 0.39894228040143267794f * expf(-0.5f * a * a)) : a * cdf;
  }
}
int launch_gelu(const float* v, const float* g, float* out, size_t n, int mode, hipStream_t s) {
  const size_t b = (n + 255) / 256;
  hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)(b < 4096 ? (b ? b : 1) : 4096)), dim3(256), 0, s, v, g, out, n, mode);
  MCEDM_LAUNCH_CHECK("gelu_kernel");
  return MCEDM_OK;
}

// act = start of the activation region (after the header)
static int forward_impl(const mcedm_plan& P, const Layout& L, const float* pk, const float* x, const float* dx, const float* cond,
                        const Coef* coef_in, int coef_batch, const float* noise_labels, int n_noise, float* out,
                        void* act, int B, int H, int W, hipStream_t s) {
  const int ch = P.desc.ch;
  const int dxm = P.desc.dx_mode;
  int rc;
  EmbArgs e{noise_labels, n_noise, ch, pk + P.freqs, pk + P.w0, pk + P.b0, pk + P.w1, pk + P.b1,
            pk + P.waff, pk + P.baff, P.film_rows, nullptr, at<float>(act, L.t[L.film].off)};
  if ((rc = launch_embedding(e, s))) return rc;
  std::vector<SumTiles> st(L.t.size());     // tiling of each tensor's fused-statistics table
  // conv_in on cat(cond, x)  (cond FIRST, adm_blocks.py:332)
  ConvArgs ci{};
  ci.xa = cond; ci.Ca = P.desc.cond_channels;
  ci.xb = x; ci.Cb = P.desc.in_channels;
  if (dxm == MCEDM_DX_CAT) {         // cat(cond, x, dx): x and dx meet in one staging tensor (the kernels take two sources)
    float* xdx = at<float>(act, L.t[L.xdx].off);
    const size_t hw = (size_t)H * W, total = (size_t)B * (P.desc.in_channels + P.desc.dx_channels) * hw;
    const size_t nb = (total + 255) / 256;
    hipLaunchKernelGGL(concat_x_dx_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, s, x, dx, P.desc.in_channels,
                       P.desc.dx_channels, hw, total, xdx);
    MCEDM_LAUNCH_CHECK("concat_x_dx_kernel");
    ci.xb = xdx; ci.Cb = P.desc.in_channels + P.desc.dx_channels;
  }
  ci.coef = coef_in; ci.coef_batch = coef_batch; ci.act = 0;
  ci.Hs = H; ci.Ws = W; ci.H = H; ci.W = W;
  ci.wpk = pk + P.conv_in.wpk; ci.bias = pk + P.conv_in.bias;
  ci.out = at<float>(act, L.t[L.t0].off); ci.Cout = P.conv_in.cout; ci.B = B;
  ci.gsum = at<float>(act, L.t[L.t0].sums); ci.gsum_tiles = &st[L.t0];
  if (dxm == MCEDM_DX_ENC) { ci.out = at<float>(act, L.t[L.xf].off); ci.gsum = nullptr; ci.gsum_tiles = nullptr; }
  if ((rc = launch_conv(ci, 9, s))) return rc;
  if (dxm == MCEDM_DX_ENC) {
    // x_feat = combine_enc(cat(conv_in(.), dx_enc(dx))), dx_enc = Conv3x3 -> GELU -> Conv3x3; dx None: zero features, NOT
    // dx_enc(0) (adm_blocks.py:352-362)
    const int c0 = P.conv_in.cout;
    float* d2 = at<float>(act, L.t[L.d2].off);
    auto plain = [&](const ConvP& cv, const float* xa, int Ca, const float* xb, int Cb, float* o) {
      ConvArgs c{};
      c.xa = xa; c.Ca = Ca; c.xb = xb; c.Cb = Cb;
      c.Hs = H; c.Ws = W; c.H = H; c.W = W;
      c.wpk = pk + cv.wpk; c.bias = pk + cv.bias;
      c.wino = cv.wino != NONE ? pk + cv.wino : nullptr;
      c.out = o; c.Cout = cv.cout; c.B = B;
      return c;
    };
    if (dx) {
      float* d1 = at<float>(act, L.t[L.d1].off);
      float* g1 = at<float>(act, L.t[L.g1].off);
      ConvArgs ca = plain(P.dx_enc0, dx, P.desc.dx_channels, nullptr, 0, d1);
      if ((rc = launch_conv(ca, 9, s))) return rc;
      if ((rc = launch_gelu(d1, nullptr, g1, (size_t)B * c0 * H * W, 0, s))) return rc;
      ConvArgs cb = plain(P.dx_enc2, g1, c0, nullptr, 0, d2);
      if ((rc = launch_conv(cb, 9, s))) return rc;
    } else {
      MCEDM_HIP_TRY(hipMemsetAsync(d2, 0, L.t[L.d2].bytes, s));
    }
    ConvArgs cc = plain(P.combine, at<float>(act, L.t[L.xf].off), c0, d2, c0, at<float>(act, L.t[L.t0].off));
    cc.gsum = at<float>(act, L.t[L.t0].sums); cc.gsum_tiles = &st[L.t0];
    if ((rc = launch_conv(cc, 9, s))) return rc;
  }
  size_t bi = 0;
  for (const BlockP& b : P.enc) if ((rc = run_block(P, b, L.blocks[bi++], L, act, pk, B, n_noise, s, st))) return rc;
  for (const BlockP& b : P.dec) if ((rc = run_block(P, b, L.blocks[bi++], L, act, pk, B, n_noise, s, st))) return rc;
  // out = out_conv(silu(out_norm(x)))
  const TRef& last = L.t[L.last];
  GnArgs go{at<float>(act, last.off), nullptr, last.C, 0, H * W, B, P.out_norm.groups, pk + P.out_norm.gamma,
            pk + P.out_norm.beta, nullptr, 0, 0, 1e-5f, at<Coef>(act, L.t[L.coef_out].off),
            L.stats_out >= 0 ? at<float>(act, L.t[L.stats_out].off) : nullptr,
            last.sums != NONE ? at<float>(act, last.sums) : nullptr, nullptr, st[L.last], SumTiles{}, W};
  ConvArgs co{};
  co.xa = at<float>(act, last.off); co.Ca = last.C;
  co.coef = at<Coef>(act, L.t[L.coef_out].off); co.coef_batch = 1; co.act = 1;
  co.Hs = H; co.Ws = W; co.H = H; co.W = W;
  co.wpk = pk + P.conv_out.wpk; co.bias = pk + P.conv_out.bias;
  co.out = out; co.Cout = P.desc.out_channels; co.B = B;
  if ((rc = gn_for_conv(go, co, false, s))) return rc;
  return launch_conv(co, 9, s);
}

__global__ void scale_to_coef_kernel(const float* __restrict__ x_scale, int n, int cond_ch, int in_ch, int dx_ch, Coef* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int Ct = cond_ch + in_ch + dx_ch;
  for (int c = 0; c < Ct; ++c) out[(size_t)i * Ct + c] = Coef{0.f, (c < cond_ch || c >= cond_ch + in_ch) ? 1.0f : x_scale[i], 0.f, 0.f};
}

// D = c_skip x + c_out F(c_in x; ln(sigma)/4; cond) with sigma from the device (n_sigma rows) or from the host
static int denoise_impl(const mcedm_plan& P, const Layout& L, const Header& hd, const float* pk, const float* x, const float* dx,
                        const float* sigma_dev, float sigma_host, int use_host, int n_sigma, const float* cond,
                        float w, float* D_out, float* F_out, void* ws, int B, int H, int W, float sigma_data,
                        hipStream_t s) {
  int rc;
  float* coefs4 = at<float>(ws, hd.coefs4);
  float* c_noise = at<float>(ws, hd.c_noise);
  Coef* coef_in = at<Coef>(ws, hd.coef_in);
  float* Fbuf = at<float>(ws, hd.F);
  void* act = at<char>(ws, hd.total);
  if ((rc = launch_precond_prepare(sigma_dev, sigma_host, use_host, n_sigma, sigma_data, P.desc.cond_channels,
                                   P.desc.in_channels, P.desc.dx_mode == MCEDM_DX_CAT ? P.desc.dx_channels : 0, coefs4, c_noise,
                                   coef_in, s))) return rc;
  if ((rc = forward_impl(P, L, pk, x, dx, cond, coef_in, n_sigma > 1 ? 1 : 0, c_noise, n_sigma, Fbuf, act, B, H, W, s))) return rc;
  const float* Fu = nullptr;
  // classifier-free branch, mcedm.py:453-458; models/ddim.py:1755-1760: taken when cond OR dx is given, and the
  // unconditional evaluation drops both
  if (fabsf(w) >= 0.001f && (cond != nullptr || dx != nullptr)) {
    float* Fubuf = at<float>(ws, hd.Fu);
    if ((rc = forward_impl(P, L, pk, x, nullptr, nullptr, coef_in, n_sigma > 1 ? 1 : 0, c_noise, n_sigma, Fubuf, act, B, H, W, s))) return rc;
    Fu = Fubuf;
  }
  const size_t per = (size_t)P.desc.out_channels * H * W;
  return launch_precond_finish(x, Fbuf, Fu, w, coefs4, n_sigma, per, per * B, D_out, F_out, s);
}

}  // namespace mcedm

extern "C" int mcedm_unet_workspace_bytes(const mcedm_plan* plan, int B, int H, int W, int training, size_t* bytes) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && bytes, "workspace_bytes: null argument");
  Layout L;
  int rc = build_layout(*plan, B, H, W, training, B, &L);
  if (rc) return rc;
  *bytes = header_for(*plan, B, H, W).total + L.total_bytes + (training ? backward_scratch_bytes(*plan, L, B, H, W) : 0);
  return MCEDM_OK;
}

extern "C" int mcedm_unet_forward_dx(const mcedm_plan* plan, const void* packed, const float* x, const float* dx, const float* cond,
                                     const float* x_scale, const float* noise_labels, int n_noise, float* out,
                                     void* workspace, size_t workspace_bytes, int B, int H, int W, int training,
                                     void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && packed && x && noise_labels && out && workspace, "unet_forward: null argument");
  MCEDM_REQUIRE(dx == nullptr || plan->desc.dx_mode != MCEDM_DX_NONE, "unet_forward: dx given to a plan without dx_cond");
  Layout L;
  int rc = build_layout(*plan, B, H, W, training, n_noise, &L);
  if (rc) return rc;
  const Header hd = header_for(*plan, B, H, W);
  if (hd.total + L.total_bytes > workspace_bytes) {
    set_error("unet_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, hd.total + L.total_bytes);
    return MCEDM_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const Coef* coef_in = nullptr;
  if (x_scale) {
    Coef* c = at<Coef>(workspace, hd.coef_in);
    hipLaunchKernelGGL(scale_to_coef_kernel, dim3(ceil_div(n_noise, 64)), dim3(64), 0, s, x_scale, n_noise,
                       plan->desc.cond_channels, plan->desc.in_channels,
                       plan->desc.dx_mode == MCEDM_DX_CAT ? plan->desc.dx_channels : 0, c);
    MCEDM_LAUNCH_CHECK("scale_to_coef_kernel");
    coef_in = c;
  }
  return forward_impl(*plan, L, (const float*)packed, x, dx, cond, coef_in, n_noise > 1 ? 1 : 0, noise_labels, n_noise, out,
                      at<char>(workspace, hd.total), B, H, W, s);
}

extern "C" int mcedm_unet_forward(const mcedm_plan* plan, const void* packed, const float* x, const float* cond,
                                  const float* x_scale, const float* noise_labels, int n_noise, float* out,
                                  void* workspace, size_t workspace_bytes, int B, int H, int W, int training,
                                  void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  return mcedm_unet_forward_dx(plan, packed, x, nullptr, cond, x_scale, noise_labels, n_noise, out, workspace, workspace_bytes, B, H,
                               W, training, stream);
}

extern "C" int mcedm_edm_denoise_dx(const mcedm_plan* plan, const void* packed, const float* x, const float* dx, const float* sigma,
                                    int n_sigma, const float* cond, float* D_out, float* F_out, void* workspace,
                                    size_t workspace_bytes, int B, int H, int W, int training, double sigma_data,
                                    void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && packed && x && sigma && D_out && workspace, "edm_denoise: null argument");
  MCEDM_REQUIRE(dx == nullptr || plan->desc.dx_mode != MCEDM_DX_NONE, "edm_denoise: dx given to a plan without dx_cond");
  Layout L;
  int rc = build_layout(*plan, B, H, W, training, n_sigma, &L);
  if (rc) return rc;
  const Header hd = header_for(*plan, B, H, W);
  if (hd.total + L.total_bytes > workspace_bytes) {
    set_error("edm_denoise: workspace too small (%zu < %zu bytes)", workspace_bytes, hd.total + L.total_bytes);
    return MCEDM_ERR_WORKSPACE;
  }
  return denoise_impl(*plan, L, hd, (const float*)packed, x, dx, sigma, 0.f, 0, n_sigma, cond, 0.f, D_out, F_out, workspace,
                      B, H, W, (float)sigma_data, (hipStream_t)stream);
}

extern "C" int mcedm_edm_denoise(const mcedm_plan* plan, const void* packed, const float* x, const float* sigma,
                                 int n_sigma, const float* cond, float* D_out, float* F_out, void* workspace,
                                 size_t workspace_bytes, int B, int H, int W, int training, double sigma_data,
                                 void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  return mcedm_edm_denoise_dx(plan, packed, x, nullptr, sigma, n_sigma, cond, D_out, F_out, workspace, workspace_bytes, B, H, W,
                              training, sigma_data, stream);
}

// ------------------------------------------------------------------------------------------
// Heun sampler (models/mcedm.py:570-638)
// ------------------------------------------------------------------------------------------
extern "C" int mcedm_edm_t_steps(const mcedm_sampler_desc* sp, double* t) {
  MCEDM_REQUIRE(sp && t, "t_steps: null argument");
  MCEDM_REQUIRE(sp->timesteps >= 2, "t_steps: timesteps=%d (the reference divides by timesteps-1)", sp->timesteps);
  const double smin = std::max(sp->sigma_min, sp->net_sigma_min);   // mcedm.py:579-580
  const double smax = std::min(sp->sigma_max, sp->net_sigma_max);
  const int N = sp->timesteps;
  const double a = std::pow(smax, 1.0 / sp->rho), b = std::pow(smin, 1.0 / sp->rho) - std::pow(smax, 1.0 / sp->rho);
  for (int i = 0; i < N; ++i) t[i] = std::pow(a + (double)i / (double)(N - 1) * b, sp->rho);
  t[N] = 0.0;
  return MCEDM_OK;
}

namespace mcedm {
struct SamplerBufs { size_t x, xn, d, x32, D, dx, g, dxin, total; };
static SamplerBufs sampler_bufs(const mcedm_plan& P, int B, int H, int W) {
  SamplerBufs s;
  size_t cur = 0;
  auto take = [&](size_t bytes) { size_t o = cur; cur += align_up(bytes, 256); return o; };
  const size_t n = (size_t)B * P.desc.in_channels * H * W;
  s.x = take(n * 8); s.xn = take(n * 8); s.d = take(n * 8); s.x32 = take(n * 4); s.D = take(n * 4);
  s.dx = take(n * 4); s.g = take(n * 4);          // PDE guidance: gradient and the Darcy interior scratch
  s.dxin = take(n * 4);                           // dx_cond: the network's dx input
  s.total = cur;
  return s;
}
}  // namespace mcedm

extern "C" int mcedm_sampler_workspace_bytes(const mcedm_plan* plan, int B, int H, int W, size_t* bytes) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(plan && bytes, "sampler_workspace_bytes: null argument");
  size_t u = 0;
  int rc = mcedm_unet_workspace_bytes(plan, B, H, W, 0, &u);
  if (rc) return rc;
  *bytes = sampler_bufs(*plan, B, H, W).total + u;
  return MCEDM_OK;
}

namespace mcedm {
// dx = get_dx_log_prob(h, denoised, guide_dx) of the single-task models (models/ddim.py:641-650 -> get_dx_pde :1424-1450):
// the residual of x_unnorm = (h from the conditioning, u = the denoised state), differentiated w.r.t. x_unnorm, then the
// MEAN over the two field gradients (calc_prob=True) -> [B, 1, H, W]
// (the same call on the current noisy state instead of D is get_dx_input(h, x) with dx_norm == 'prob', ddim.py:601-613)
static int guidance_dx(const mcedm_plan& P, const mcedm_guidance_desc& g, const float* cond, const float* D, float* dx,
                       float* scratch, int B, int H, int W, hipStream_t s) {
  GuideIO io{};
  const long hw = (long)H * W;
  io.in[0] = cond; io.in[1] = D; io.gt[0] = cond; io.gt[1] = D;
  io.in_sb[0] = (long)P.desc.cond_channels * hw; io.in_sb[1] = hw; io.st = W; io.sx = 1;
  io.out[0] = dx; io.out[1] = nullptr; io.out_sb[0] = hw; io.out_sb[1] = 0; io.out_st = W; io.out_sx = 1;
  io.sub[0] = g.sub_h; io.sub[1] = g.sub_u; io.div[0] = g.div_h; io.div[1] = g.div_u;
  io.mean = 1;
  if (g.system == 1)      // SweFvLoss: half_dt = 0.5 * Tn / n_times, dx = x[1] - x[0] of gen_x, both formed by the caller in fp32
    return launch_swe_guidance(io, B, H, W, g.half_dt, g.dx, g.div_h * g.div_h, g.div_u * g.div_u, s);
  MCEDM_REQUIRE(H == W && H > 4, "guidance: the Darcy residual needs a square grid larger than 4 x 4 (got %d x %d)", H, W);
  return launch_darcy_guidance(io, scratch, B, H, g.two_dx, /*calc_prob=*/1, s);
}
}  // namespace mcedm

static int heun_sample_impl(const mcedm_plan* plan, const void* packed, const mcedm_sampler_desc* sp,
                            const float* cond, const float* mask, const float* init_noise,
                            const double* step_noise, double* out, int return_last, void* workspace,
                            size_t workspace_bytes, int B, int H, int W, const mcedm_guidance_desc* gd, void* stream,
                            const mcedm_guidance_desc* dxc = nullptr, const uint64_t* rng_seed = nullptr);

extern "C" int mcedm_heun_sample(const mcedm_plan* plan, const void* packed, const mcedm_sampler_desc* sp,
                                 const float* cond, const float* mask, const float* init_noise,
                                 const double* step_noise, double* out, int return_last, void* workspace,
                                 size_t workspace_bytes, int B, int H, int W, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  return heun_sample_impl(plan, packed, sp, cond, mask, init_noise, step_noise, out, return_last, workspace, workspace_bytes,
                          B, H, W, nullptr, stream);
}

// The churn noise of every step generated inside the kernel that applies it (Philox4x32-10 keyed by *rng_seed, draw = step
// index): no [timesteps][B][C][H][W] fp64 tensor (2.1 GB at 50 x 160 x 2 x 128 x 128, the reference's shipped sampler config,
// configs/diff_sampler/edm_sampler.yaml) and one HIP graph replays with fresh noise after the host bumps the seed.
extern "C" int mcedm_heun_sample_rng(const mcedm_plan* plan, const void* packed, const mcedm_sampler_desc* sp,
                                     const float* cond, const float* mask, const float* init_noise, const uint64_t* rng_seed,
                                     double* out, int return_last, void* workspace, size_t workspace_bytes, int B, int H, int W,
                                     void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(rng_seed != nullptr, "heun_sample_rng: rng_seed (a 64-bit seed in device memory) is null");
  return heun_sample_impl(plan, packed, sp, cond, mask, init_noise, nullptr, out, return_last, workspace, workspace_bytes,
                          B, H, W, nullptr, stream, nullptr, rng_seed);
}

extern "C" int mcedm_heun_sample_guided(const mcedm_plan* plan, const void* packed, const mcedm_sampler_desc* sp,
                                        const mcedm_guidance_desc* gd, const float* cond, const float* mask,
                                        const float* init_noise, const double* step_noise, double* out, int return_last,
                                        void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(gd != nullptr && (gd->system == 1 || gd->system == 2), "heun_sample_guided: guidance system must be 1 (SWE) or 2 (Darcy)");
  MCEDM_REQUIRE(plan && plan->desc.in_channels == 1 && plan->desc.cond_channels >= 1 && cond != nullptr && mask == nullptr,
                "heun_sample_guided: PDE guidance is defined for the single-task sampler (state u, conditioning h in cond[:, 0]; "
                "mask NULL), models/ddim.py:1532-1601; the joint model's hook fails in the reference (models/mcedm.py:500-518)");
  return heun_sample_impl(plan, packed, sp, cond, mask, init_noise, step_noise, out, return_last, workspace, workspace_bytes,
                          B, H, W, gd, stream);
}

extern "C" int mcedm_heun_sample_dxcond(const mcedm_plan* plan, const void* packed, const mcedm_sampler_desc* sp,
                                        const mcedm_guidance_desc* dxc, const mcedm_guidance_desc* gd, const float* cond,
                                        const float* init_noise, const double* step_noise, double* out, int return_last,
                                        void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream) {
  VariantScope variant_scope__(plan ? &plan->variants : nullptr);
  MCEDM_REQUIRE(dxc != nullptr && (dxc->system == 1 || dxc->system == 2), "heun_sample_dxcond: dx system must be 1 (SWE) or 2 (Darcy)");
  MCEDM_REQUIRE(gd == nullptr || gd->system == 1 || gd->system == 2, "heun_sample_dxcond: guidance system must be 1 (SWE) or 2 (Darcy)");
  MCEDM_REQUIRE(plan && plan->desc.dx_mode != MCEDM_DX_NONE && plan->desc.dx_channels == 1 && plan->desc.in_channels == 1 &&
                    plan->desc.cond_channels >= 1 && cond != nullptr,
                "heun_sample_dxcond: needs a dx_cond plan of the single-task model (state u, conditioning h in cond[:, 0], one dx "
                "channel), models/ddim.py:1424-1450, 1532-1601");
  return heun_sample_impl(plan, packed, sp, cond, nullptr, init_noise, step_noise, out, return_last, workspace, workspace_bytes,
                          B, H, W, gd, stream, dxc);
}

static int heun_sample_impl(const mcedm_plan* plan, const void* packed, const mcedm_sampler_desc* sp,
                            const float* cond, const float* mask, const float* init_noise,
                            const double* step_noise, double* out, int return_last, void* workspace,
                            size_t workspace_bytes, int B, int H, int W, const mcedm_guidance_desc* gd, void* stream,
                            const mcedm_guidance_desc* dxc, const uint64_t* rng_seed) {
  MCEDM_REQUIRE(plan && packed && sp && init_noise && out && workspace, "heun_sample: null argument");
  const mcedm_plan& P = *plan;
  MCEDM_REQUIRE(P.desc.in_channels == P.desc.out_channels, "heun_sample: in_channels != out_channels");
  MCEDM_REQUIRE(mask == nullptr || (cond != nullptr && P.desc.cond_channels >= P.desc.in_channels),
                "heun_sample: with a mask, cond must carry hu_known in its first %d channels", P.desc.in_channels);
  MCEDM_REQUIRE(sp->timesteps >= 2 && sp->timesteps <= 4096, "heun_sample: timesteps=%d out of range", sp->timesteps);
  const int N = sp->timesteps;
  std::vector<double> t(N + 1);
  int rc = mcedm_edm_t_steps(sp, t.data());
  if (rc) return rc;
  std::vector<double> gammas(N);
  for (int i = 0; i < N; ++i) {
    const bool in_range = sp->S_min <= t[i] && t[i] <= sp->S_max;                    // mcedm.py:606
    gammas[i] = in_range ? std::min(sp->S_churn / N, std::sqrt(2.0) - 1.0) : 0.0;
    MCEDM_REQUIRE(gammas[i] == 0.0 || step_noise != nullptr || rng_seed != nullptr, "heun_sample: S_churn > 0 needs step_noise (or mcedm_heun_sample_rng)");
  }
  Layout L;
  if ((rc = build_layout(P, B, H, W, 0, 1, &L))) return rc;
  const Header hd = header_for(P, B, H, W);
  const SamplerBufs sb = sampler_bufs(P, B, H, W);
  if (sb.total + hd.total + L.total_bytes > workspace_bytes) {
    set_error("heun_sample: workspace too small (%zu < %zu bytes)", workspace_bytes, sb.total + hd.total + L.total_bytes);
    return MCEDM_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const float* pk = (const float*)packed;
  double* x = at<double>(workspace, sb.x);
  double* xn = at<double>(workspace, sb.xn);
  double* dcur = at<double>(workspace, sb.d);
  float* x32 = at<float>(workspace, sb.x32);
  float* D = at<float>(workspace, sb.D);
  void* uws = at<char>(workspace, sb.total);
  const int C = P.desc.in_channels;
  const size_t hw = (size_t)H * W, total = (size_t)B * C * hw;
  const int Tout = return_last ? 1 : N + 1;
  const float w = (float)sp->w;
  const float sd = (float)sp->sigma_data;

  if ((rc = launch_heun_init(cond, P.desc.cond_channels, C, hw, mask, init_noise, t[0], total, x, x32, s))) return rc;
  if (!return_last && (rc = launch_heun_store(x, C, hw, 0, Tout, total, out, s))) return rc;
  for (int i = 0; i < N; ++i) {
    const double t_cur = t[i], t_next = t[i + 1];
    const double t_hat = t_cur + gammas[i] * t_cur;                                   // mcedm.py:607
    if (gammas[i] != 0.0) {
      const double c = std::sqrt(t_hat * t_hat - t_cur * t_cur) * sp->S_noise;
      if (step_noise) rc = launch_heun_churn(x, step_noise + (size_t)i * total, mask, c, total, x32, s);
      else rc = launch_heun_churn_rng(x, reinterpret_cast<const unsigned long long*>(rng_seed), (unsigned long long)i, c, total, x32, s, mask);
      if (rc) return rc;
    }
    // Euler step (mcedm.py:611-618); dx_cond: dx_in = get_dx_input(h, x_hat) first (ddim.py:1571)
    float* dxin = nullptr;
    if (dxc) {
      dxin = at<float>(workspace, sb.dxin);
      if ((rc = guidance_dx(P, *dxc, cond, x32, dxin, at<float>(workspace, sb.g), B, H, W, s))) return rc;
    }
    if ((rc = denoise_impl(P, L, hd, pk, x32, dxin, nullptr, (float)t_hat, 1, 1, cond, w, D, nullptr, uws, B, H, W, sd, s))) return rc;
    const float* dxg = nullptr;
    const float wgt = gd ? (float)gd->weight : 0.f;
    if (gd) {
      if ((rc = guidance_dx(P, *gd, cond, D, at<float>(workspace, sb.dx), at<float>(workspace, sb.g), B, H, W, s))) return rc;
      dxg = at<float>(workspace, sb.dx);
    }
    if ((rc = launch_heun_euler(x, D, mask, t_hat, t_next - t_hat, total, dcur, xn, x32, s, dxg, wgt, (float)t_hat))) return rc;
    // 2nd-order correction (mcedm.py:621-628)
    if (i < N - 1) {
      if (dxc && (rc = guidance_dx(P, *dxc, cond, x32, dxin, at<float>(workspace, sb.g), B, H, W, s))) return rc;   // on x_next (ddim.py:1584)
      if ((rc = denoise_impl(P, L, hd, pk, x32, dxin, nullptr, (float)t_next, 1, 1, cond, w, D, nullptr, uws, B, H, W, sd, s))) return rc;
      if (gd && (rc = guidance_dx(P, *gd, cond, D, at<float>(workspace, sb.dx), at<float>(workspace, sb.g), B, H, W, s))) return rc;
      if ((rc = launch_heun_correct(x, dcur, D, mask, t_next, t_next - t_hat, total, xn, x32, s, dxg, wgt, (float)t_hat))) return rc;
    }
    std::swap(x, xn);
    if (!return_last && (rc = launch_heun_store(x, C, hw, i + 1, Tout, total, out, s))) return rc;
  }
  if (return_last && (rc = launch_heun_store(x, C, hw, 0, 1, total, out, s))) return rc;
  return MCEDM_OK;
}

