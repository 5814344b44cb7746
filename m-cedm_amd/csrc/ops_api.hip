// ops_api.hip -- kernel-level C entry points (tests, per-kernel timing).
#include "common.hpp"
#include "conv_wino.hpp"
#include "bwd.hpp"

using namespace mcedm;

static_assert(sizeof(mcedm_coef) == sizeof(Coef), "mcedm_coef must mirror Coef");

extern "C" size_t mcedm_op_conv_packed_floats(int Cout, int Cin, int k) {
  if (Cout <= 0 || Cin <= 0 || (k != 1 && k != 3)) return 0;
  return conv_packed_floats(Cout, Cin, k * k);
}

extern "C" int mcedm_op_pack_conv(const float* w, const float* b, int Cout, int Cin, int k, int qkv_heads, int dgrad,
                                  float* wpk, float* bias_pk, void* stream) {
  MCEDM_REQUIRE(w && wpk, "op_pack_conv: null pointer");
  MCEDM_REQUIRE(Cout > 0 && Cin > 0 && (k == 1 || k == 3), "op_pack_conv: bad shape");
  MCEDM_REQUIRE(qkv_heads == 0 || (Cout % (3 * qkv_heads) == 0 && !dgrad), "op_pack_conv: bad qkv_heads");
  int rc = dgrad ? launch_pack_conv(w, wpk, Cin, Cout, k * k, 0, 1, (hipStream_t)stream)
                 : launch_pack_conv(w, wpk, Cout, Cin, k * k, qkv_heads, 0, (hipStream_t)stream);
  if (rc) return rc;
  if (b && bias_pk) rc = launch_pack_bias(b, bias_pk, Cout, qkv_heads, (hipStream_t)stream);
  return rc;
}

extern "C" int mcedm_op_gn_coef(const float* xa, const float* xb, int Ca, int Cb, int B, int HW, const float* gamma,
                                const float* beta, const float* film, int film_batch, int film_stride, float eps,
                                mcedm_coef* coef_out, float* stats_out, void* stream) {
  MCEDM_REQUIRE(xa && gamma && beta && coef_out, "op_gn_coef: null pointer");
  MCEDM_REQUIRE(B > 0 && HW > 0 && Ca > 0 && Cb >= 0, "op_gn_coef: bad shape");
  const int C = Ca + Cb;
  MCEDM_REQUIRE(C >= 4, "op_gn_coef: C=%d < 4 gives zero groups (adm_blocks.py:89)", C);
  GnArgs a{xa, xb, Ca, Cb, HW, B, C / 4 < 32 ? C / 4 : 32, gamma, beta, film, film_batch, film_stride, eps,
           reinterpret_cast<Coef*>(coef_out), stats_out, nullptr, nullptr, SumTiles{}, SumTiles{}, 0};
  return launch_gn_coef(a, (hipStream_t)stream);
}

extern "C" int mcedm_op_conv(const float* xa, const float* xb, int Ca, int Cb, const mcedm_coef* coef, int coef_batch,
                             int act, int resample, int Hs, int Ws, int H, int W, const float* wpk,
                             const float* bias_pk, const float* res, int res_mode, float* out, int Cout, int B, int k,
                             void* stream) {
  MCEDM_REQUIRE(k == 1 || k == 3, "op_conv: k must be 1 or 3");
  MCEDM_REQUIRE(resample >= 0 && resample <= 3 && res_mode >= 0 && res_mode <= 2, "op_conv: bad resample mode");
  ConvArgs a{};
  a.xa = xa; a.xb = xb; a.Ca = Ca; a.Cb = Cb;
  a.coef = reinterpret_cast<const Coef*>(coef); a.coef_batch = coef_batch; a.act = act;
  a.resample = resample; a.Hs = Hs; a.Ws = Ws; a.H = H; a.W = W;
  a.wpk = wpk; a.bias = bias_pk; a.res = res; a.res_mode = res_mode;
  a.out = out; a.Cout = Cout; a.B = B;
  return launch_conv(a, k * k, (hipStream_t)stream);
}

extern "C" size_t mcedm_op_conv_wino_packed_floats(int Cout, int Cin) { return conv_wino_packed_floats(Cout, Cin); }

extern "C" int mcedm_op_pack_conv_wino(const float* w, int Cout, int Cin, float* wino, void* stream) {
  return launch_pack_conv_wino(w, wino, Cout, Cin, 0, (hipStream_t)stream);
}

extern "C" int mcedm_op_conv_wino(const float* xa, const float* xb, int Ca, int Cb, const mcedm_coef* coef, int coef_batch,
                                  int act, int resample, int H, int W, const float* wino, const float* bias, const float* res,
                                  int res_mode, float* out, int Cout, int B, void* stream) {
  MCEDM_REQUIRE((resample == RS_NONE || resample == RS_UP) && (res_mode == RS_NONE || res_mode == RS_UP || res_mode == RS_DOWN),
                "op_conv_wino: resample must be 0 (none) or 1 (nearest-2x up), res_mode 0, 1 or 2 (2x2-mean down)");
  ConvArgs a{};
  a.xa = xa; a.xb = xb; a.Ca = Ca; a.Cb = Cb;
  a.coef = reinterpret_cast<const Coef*>(coef); a.coef_batch = coef_batch; a.act = act;
  a.resample = resample; a.H = H; a.W = W;
  a.Hs = resample == RS_UP ? H / 2 : H; a.Ws = resample == RS_UP ? W / 2 : W;
  a.wino = wino; a.bias = bias; a.res = res; a.res_mode = res_mode;
  a.out = out; a.Cout = Cout; a.B = B;
  MCEDM_REQUIRE(conv_wino_applicable(a, 9), "op_conv_wino: needs Cout %% 64 == 0, H %% 8 == 0, W %% 16 == 0, Cin %% 8 == 0 and 16-byte aligned inputs / weight table");
  return launch_conv_wino(a, (hipStream_t)stream);
}

// freqs[k] = (1/10000)^(k/half) exactly as the plan packs them (adm_blocks.py:193-196, endpoint=False)
__global__ void op_freqs_kernel(float* f, int half) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < half) f[k] = powf(1.0f / 10000.0f, (float)k / (float)half);
}

extern "C" int mcedm_op_embedding(const float* labels, int n, int ch, const float* w0, const float* b0, const float* w1,
                                  const float* b1, const float* waff, const float* baff, int rows, float* freqs_scratch,
                                  float* emb_out, float* film_out, void* stream) {
  MCEDM_REQUIRE(labels && w0 && b0 && w1 && b1 && waff && baff && freqs_scratch && film_out, "op_embedding: null pointer");
  MCEDM_REQUIRE(n > 0 && ch > 0 && ch % 2 == 0 && rows > 0, "op_embedding: bad shape n=%d ch=%d rows=%d", n, ch, rows);
  hipLaunchKernelGGL(op_freqs_kernel, dim3(ceil_div(ch / 2, 64)), dim3(64), 0, (hipStream_t)stream, freqs_scratch, ch / 2);
  MCEDM_LAUNCH_CHECK("op_freqs_kernel");
  EmbArgs e{labels, n, ch, freqs_scratch, w0, b0, w1, b1, waff, baff, rows, emb_out, film_out};
  return launch_embedding(e, (hipStream_t)stream);
}

extern "C" int mcedm_op_attention(const float* qkv, float* out, int B, int heads, int T, void* stream) {
  MCEDM_REQUIRE(qkv && out, "op_attention: null pointer");
  return launch_attention(qkv, out, B, heads, T, (hipStream_t)stream);
}

extern "C" int mcedm_op_set_conv_tile(int mt, int ph, int pw) {
  set_conv_tile_override(mt, ph, pw);
  return MCEDM_OK;
}

extern "C" size_t mcedm_op_wgrad_scratch_floats(int Cout, int Cin, int k, int B, int H, int W) {
  if (Cout <= 0 || Cin <= 0 || (k != 1 && k != 3) || B <= 0 || H <= 0 || W <= 0) return 0;
  return align_up(wgrad_scratch_floats(Cout, Cin, k * k), 64) + (size_t)B * Cin * H * W;
}

extern "C" int mcedm_op_conv_wgrad(const float* dy, const float* xa, const float* xb, int Ca, int Cb,
                                   const mcedm_coef* coef, int coef_batch, int act, int resample, int Hs, int Ws, int H,
                                   int W, int Cout, int B, int k, int qkv_heads, float* scratch, float* dw, float* db,
                                   void* stream) {
  MCEDM_REQUIRE(k == 1 || k == 3, "op_conv_wgrad: k must be 1 or 3");
  MCEDM_REQUIRE(dy && scratch && dw, "op_conv_wgrad: null pointer");
  WgradArgs a{dy, xa, xb, Ca, Cb, reinterpret_cast<const Coef*>(coef), coef_batch, act, resample, Hs, Ws, H, W, Cout, B,
              scratch, nullptr};
  float* act_tmp = scratch + align_up(wgrad_scratch_floats(Cout, Ca + Cb, k * k), 64);
  return launch_wgrad(a, k * k, dw, db, qkv_heads, act_tmp, (hipStream_t)stream);
}

static int op_gn_bwd_impl(const float* dact, int resample, const float* xa, const float* xb, int Ca, int Cb, int Hs,
                               int Ws, int B, const mcedm_coef* coef, const float* stats, const float* gamma,
                               const float* beta, const float* film, int film_batch, int film_stride, int act,
                               float* dxa, float* dxb, int accumulate, const float* add, int add_mode, float* ab,
                               float* dgamma, float* dbeta, float* dfilm, int dfilm_stride, unsigned* sync, void* stream) {
  MCEDM_REQUIRE(dact && xa && coef && stats && gamma && beta && dxa && ab && dgamma && dbeta, "op_gn_bwd: null pointer");
  const int C = Ca + Cb;
  MCEDM_REQUIRE(C >= 4, "op_gn_bwd: C < 4");
  GnBwdArgs a{dact, resample, xa, xb, Ca, Cb, Hs, Ws, B, C / 4 < 32 ? C / 4 : 32, reinterpret_cast<const Coef*>(coef),
              stats, gamma, film, film_batch, film_stride, act, dxa, dxb, accumulate, add, add_mode, C, ab};
  a.sync = sync;
  int rc = launch_gn_bwd(a, (hipStream_t)stream);
  if (rc) return rc;
  return launch_gn_param_grads(ab, gamma, beta, film, film_batch, film_stride, B, C, dgamma, dbeta, dfilm, dfilm_stride,
                               (hipStream_t)stream);
}

extern "C" int mcedm_op_gn_bwd(const float* dact, int resample, const float* xa, const float* xb, int Ca, int Cb, int Hs,
                               int Ws, int B, const mcedm_coef* coef, const float* stats, const float* gamma,
                               const float* beta, const float* film, int film_batch, int film_stride, int act,
                               float* dxa, float* dxb, int accumulate, const float* add, int add_mode, float* ab,
                               float* dgamma, float* dbeta, float* dfilm, int dfilm_stride, void* stream) {
  return op_gn_bwd_impl(dact, resample, xa, xb, Ca, Cb, Hs, Ws, B, coef, stats, gamma, beta, film, film_batch, film_stride, act, dxa, dxb,
                        accumulate, add, add_mode, ab, dgamma, dbeta, dfilm, dfilm_stride, nullptr, stream);
}

extern "C" int mcedm_op_gn_bwd_sync(const float* dact, int resample, const float* xa, const float* xb, int Ca, int Cb, int Hs,
                                    int Ws, int B, const mcedm_coef* coef, const float* stats, const float* gamma,
                                    const float* beta, const float* film, int film_batch, int film_stride, int act,
                                    float* dxa, float* dxb, int accumulate, const float* add, int add_mode, float* ab,
                                    float* dgamma, float* dbeta, float* dfilm, int dfilm_stride, unsigned int* sync, void* stream) {
  return op_gn_bwd_impl(dact, resample, xa, xb, Ca, Cb, Hs, Ws, B, coef, stats, gamma, beta, film, film_batch, film_stride, act, dxa, dxb,
                        accumulate, add, add_mode, ab, dgamma, dbeta, dfilm, dfilm_stride, sync, stream);
}

extern "C" int mcedm_op_attention_bwd(const float* qkv, const float* a, const float* da, float* dqkv, float* lse_scratch,
                                      int B, int heads, int T, void* stream) {
  return launch_attention_bwd(qkv, a, da, dqkv, lse_scratch, B, heads, T, (hipStream_t)stream);
}

extern "C" int mcedm_op_set_conv8(int enable) {
  set_conv8(enable);
  return MCEDM_OK;
}

extern "C" int mcedm_op_set_conv_wino(int enable) {
  set_conv_wino(enable);
  return MCEDM_OK;
}

extern "C" int mcedm_op_set_conv_wino1(int enable) {
  set_conv_wino1(enable);
  return MCEDM_OK;
}

extern "C" int mcedm_op_set_conv1x1_reg(int enable) {
  set_conv1x1_reg(enable);
  return MCEDM_OK;
}

extern "C" int mcedm_op_set_wgrad_wino(int enable) {
  set_wgrad_wino(enable);
  return MCEDM_OK;
}

extern "C" int mcedm_op_set_conv_resident(int enable) {
  set_conv_resident(enable);
  return MCEDM_OK;
}

extern "C" int mcedm_op_set_attn_fused(int enable) {
  set_attn_fused(enable);
  return MCEDM_OK;
}

extern "C" int mcedm_op_set_conv_debug(unsigned long long* buf) {
  set_conv_debug(buf);
  return MCEDM_OK;
}
