// bwd.hpp -- launchers of the backward kernels (wgrad_mfma.hip, bwd_kernels.hip).
#pragma once
#include "common.hpp"

namespace mcedm {

// weight / bias gradient of the fused conv (same input description as ConvArgs)
struct WgradArgs {
  const float* dy;   // [B, Cout, H, W] gradient of the conv output
  const float* xa; const float* xb; int Ca, Cb;
  const Coef* coef; int coef_batch; int act; int resample;
  int Hs, Ws, H, W;
  int Cout, B;
  float* dwp;        // scratch: wgrad_scratch_floats(Cout, Cin, taps) floats
  float* dbp;        // set by the launcher (inside dwp)
};
size_t wgrad_scratch_floats(int Cout, int Cin, int taps);
// dw [Cout][Cin][kh][kw], db [Cout] (may be null).  qkv_heads > 0: dy rows are in packed qkv order.
// act_tmp: scratch of B * (Ca+Cb) * H * W floats for the materialised conv input, NOT the first bytes of an allocation (see
// wgrad_wino.hip); have_act: act_tmp already holds it (written
// by the GroupNorm backward of the same input, GnBwdArgs::xact: one pass over x less)
int launch_wgrad(const WgradArgs& a, int taps, float* dw, float* db, int qkv_heads, float* act_tmp, hipStream_t s,
                 bool have_act = false);
// wgrad_wino.hip: the un-resampled 3x3 weight gradient in Winograd F(3x3, 2x2) form (Cout, Cin multiples of 128 and W % 32 == 0, or multiples of 64 and W % 16 == 0).
// x: the materialised conv input [B][Cin][H][W]; the 16 bytes in FRONT of it must be readable (launch_wgrad passes act_tmp).
bool wgrad_wino_applicable(const WgradArgs& a, int taps, int qkv_heads);
size_t wgrad_wino_scratch_floats(int Cout, int Cin, int taps);     // 0 when the channel counts are not served
int launch_wgrad_wino(const WgradArgs& a, const float* x, float* dw, float* db, hipStream_t s);
// the 1x1 weight gradient as a GEMM on the same stage machinery (Cout, Cin multiples of 128, H * W % 64 == 0); its slices are in the
// direct kernel's scratch layout ([slice][co][ci], bias rows behind them at dbp): wgrad_reduce_kernel finishes both
size_t wgrad_gemm1_scratch_floats(int Cout, int Cin, int taps);
bool wgrad_gemm1_applicable(const WgradArgs& a, int taps, const float* x, bool in_place_concat);
int launch_wgrad_gemm1(const WgradArgs& a, const float* x, float* dbp, int* nslices, hipStream_t s);
void set_wgrad_wino(int enable);                                   // 1 / 0, -1: default (env MCEDM_WGRAD_WINO, else on)
// the materialisation step alone: out[B, Ca+Cb, H, W] = resample(act(coef(cat(xa, xb)))); dy / Cout / dwp are not read
int launch_act_materialize(const WgradArgs& a, float* out, hipStream_t s);

// backward through  act(film(group_norm(cat(xa,xb))))  followed by an optional 2x resampling
struct GnBwdArgs {
  const float* dact;   // gradient w.r.t. the conv input, [B, C, Hc, Wc] at the conv resolution
  int resample;        // forward Resample between the activation and the conv
  const float* xa; const float* xb; int Ca, Cb;
  int Hs, Ws;          // source (= x) spatial size
  int B, groups;
  const Coef* coef;    // forward transform rows [B][C]
  const float* stats;  // [B][groups][2] (mean, rstd)
  const float* gamma;  // [C]
  const float* film; int film_batch, film_stride;   // forward FiLM rows (scale | shift) or null
  int act;             // forward applied SiLU
  float* dxa; float* dxb;   // outputs [B, Ca, Hs, Ws], [B, Cb, Hs, Ws]
  int accumulate;      // 1: add to what dxa/dxb already hold
  const float* add;    // optional extra gradient to add, or null
  int add_mode;        // 1: [B, C, Hs, Ws] (source resolution); 2: conv resolution, mapped back like dact
  int add_C;           // channel count of `add` when add_mode == 2 (== C)
  float* ab;           // out [B][C][2]: sum(dt), sum(dt * xhat) per (sample, channel)
  float* xact;         // optional out [B, C, Hs, Ws]: act(coef(x)), the conv's (un-resampled) input for its weight gradient
  unsigned* sync;      // optional gn_bwd_sync_words(B, groups) 32-bit words, ZERO on entry (the counters are left zero on exit): lets
                       // the workgroups that share a (sample, group) slab keep their pieces in LDS between the passes and exchange
                       // their sums (gn_bwd_lds_kernel); null: the two-pass kernel
};
size_t gn_bwd_sync_words(int B, int groups);
int launch_gn_bwd(const GnBwdArgs& a, hipStream_t s);

// dgamma/dbeta (summed over the batch, overwritten) and the FiLM-row gradients from the ab table
int launch_gn_param_grads(const float* ab, const float* gamma, const float* beta, const float* film, int film_batch,
                          int film_stride, int B, int C, float* dgamma, float* dbeta, float* dfilm, int dfilm_stride,
                          hipStream_t s);

// torch.nn.GELU() of the dx_enc head (plan.hip): mode 0: out = GELU(v);  mode 1: out = g * dGELU(v)
int launch_gelu(const float* v, const float* g, float* out, size_t n, int mode, hipStream_t s);

// tiny dense helpers for the embedding MLP backward:  C[m][n] (+)= sum_k op(A)[m][k] * op(B)[k][n]
int launch_small_gemm(const float* A, const float* Bm, float* Cm, int M, int N, int K, int lda, int ldb, int ldc,
                      int transA, int transB, int accumulate, hipStream_t s);

// attention backward: qkv, dqkv in packed [B][heads][3][64][T]; a, da [B][heads*64][T]; lse scratch [B*heads][T][2]
int launch_attention_bwd(const float* qkv, const float* a, const float* da, float* dqkv, float* lse, int B, int heads,
                         int T, hipStream_t s);

}  // namespace mcedm
