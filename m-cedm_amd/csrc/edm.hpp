// edm.hpp -- launchers of the EDM preconditioning / Heun sampler kernels (edm.hip).
#pragma once
#include "common.hpp"

namespace mcedm {

int launch_precond_prepare(const float* sigma_dev, float sigma_host, int use_host, int n, float sigma_data,
                           int cond_ch, int in_ch, int dx_ch, float* coefs4, float* c_noise, Coef* conv_in_coef,
                           hipStream_t stream);
int launch_precond_finish(const float* x, const float* F, const float* Fu, float w, const float* coefs4, int n_sigma,
                          size_t per_sample, size_t total, float* D, float* F_out, hipStream_t stream);
int launch_heun_init(const float* cond, int cond_ch, int in_ch, size_t hw, const float* mask, const float* noise,
                     double t0, size_t total, double* x, float* x32, hipStream_t s);
int launch_heun_churn(double* x, const double* eps, const float* mask, double c, size_t total, float* x32,
                      hipStream_t s);
// x += c * N(0, 1) with the noise generated in the kernel (Philox4x32-10 keyed by the 64-bit seed at *seed_dev, counter =
// (element pair, draw)); launch_normal_fill writes the same draw out as a tensor
int launch_heun_churn_rng(double* x, const unsigned long long* seed_dev, unsigned long long draw, double c, size_t total, float* x32,
                          hipStream_t s, const float* mask = nullptr);
int launch_normal_fill(double* out, const unsigned long long* seed_dev, unsigned long long draw, size_t total, hipStream_t s);
// dxg / wgt / gdiv: optional PDE-guidance term of the single-task sampler, d -= (double)((wgt * dxg) / gdiv) formed in fp32
// (models/ddim.py:1577-1579, 1590-1591: `weight * dx / t_hat`, with t_hat in BOTH stages)
int launch_heun_euler(const double* x_hat, const float* D, const float* mask, double t_hat, double dt, size_t total,
                      double* d_cur, double* x_next, float* x32, hipStream_t s, const float* dxg = nullptr, float wgt = 0.f,
                      float gdiv = 1.f);
int launch_heun_correct(const double* x_hat, const double* d_cur, const float* D, const float* mask, double t_next,
                        double dt, size_t total, double* x_next, float* x32, hipStream_t s, const float* dxg = nullptr,
                        float wgt = 0.f, float gdiv = 1.f);
int launch_heun_store(const double* x, int C, size_t hw, int t, int T, size_t total, double* out, hipStream_t s);

// PDE guidance gradients (pde.hip): analytic adjoints of the residual stencils.  The two fields of the state may live in
// different tensors (the single-task models take h from the conditioning and u from the denoised state), so every field
// has its own base pointer and batch stride; row / column strides are shared.
struct GuideIO {
  const float* in[2]; const float* gt[2];     // state fields and the residual's target (SWE; usually the same tensors)
  long in_sb[2], st, sx;
  float* out[2]; long out_sb[2], out_st, out_sx;
  float sub[2], div[2];                       // un-normalisation x * div + sub applied while reading
  int mean;                                   // 1: write the mean of the two field gradients to out[0] only
};
int launch_swe_guidance(const GuideIO& io, int B, int T, int X, float half_dt, float dx, float scale2_h, float scale2_u,
                        hipStream_t s);
int launch_darcy_guidance(const GuideIO& io, float* scratch, int B, int S, float two_dx, int calc_prob, hipStream_t s);

// ---- RePaint-style sampler on the DDPM U-Net (models/ddim.py:915-1051): VP preconditioning and known-region kernels
// D = x + (-sigma) * F                                                   (get_denoised, ddim.py:923-946: c_skip 1, c_out -sigma)
int launch_vp_finish(const float* x, const float* F, float sigma, size_t total, float* D, hipStream_t s);
// conv_in transform rows of the VP network input: n_self zero/self-conditioning channels pass, the state is scaled by c_in
int launch_vp_coef(float c_in, int n_self, int n_in, Coef* out, hipStream_t s);
// x0 = ((hu*sa + nz*sb)*m + nz*(1-m)) [fp32] -> fp64, times t0        (ddim.py:989-994), m = 1 marks KNOWN entries
int launch_repaint_init(const float* hu, const float* noise, const float* mask, float sa, float sb, double t0, size_t total,
                        double* x, float* x32, hipStream_t s);
// x = (sa*hu + sb*nz)*m [fp32] + x*(1-m) [fp64]                          (ddim.py:1029-1031); final: x = hu*m + x*(1-m) (:1041-1043)
int launch_repaint_known(double* x, const float* hu, const float* noise, const float* mask, float sa, float sb, int final_clean,
                         size_t total, float* x32, hipStream_t s);

// ---- DDIM sampler with RePaint loops (PlDdim.sample_with_repeat, models/ddim.py:808-913), all fp32
// x0 = (xt - et * s1) / s0;  x0 = hu * m + x0 * (1 - m);  renoise: xt = s0 * x0 + s1 * et     (:873-882)
int launch_ddim_x0(float* xt, const float* et, const float* hu, const float* mask, float s0, float s1, int renoise, size_t total,
                   float* x0, hipStream_t s);
// xt_next = sa * x0 + [c1 * noise +] c2 * et;  xt_next = (sa * hu + c2 * hn) * m + xt_next * (1 - m)   (:884-895)
int launch_ddim_next(const float* x0, const float* et, const float* hu, const float* hn, const float* mask, const float* noise,
                     float sa, float c1, float c2, size_t total, float* xt, hipStream_t s);
// x = (hu * sa + hn * sb) * m + hn * (1 - m)   (:837-838)
int launch_ddim_init(const float* hu, const float* hn, const float* mask, float sa, float sb, size_t total, float* xt, hipStream_t s);
// 'b c h w -> b t h w c' for one time slot, fp32
int launch_store_f32(const float* x, int C, size_t hw, int t, int T, size_t total, float* out, hipStream_t s);

}  // namespace mcedm
