// edm.hip -- K7 (EDM preconditioning), K8 (Heun sampler state updates, fp64), K9 (masked EDM
// loss + its gradient), K11 (grad-clip + Adam + EMA).  All HBM-bound elementwise / reduction
// kernels: one pass over the tensors, coalesced, grid-stride.
//
// The fp64 sampler arithmetic mirrors the evaluation order of models/mcedm.py:594-628 term by
// term, so this file is compiled with -ffp-contract=off (see build.py).
#include "common.hpp"
#include "edm.hpp"

namespace mcedm {

static inline int grid_for(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b < 2048 ? (b ? b : 1) : 2048);
}

// ---- K7a: per-sample EDM coefficients (mcedm.py:203-206 / 448-451), fp32 like the reference.
// Writes c_skip/c_out/c_in/c_noise rows and the conv_in transform table: cond channels pass
// through, state channels are scaled by c_in (mcedm.py:208: only x is scaled, not cond).
__global__ void precond_prepare_kernel(const float* __restrict__ sigma_dev, float sigma_host, int use_host, int n,
                                       float sigma_data, int cond_ch, int in_ch, int dx_ch, float* __restrict__ coefs4,
                                       float* __restrict__ c_noise, Coef* __restrict__ conv_in_coef) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = use_host ? sigma_host : sigma_dev[i];
  const float sd2 = sigma_data * sigma_data;
  const float s2 = s * s;
  const float c_skip = sd2 / (s2 + sd2);
  const float c_out = s * sigma_data / sqrtf(s2 + sd2);
  const float c_in = 1.0f / sqrtf(sd2 + s2);
  const float cn = logf(s) / 4.0f;
  coefs4[4 * i + 0] = c_skip;
  coefs4[4 * i + 1] = c_out;
  coefs4[4 * i + 2] = c_in;
  coefs4[4 * i + 3] = cn;
  c_noise[i] = cn;
  // rows of cat(cond, x, dx): only x is scaled by c_in (adm_blocks.py:319-340 is handed c_in * x, mcedm.py:208)
  const int Ct = cond_ch + in_ch + dx_ch;
  for (int c = 0; c < Ct; ++c)
    conv_in_coef[(size_t)i * Ct + c] = Coef{0.f, (c < cond_ch || c >= cond_ch + in_ch) ? 1.0f : c_in, 0.f, 0.f};
}

int launch_precond_prepare(const float* sigma_dev, float sigma_host, int use_host, int n, float sigma_data,
                           int cond_ch, int in_ch, int dx_ch, float* coefs4, float* c_noise, Coef* conv_in_coef,
                           hipStream_t stream) {
  hipLaunchKernelGGL(precond_prepare_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, stream, sigma_dev, sigma_host,
                     use_host, n, sigma_data, cond_ch, in_ch, dx_ch, coefs4, c_noise, conv_in_coef);
  MCEDM_LAUNCH_CHECK("precond_prepare_kernel");
  return MCEDM_OK;
}

// ---- K7b: D = c_skip*x + c_out*F  (mcedm.py:210 / 460); optional classifier-free blend
// F = (w+1)*F - w*F_uncond first (mcedm.py:457-458).
__global__ void precond_finish_kernel(const float* __restrict__ x, const float* __restrict__ F,
                                      const float* __restrict__ Fu, float w, const float* __restrict__ coefs4,
                                      int n_sigma, size_t per_sample, size_t total, float* __restrict__ D,
                                      float* __restrict__ F_out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = n_sigma == 1 ? 0 : i / per_sample;
    float f = F[i];
    if (Fu) f = (w + 1.0f) * f - w * Fu[i];
    if (F_out) F_out[i] = f;
    D[i] = coefs4[4 * b] * x[i] + coefs4[4 * b + 1] * f;
  }
}

int launch_precond_finish(const float* x, const float* F, const float* Fu, float w, const float* coefs4, int n_sigma,
                          size_t per_sample, size_t total, float* D, float* F_out, hipStream_t stream) {
  hipLaunchKernelGGL(precond_finish_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, F, Fu, w, coefs4, n_sigma,
                     per_sample, total, D, F_out);
  MCEDM_LAUNCH_CHECK("precond_finish_kernel");
  return MCEDM_OK;
}

// ---- K8: sampler state (fp64) ---------------------------------------------------------------
// x0 = hu_known*(1-mask) + (noise*t0)*mask                      (mcedm.py:594-597)
__global__ void heun_init_kernel(const float* __restrict__ cond, int cond_ch, int in_ch, size_t hw,
                                 const float* __restrict__ mask, const float* __restrict__ noise, double t0,
                                 size_t total, double* __restrict__ x, float* __restrict__ x32) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / (in_ch * hw), r = i % (in_ch * hw);
    // mask == NULL: the unmasked sampler of PlCondEdm (models/ddim.py:1556): x0 = noise * t0 (0 + v and v * 1 are exact)
    const float m = mask ? mask[i] : 1.0f;
    const float known = mask ? cond[b * cond_ch * hw + r] : 0.0f;          // cond[:, 0:in_ch]
    const float keep = known * (1.0f - m);                   // fp32 product like the reference
    const double v = (double)keep + ((double)noise[i] * t0) * (double)m;
    x[i] = v;
    x32[i] = (float)v;
  }
}

// x_hat = x_cur + (c*eps)*mask, c = sqrt(t_hat^2 - t_cur^2)*S_noise   (mcedm.py:608)
__global__ void heun_churn_kernel(double* __restrict__ x, const double* __restrict__ eps, const float* __restrict__ mask,
                                  double c, size_t total, float* __restrict__ x32) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const double v = x[i] + (c * eps[i]) * (mask ? (double)mask[i] : 1.0);
    x[i] = v;
    x32[i] = (float)v;
  }
}

// d = (x_hat - D)/t_hat ; x_next = x_hat + ((t_next - t_hat)*d)*mask    (mcedm.py:617-618)
// dxg != NULL: PDE guidance of the single-task sampler, d = (x_hat - D)/t_hat - weight * dx / t_hat with the guidance
// term formed in fp32 (models/ddim.py:1577-1579)
__global__ void heun_euler_kernel(const double* __restrict__ x_hat, const float* __restrict__ D,
                                  const float* __restrict__ mask, double t_hat, double dt, size_t total,
                                  double* __restrict__ d_cur, double* __restrict__ x_next, float* __restrict__ x32,
                                  const float* __restrict__ dxg, float wgt, float gdiv) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const double xh = x_hat[i];
    double d = (xh - (double)D[i]) / t_hat;
    if (dxg) d = d - (double)((wgt * dxg[i]) / gdiv);
    const double v = xh + (dt * d) * (mask ? (double)mask[i] : 1.0);
    d_cur[i] = d;
    x_next[i] = v;
    x32[i] = (float)v;
  }
}

// d' = (x_next - D')/t_next ; x_next = x_hat + (dt*(0.5 d + 0.5 d'))*mask   (mcedm.py:627-628)
__global__ void heun_correct_kernel(const double* __restrict__ x_hat, const double* __restrict__ d_cur,
                                    const float* __restrict__ D, const float* __restrict__ mask, double t_next, double dt,
                                    size_t total, double* __restrict__ x_next, float* __restrict__ x32,
                                    const float* __restrict__ dxg, float wgt, float gdiv) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    double dp = (x_next[i] - (double)D[i]) / t_next;
    if (dxg) dp = dp - (double)((wgt * dxg[i]) / gdiv);                 // models/ddim.py:1590-1591 (divides by t_hat here too)
    const double v = x_hat[i] + (dt * (0.5 * d_cur[i] + 0.5 * dp)) * (mask ? (double)mask[i] : 1.0);
    x_next[i] = v;
    x32[i] = (float)v;
  }
}

// 'b c h w -> b t h w c' for one time slot (mcedm.py:636-638)
__global__ void heun_store_kernel(const double* __restrict__ x, int C, size_t hw, int t, int T, size_t total,
                                  double* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i enumerates the OUTPUT (b, p, c) so stores are contiguous
    const size_t c = i % C;
    const size_t p = (i / C) % hw;
    const size_t b = i / (C * hw);
    out[((b * T + t) * hw + p) * C + c] = x[(b * C + c) * hw + p];
  }
}

int launch_heun_init(const float* cond, int cond_ch, int in_ch, size_t hw, const float* mask, const float* noise,
                     double t0, size_t total, double* x, float* x32, hipStream_t s) {
  hipLaunchKernelGGL(heun_init_kernel, dim3(grid_for(total)), dim3(256), 0, s, cond, cond_ch, in_ch, hw, mask, noise, t0,
                     total, x, x32);
  MCEDM_LAUNCH_CHECK("heun_init_kernel");
  return MCEDM_OK;
}
int launch_heun_churn(double* x, const double* eps, const float* mask, double c, size_t total, float* x32, hipStream_t s) {
  hipLaunchKernelGGL(heun_churn_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, eps, mask, c, total, x32);
  MCEDM_LAUNCH_CHECK("heun_churn_kernel");
  return MCEDM_OK;
}
int launch_heun_euler(const double* x_hat, const float* D, const float* mask, double t_hat, double dt, size_t total,
                      double* d_cur, double* x_next, float* x32, hipStream_t s, const float* dxg, float wgt, float gdiv) {
  hipLaunchKernelGGL(heun_euler_kernel, dim3(grid_for(total)), dim3(256), 0, s, x_hat, D, mask, t_hat, dt, total, d_cur,
                     x_next, x32, dxg, wgt, gdiv);
  MCEDM_LAUNCH_CHECK("heun_euler_kernel");
  return MCEDM_OK;
}
int launch_heun_correct(const double* x_hat, const double* d_cur, const float* D, const float* mask, double t_next,
                        double dt, size_t total, double* x_next, float* x32, hipStream_t s, const float* dxg, float wgt,
                        float gdiv) {
  hipLaunchKernelGGL(heun_correct_kernel, dim3(grid_for(total)), dim3(256), 0, s, x_hat, d_cur, D, mask, t_next, dt,
                     total, x_next, x32, dxg, wgt, gdiv);
  MCEDM_LAUNCH_CHECK("heun_correct_kernel");
  return MCEDM_OK;
}
int launch_heun_store(const double* x, int C, size_t hw, int t, int T, size_t total, double* out, hipStream_t s) {
  hipLaunchKernelGGL(heun_store_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, C, hw, t, T, total, out);
  MCEDM_LAUNCH_CHECK("heun_store_kernel");
  return MCEDM_OK;
}

// ---- counter-based normal noise on the device (Philox4x32-10, Salmon et al. 2011) -----------------------------------
// The reference draws one randn_like tensor per Heun step and per resampling loop (models/ddim.py:1004, 1037); at
// BASELINE config 5 (18 steps x 32 loops) that is 576 fp64 tensors per call.  Instead of materialising them, the
// re-noising kernel generates its own: counter = (element pair index, draw index), key = a 64-bit seed read from DEVICE
// memory (so a captured HIP graph replays with fresh noise after the host bumps the seed).  Statistically equivalent to,
// but not the same stream as, torch's generator.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
// two independent N(0, 1) doubles from 128 random bits (Box-Muller on 53-bit uniforms; u1 in (0, 1])
__device__ __forceinline__ void normal_pair(const unsigned (&c)[4], double& z0, double& z1) {
  const double u1 = ((double)((((unsigned long long)c[0] << 32) | c[1]) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
  const double u2 = (double)((((unsigned long long)c[2] << 32) | c[3]) >> 11) * (1.0 / 9007199254740992.0);
  const double r = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincos(6.283185307179586476925 * u2, &sn, &cs);
  z0 = r * cs; z1 = r * sn;
}
// mask (optional): x_hat = x_cur + (c * eps) * mask, the joint model's churn (models/mcedm.py:608); NULL multiplies by 1.0 exactly
__global__ void heun_churn_rng_kernel(double* __restrict__ x, const unsigned long long* __restrict__ seed_dev,
                                      unsigned long long draw, double c, size_t total, float* __restrict__ x32,
                                      const float* __restrict__ mask) {
  const unsigned long long seed = *seed_dev;
  const size_t pairs = (total + 1) / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
    unsigned ctr[4] = {(unsigned)i, (unsigned)((unsigned long long)i >> 32), (unsigned)draw, (unsigned)(draw >> 32)};
    philox4x32_10(ctr, (unsigned)seed, (unsigned)(seed >> 32));
    double z0, z1;
    normal_pair(ctr, z0, z1);
    const double m0 = mask ? (double)mask[2 * i] : 1.0;
    const double v0 = x[2 * i] + (c * z0) * m0;
    x[2 * i] = v0; x32[2 * i] = (float)v0;
    if (2 * i + 1 < total) {
      const double m1 = mask ? (double)mask[2 * i + 1] : 1.0;
      const double v1 = x[2 * i + 1] + (c * z1) * m1;
      x[2 * i + 1] = v1; x32[2 * i + 1] = (float)v1;
    }
  }
}
int launch_heun_churn_rng(double* x, const unsigned long long* seed_dev, unsigned long long draw, double c, size_t total, float* x32,
                          hipStream_t s, const float* mask) {
  hipLaunchKernelGGL(heun_churn_rng_kernel, dim3(grid_for((total + 1) / 2)), dim3(256), 0, s, x, seed_dev, draw, c, total, x32, mask);
  MCEDM_LAUNCH_CHECK("heun_churn_rng_kernel");
  return MCEDM_OK;
}
// out[i] = N(0, 1) of draw `draw` (the generator on its own: tests, and callers that want the tensor)
__global__ void normal_fill_kernel(double* __restrict__ out, const unsigned long long* __restrict__ seed_dev, unsigned long long draw,
                                   size_t total) {
  const unsigned long long seed = *seed_dev;
  const size_t pairs = (total + 1) / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
    unsigned ctr[4] = {(unsigned)i, (unsigned)((unsigned long long)i >> 32), (unsigned)draw, (unsigned)(draw >> 32)};
    philox4x32_10(ctr, (unsigned)seed, (unsigned)(seed >> 32));
    double z0, z1;
    normal_pair(ctr, z0, z1);
    out[2 * i] = z0;
    if (2 * i + 1 < total) out[2 * i + 1] = z1;
  }
}
int launch_normal_fill(double* out, const unsigned long long* seed_dev, unsigned long long draw, size_t total, hipStream_t s) {
  hipLaunchKernelGGL(normal_fill_kernel, dim3(grid_for((total + 1) / 2)), dim3(256), 0, s, out, seed_dev, draw, total);
  MCEDM_LAUNCH_CHECK("normal_fill_kernel");
  return MCEDM_OK;
}

// ---- RePaint-style EDM sampling on the DDPM U-Net (PlDdim, models/ddim.py:915-1051) -------------------------------
__global__ void vp_finish_kernel(const float* __restrict__ x, const float* __restrict__ F, float c_out, size_t total,
                                 float* __restrict__ D) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    D[i] = 1.0f * x[i] + c_out * F[i];                                // c_skip * xt + c_out * F_x, ddim.py:945
}
int launch_vp_finish(const float* x, const float* F, float sigma, size_t total, float* D, hipStream_t s) {
  hipLaunchKernelGGL(vp_finish_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, F, -sigma, total, D);
  MCEDM_LAUNCH_CHECK("vp_finish_kernel");
  return MCEDM_OK;
}

__global__ void vp_coef_kernel(float c_in, int n_self, int n_in, Coef* __restrict__ out) {
  const int c = threadIdx.x;
  if (c < n_self + n_in) out[c] = Coef{0.f, c < n_self ? 1.0f : c_in, 0.f, 0.f};
}
int launch_vp_coef(float c_in, int n_self, int n_in, Coef* out, hipStream_t s) {
  MCEDM_REQUIRE(n_self + n_in <= 64, "vp_coef: too many input channels");
  hipLaunchKernelGGL(vp_coef_kernel, dim3(1), dim3(64), 0, s, c_in, n_self, n_in, out);
  MCEDM_LAUNCH_CHECK("vp_coef_kernel");
  return MCEDM_OK;
}

__global__ void repaint_init_kernel(const float* __restrict__ hu, const float* __restrict__ noise,
                                    const float* __restrict__ mask, float sa, float sb, double t0, size_t total,
                                    double* __restrict__ x, float* __restrict__ x32) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float m = mask[i], nz = noise[i];
    const float known = hu[i] * sa + nz * sb;                          // hu * aT.sqrt() + hu_noise * (1 - aT).sqrt(), fp32
    const float v32 = known * m + nz * (1.0f - m);
    const double v = (double)v32 * t0;
    x[i] = v;
    x32[i] = (float)v;
  }
}
int launch_repaint_init(const float* hu, const float* noise, const float* mask, float sa, float sb, double t0, size_t total,
                        double* x, float* x32, hipStream_t s) {
  hipLaunchKernelGGL(repaint_init_kernel, dim3(grid_for(total)), dim3(256), 0, s, hu, noise, mask, sa, sb, t0, total, x, x32);
  MCEDM_LAUNCH_CHECK("repaint_init_kernel");
  return MCEDM_OK;
}

__global__ void repaint_known_kernel(double* __restrict__ x, const float* __restrict__ hu, const float* __restrict__ noise,
                                     const float* __restrict__ mask, float sa, float sb, int final_clean, size_t total,
                                     float* __restrict__ x32) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float m = mask[i];
    // at_next.sqrt() * hu + (1 - at_next).sqrt() * hu_noise in fp32; the final replacement uses the clean hu
    const float known = final_clean ? hu[i] : sa * hu[i] + sb * noise[i];
    const float km = known * m;                                        // fp32 * fp32
    const double v = (double)km + x[i] * (double)(1.0f - m);           // fp64 * fp32 promotes to fp64
    x[i] = v;
    x32[i] = (float)v;
  }
}
int launch_repaint_known(double* x, const float* hu, const float* noise, const float* mask, float sa, float sb, int final_clean,
                         size_t total, float* x32, hipStream_t s) {
  hipLaunchKernelGGL(repaint_known_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, hu, noise, mask, sa, sb, final_clean,
                     total, x32);
  MCEDM_LAUNCH_CHECK("repaint_known_kernel");
  return MCEDM_OK;
}

// ---- DDIM sampler with RePaint loops (PlDdim.sample_with_repeat, models/ddim.py:808-913): fp32 like the reference ----
__global__ void ddim_x0_kernel(float* __restrict__ xt, const float* __restrict__ et, const float* __restrict__ hu,
                               const float* __restrict__ mask, float s0, float s1, int renoise, size_t total,
                               float* __restrict__ x0) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float e = et[i], m = mask[i];
    float v = (xt[i] - e * s1) / s0;                                   // (xt - et * (1 - at).sqrt()) / at.sqrt()
    v = hu[i] * m + v * (1.0f - m);                                    // add known part (:878-879)
    x0[i] = v;
    if (renoise) xt[i] = s0 * v + s1 * e;                              // at.sqrt() * x0_t + (1 - at).sqrt() * et (:881-882)
  }
}
int launch_ddim_x0(float* xt, const float* et, const float* hu, const float* mask, float s0, float s1, int renoise, size_t total,
                   float* x0, hipStream_t s) {
  hipLaunchKernelGGL(ddim_x0_kernel, dim3(grid_for(total)), dim3(256), 0, s, xt, et, hu, mask, s0, s1, renoise, total, x0);
  MCEDM_LAUNCH_CHECK("ddim_x0_kernel");
  return MCEDM_OK;
}
__global__ void ddim_next_kernel(const float* __restrict__ x0, const float* __restrict__ et, const float* __restrict__ hu,
                                 const float* __restrict__ hn, const float* __restrict__ mask, const float* __restrict__ noise,
                                 float sa, float c1, float c2, size_t total, float* __restrict__ xt) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float m = mask[i];
    float v = noise ? (sa * x0[i] + c1 * noise[i]) + c2 * et[i] : sa * x0[i] + c2 * et[i];      // (:891-895)
    const float known = sa * hu[i] + c2 * hn[i];                                                // at_next.sqrt() * hu + c2 * hu_noise
    xt[i] = known * m + v * (1.0f - m);
  }
}
int launch_ddim_next(const float* x0, const float* et, const float* hu, const float* hn, const float* mask, const float* noise,
                     float sa, float c1, float c2, size_t total, float* xt, hipStream_t s) {
  hipLaunchKernelGGL(ddim_next_kernel, dim3(grid_for(total)), dim3(256), 0, s, x0, et, hu, hn, mask, noise, sa, c1, c2, total, xt);
  MCEDM_LAUNCH_CHECK("ddim_next_kernel");
  return MCEDM_OK;
}
__global__ void ddim_init_kernel(const float* __restrict__ hu, const float* __restrict__ hn, const float* __restrict__ mask,
                                 float sa, float sb, size_t total, float* __restrict__ xt) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float m = mask[i], nz = hn[i];
    xt[i] = (hu[i] * sa + nz * sb) * m + nz * (1.0f - m);             // ddim.py:837-838
  }
}
int launch_ddim_init(const float* hu, const float* hn, const float* mask, float sa, float sb, size_t total, float* xt, hipStream_t s) {
  hipLaunchKernelGGL(ddim_init_kernel, dim3(grid_for(total)), dim3(256), 0, s, hu, hn, mask, sa, sb, total, xt);
  MCEDM_LAUNCH_CHECK("ddim_init_kernel");
  return MCEDM_OK;
}
__global__ void store_f32_kernel(const float* __restrict__ x, int C, size_t hw, int t, int T, size_t total, float* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t c = i % C, p = (i / C) % hw, b = i / (C * hw);
    out[((b * T + t) * hw + p) * C + c] = x[(b * C + c) * hw + p];
  }
}
int launch_store_f32(const float* x, int C, size_t hw, int t, int T, size_t total, float* out, hipStream_t s) {
  hipLaunchKernelGGL(store_f32_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, C, hw, t, T, total, out);
  MCEDM_LAUNCH_CHECK("store_f32_kernel");
  return MCEDM_OK;
}

// ---- training-side elementwise ----------------------------------------------------------------
// sigma = exp(rnd*P_std + P_mean) (mcedm.py:271); x_noise = x + mask*noise*sigma (mcedm.py:216)
__global__ void noise_inputs_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                    const float* __restrict__ noise, const float* __restrict__ rnd, float P_mean,
                                    float P_std, size_t per_sample, size_t total, float* __restrict__ x_noise,
                                    float* __restrict__ sigma_out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / per_sample;
    const float sigma = expf(rnd[b] * P_std + P_mean);
    if (i % per_sample == 0) sigma_out[b] = sigma;
    x_noise[i] = mask ? x[i] + mask[i] * noise[i] * sigma : x[i] + noise[i] * sigma;      // mcedm.py:216-218
  }
}

// ---- deterministic grid-wide sums (loss, gradient norm) ---------------------------------------------------------
// Every block stores its partial sum (fp64) in a slot of a scratch array, takes a ticket, and the block that draws the LAST
// ticket adds the slots in index order: the value does not depend on the order the blocks ran in (an atomicAdd of the
// partials would make the loss, the clip factor and with them the whole training run differ in the last bits from run to
// run).  The array and the ticket live in CALLER-owned scratch (ABI 3: MCEDM_REDUCE_SCRATCH_BYTES per concurrent call), so
// two plans on two streams of one device never share state; the entry point zeroes the ticket on the stream in front of
// the kernel.  Layout: [RED_MAX] fp64 partials, then the 32-bit ticket.
constexpr int RED_MAX = 4096;
static_assert(MCEDM_REDUCE_SCRATCH_BYTES >= RED_MAX * 8 + 4, "mcedm_hip.h: MCEDM_REDUCE_SCRATCH_BYTES too small for the reduction");
struct RedScratch { double* part; unsigned* ticket; };
static inline RedScratch red_scratch(void* scratch) {
  return RedScratch{reinterpret_cast<double*>(scratch), reinterpret_cast<unsigned*>(reinterpret_cast<char*>(scratch) + (size_t)RED_MAX * 8)};
}

// block_sum: this block's partial in thread 0.  Returns true in every thread of the LAST block, with the total in *total
// (thread 0 only).
__device__ __forceinline__ bool grid_sum_fixed_order(RedScratch rs, double block_sum, unsigned bid, unsigned nblocks, double* total) {
  __shared__ unsigned s_last;
  __shared__ double s_red[4];
  if (threadIdx.x == 0) {
    __hip_atomic_store(&rs.part[bid], block_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = (atomicAdd(rs.ticket, 1u) == nblocks - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return false;
  __threadfence();
  double t = 0.0;
  for (unsigned i = threadIdx.x; i < nblocks; i += blockDim.x)      // fixed assignment of slots to threads
    t += __hip_atomic_load(&rs.part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    *total = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    *rs.ticket = 0u;
  }
  return true;
}

// loss = mean_b sum_chw w_b*(D*m - x*m)^2 ; dD = (2 w_b / B) * (D*m - x*m) * m   (mcedm.py:278, losses.py:48-53)
__global__ __launch_bounds__(256) void edm_loss_kernel(const float* __restrict__ D, const float* __restrict__ x,
                                                       const float* __restrict__ mask, const float* __restrict__ sigma,
                                                       float sigma_data, int B, size_t per_sample,
                                                       float* __restrict__ loss, float* __restrict__ dD, RedScratch rs) {
  const int b = blockIdx.y;
  const float s = sigma[b];
  const float wgt = (s * s + sigma_data * sigma_data) / ((s * sigma_data) * (s * sigma_data));
  const size_t base = (size_t)b * per_sample;
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per_sample; i += (size_t)gridDim.x * blockDim.x) {
    const float m = mask ? mask[base + i] : 1.0f;
    const float diff = D[base + i] * m - x[base + i] * m;
    acc += wgt * (diff * diff);
    if (dD) dD[base + i] = (2.0f * wgt / (float)B) * diff * m;
  }
  double t = acc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  double total;
  if (grid_sum_fixed_order(rs, (red[0] + red[1]) + (red[2] + red[3]), blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, &total) &&
      threadIdx.x == 0)
    *loss = (float)(total / (double)B);
}

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, size_t n, double* __restrict__ out, RedScratch rs) {
  float a0 = 0.f, a1 = 0.f;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i + stride < n; i += 2 * stride) {
    const float u = g[i], v = g[i + stride];
    a0 += u * u; a1 += v * v;
  }
  if (i < n) { const float u = g[i]; a0 += u * u; }
  double t = (double)a0 + (double)a1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  double total;
  if (grid_sum_fixed_order(rs, (red[0] + red[1]) + (red[2] + red[3]), blockIdx.x, gridDim.x, &total) && threadIdx.x == 0) *out = total;
}

// torch.optim.Adam (no amsgrad) on clipped grads, then EmaModel.update (ddim_blocks.py:44-54).
__global__ void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, float* __restrict__ ema, size_t n, float step_size, float w1,
                                float b2, float w2, float eps, float wd, const double* __restrict__ sqnorm,
                                double max_norm, float gscale, float ema_beta, float ema_w, float bc2_sqrt) {
  float clip = 1.f;
  if (sqnorm) {
    // clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1; norm of the (scaled) grads
    const double total = sqrt(*sqnorm) * (double)gscale;
    const double c = max_norm / (total + 1e-6);
    clip = (float)(c < 1.0 ? c : 1.0);
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale * clip;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    // exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1-b2); addcdiv_(m, denom, -step_size)
    const float m0 = m[i];
    const float mi = m0 + w1 * (gi - m0);
    const float vi = v[i] * b2 + w2 * (gi * gi);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pn = pi - step_size * (mi / denom);
    m[i] = mi; v[i] = vi; p[i] = pn;
    if (ema) ema[i] = ema[i] * ema_beta + ema_w * pn;
  }
}

}  // namespace mcedm

using namespace mcedm;

extern "C" int mcedm_edm_noise_inputs(const float* x, const float* mask, const float* noise, const float* rnd_normal,
                                      int B, int C, int H, int W, double P_mean, double P_std, float* x_noise,
                                      float* sigma_out, void* stream) {
  MCEDM_REQUIRE(x && noise && rnd_normal && x_noise && sigma_out, "noise_inputs: null pointer");
  MCEDM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "noise_inputs: empty shape");
  const size_t per = (size_t)C * H * W, total = per * B;
  hipLaunchKernelGGL(noise_inputs_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, mask, noise,
                     rnd_normal, (float)P_mean, (float)P_std, per, total, x_noise, sigma_out);
  MCEDM_LAUNCH_CHECK("noise_inputs_kernel");
  return MCEDM_OK;
}

extern "C" int mcedm_edm_loss(const float* D, const float* x, const float* mask, const float* sigma, int B, int C, int H,
                              int W, double sigma_data, float* loss_out, float* dD_out, void* scratch, size_t scratch_bytes,
                              void* stream) {
  MCEDM_REQUIRE(D && x && sigma && loss_out, "edm_loss: null pointer");
  MCEDM_REQUIRE(scratch && scratch_bytes >= MCEDM_REDUCE_SCRATCH_BYTES && ((size_t)scratch & 7) == 0,
                "edm_loss: needs %d bytes of 8-byte aligned device scratch (MCEDM_REDUCE_SCRATCH_BYTES)", MCEDM_REDUCE_SCRATCH_BYTES);
  MCEDM_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0, "edm_loss: bad shape");
  const size_t per = (size_t)C * H * W;
  MCEDM_REQUIRE(B <= RED_MAX, "edm_loss: batch %d exceeds the reduction table (%d)", B, RED_MAX);
  int gx = (int)((per + 255) / 256);
  if (gx > 64) gx = 64;
  if (gx > RED_MAX / B) gx = RED_MAX / B;            // one slot of the fixed-order reduction per block
  const RedScratch rs = red_scratch(scratch);
  MCEDM_HIP_TRY(hipMemsetAsync(rs.ticket, 0, sizeof(unsigned), (hipStream_t)stream));
  hipLaunchKernelGGL(edm_loss_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, D, x, mask, sigma,
                     (float)sigma_data, B, per, loss_out, dD_out, rs);
  MCEDM_LAUNCH_CHECK("edm_loss_kernel");
  return MCEDM_OK;
}

extern "C" int mcedm_sqnorm(const float* g, size_t n, double* sqnorm_out, void* scratch, size_t scratch_bytes, void* stream) {
  MCEDM_REQUIRE(g && sqnorm_out, "sqnorm: null pointer");
  MCEDM_REQUIRE(scratch && scratch_bytes >= MCEDM_REDUCE_SCRATCH_BYTES && ((size_t)scratch & 7) == 0,
                "sqnorm: needs %d bytes of 8-byte aligned device scratch (MCEDM_REDUCE_SCRATCH_BYTES)", MCEDM_REDUCE_SCRATCH_BYTES);
  if (n == 0) { MCEDM_HIP_TRY(hipMemsetAsync(sqnorm_out, 0, sizeof(double), (hipStream_t)stream)); return MCEDM_OK; }
  int blocks = grid_for(n);
  if (blocks > RED_MAX) blocks = RED_MAX;
  const RedScratch rs = red_scratch(scratch);
  MCEDM_HIP_TRY(hipMemsetAsync(rs.ticket, 0, sizeof(unsigned), (hipStream_t)stream));
  hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, sqnorm_out, rs);
  MCEDM_LAUNCH_CHECK("sqnorm_kernel");
  return MCEDM_OK;
}

extern "C" int mcedm_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                                   size_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                                   const double* sqnorm, double max_norm, double grad_scale, double ema_beta,
                                   int64_t step, void* stream) {
  MCEDM_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam: null pointer");
  MCEDM_REQUIRE(step >= 1, "adam: step counts from 1 (got %lld)", (long long)step);
  if (n == 0) return MCEDM_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  // scalar factors are formed in double on the host exactly as torch.optim.Adam / EmaModel form them in Python
  hipLaunchKernelGGL(adam_ema_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, ema, n, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                     (float)eps, (float)weight_decay, sqnorm, max_norm, (float)grad_scale, (float)ema_beta,
                     (float)(1.0 - ema_beta), (float)sqrt(bc2));
  MCEDM_LAUNCH_CHECK("adam_ema_kernel");
  return MCEDM_OK;
}
