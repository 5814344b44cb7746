// edm.hpp -- launchers of the EDM preconditioning / Heun sampler kernels (edm.hip).
#pragma once
#include "common.hpp"

namespace mcedm {

int launch_precond_prepare(const float* sigma_dev, float sigma_host, int use_host, int n, float sigma_data,
                           int cond_ch, int in_ch, float* coefs4, float* c_noise, Coef* conv_in_coef,
                           hipStream_t stream);
int launch_precond_finish(const float* x, const float* F, const float* Fu, float w, const float* coefs4, int n_sigma,
                          size_t per_sample, size_t total, float* D, float* F_out, hipStream_t stream);
int launch_heun_init(const float* cond, int cond_ch, int in_ch, size_t hw, const float* mask, const float* noise,
                     double t0, size_t total, double* x, float* x32, hipStream_t s);
int launch_heun_churn(double* x, const double* eps, const float* mask, double c, size_t total, float* x32,
                      hipStream_t s);
int launch_heun_euler(const double* x_hat, const float* D, const float* mask, double t_hat, double dt, size_t total,
                      double* d_cur, double* x_next, float* x32, hipStream_t s);
int launch_heun_correct(const double* x_hat, const double* d_cur, const float* D, const float* mask, double t_next,
                        double dt, size_t total, double* x_next, float* x32, hipStream_t s);
int launch_heun_store(const double* x, int C, size_t hw, int t, int T, size_t total, double* out, hipStream_t s);

}  // namespace mcedm
