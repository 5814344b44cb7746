// pde.hip -- PDE residuals of the reference's test-time metrics (models/pde_loss.py), SURVEY.md section 8 f3 (forward).
//
//   swe_fv_*   SweFvLoss.f_t_swp1d / calculate_loss  (models/pde_loss.py:131-165, 211-225): FORCE finite-volume step
//              along x for every (b, t) row; the residual compares step(pred[t-1]) with gt[t] (row 0: pred[0]).
//   darcy_*    DarcyLoss.calculate_loss              (models/pde_loss.py:30-54): -div(a grad u) = 1, central differences.
//
// Pure streaming kernels (one thread per output cell, 3-point / 5x5 neighbourhoods served by L1/L2): HBM-bound at
// 16 B (SWE) / 12 B (Darcy) per cell.  Built with -ffp-contract=off and written in the reference's evaluation order,
// so the results are bit-identical to the PyTorch CPU path (tests compare with torch.equal).
#include "common.hpp"

namespace mcedm {

struct SweCell { float h, u; };

// one FORCE step for cell x of a row (models/pde_loss.py:147-163); row = (h, u) pairs
__device__ __forceinline__ SweCell swe_force_cell(const float2* __restrict__ row, int x, int X, float half_dt, float dx) {
  const float eps = 1e-8f;
  float h[3], hu[3], upd[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int xi = x + i - 1;
    xi = xi < 0 ? 0 : (xi > X - 1 ? X - 1 : xi);          // replicate padding (set_boundary, :120-129)
    const float2 c = row[xi];
    h[i] = c.x;
    hu[i] = c.y * c.x;
    upd[i] = hu[i] * hu[i] / (h[i] + eps) + 0.5f * (h[i] * h[i]);
  }
  float hm[2], hum[2], upd2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    hm[j] = 0.5f * (h[j] + h[j + 1]) - half_dt * (hu[j + 1] - hu[j]) / dx;
    hum[j] = 0.5f * (hu[j] + hu[j + 1]) - half_dt * (upd[j + 1] - upd[j]) / dx;
    upd2[j] = hum[j] * hum[j] / (hm[j] + eps) + 0.5f * (hm[j] * hm[j]);
  }
  SweCell o;
  o.h = 0.5f * (hm[0] + hm[1]) - half_dt * (hum[1] - hum[0]) / dx;
  const float hu_next = 0.5f * (hum[0] + hum[1]) - half_dt * (upd2[1] - upd2[0]) / dx;
  o.u = hu_next / (o.h + eps);
  return o;
}

__global__ __launch_bounds__(256) void swe_fv_step_kernel(const float2* __restrict__ s, float2* __restrict__ out, int X,
                                                          size_t cells, float half_dt, float dx) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const size_t r = i / X;
  const int x = (int)(i - r * X);
  const SweCell c = swe_force_cell(s + r * X, x, X, half_dt, dx);
  out[i] = make_float2(c.h, c.u);
}

__global__ __launch_bounds__(256) void swe_fv_residual_kernel(const float2* __restrict__ pred, const float2* __restrict__ gt,
                                                              float2* __restrict__ out, int T, int X, size_t cells,
                                                              float half_dt, float dx, float scale2_h, float scale2_u,
                                                              int clamp) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const size_t r = i / X;                    // row = b * T + t
  const int x = (int)(i - r * X);
  const int t = (int)(r % T);
  float vh, vu;
  if (t == 0) {                              // the initial condition row is kept (:217)
    const float2 c = pred[i];
    vh = c.x; vu = c.y;
  } else {
    const SweCell c = swe_force_cell(pred + (r - 1) * X, x, X, half_dt, dx);
    vh = c.h; vu = c.u;
  }
  if (vh != vh) vh = 0.f;                    // pred_next_with_ic[isnan] = 0 (:218)
  if (vu != vu) vu = 0.f;
  const float2 g = gt[i];
  float lh = (vh - g.x) * (vh - g.x) / scale2_h;
  float lu = (vu - g.y) * (vu - g.y) / scale2_u;
  if (clamp) { lh = lh > 1.f ? 1.f : lh; lu = lu > 1.f ? 1.f : lu; }     // torch.clamp(max=1): NaN stays NaN
  out[i] = make_float2(lh, lu);
}

__global__ __launch_bounds__(256) void darcy_residual_kernel(const float2* __restrict__ pred, float* __restrict__ out, int S,
                                                             size_t cells, float two_dx, float denom, int clamp) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const int n = S - 4;
  const size_t b = i / ((size_t)n * n);
  const int rem = (int)(i - b * n * n);
  const int oi = rem / n, oj = rem - oi * n;
  const float2* f = pred + b * S * S;        // (a, u) pairs, row-major (first spatial index = rows)
  auto A = [&](int p, int q) { return f[(size_t)p * S + q].x; };
  auto U = [&](int p, int q) { return f[(size_t)p * S + q].y; };
  // index space of ux / uy / aux / auy is (S-2)^2: entry (p, q) sits at grid point (p+1, q+1)
  auto aux = [&](int p, int q) { return A(p + 1, q + 1) * ((U(p + 2, q + 1) - U(p, q + 1)) / two_dx); };
  auto auy = [&](int p, int q) { return A(p + 1, q + 1) * ((U(p + 1, q + 2) - U(p + 1, q)) / two_dx); };
  const float auxx = (aux(oi + 2, oj + 1) - aux(oi, oj + 1)) / two_dx;
  const float auyy = (auy(oi + 1, oj + 2) - auy(oi + 1, oj)) / two_dx;
  const float Du = -(auxx + auyy);
  float l = (Du - 1.f) * (Du - 1.f);
  l = l / denom;
  if (clamp) l = l > 1.f ? 1.f : l;
  out[i] = l;
}

}  // namespace mcedm

using namespace mcedm;

extern "C" int mcedm_swe_fv_step(const float* s, float* out, int B, int T, int X, float half_dt, float dx, void* stream) {
  MCEDM_REQUIRE(s && out && B > 0 && T > 0 && X > 0, "swe_fv_step: bad arguments");
  const size_t cells = (size_t)B * T * X;
  hipLaunchKernelGGL(swe_fv_step_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)s, (float2*)out, X, cells, half_dt, dx);
  MCEDM_LAUNCH_CHECK("swe_fv_step_kernel");
  return MCEDM_OK;
}

extern "C" int mcedm_swe_fv_residual(const float* pred, const float* gt, float* out, int B, int T, int X, float half_dt,
                                     float dx, float scale2_h, float scale2_u, int clamp, void* stream) {
  MCEDM_REQUIRE(pred && gt && out && B > 0 && T > 0 && X > 0, "swe_fv_residual: bad arguments");
  const size_t cells = (size_t)B * T * X;
  hipLaunchKernelGGL(swe_fv_residual_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)pred, (const float2*)gt, (float2*)out, T, X, cells, half_dt, dx, scale2_h, scale2_u, clamp);
  MCEDM_LAUNCH_CHECK("swe_fv_residual_kernel");
  return MCEDM_OK;
}

extern "C" int mcedm_darcy_residual(const float* pred, float* out, int B, int S, float two_dx, float denom, int clamp,
                                    void* stream) {
  MCEDM_REQUIRE(pred && out && B > 0 && S > 4, "darcy_residual: needs a grid larger than 4 x 4");
  const size_t cells = (size_t)B * (S - 4) * (S - 4);
  hipLaunchKernelGGL(darcy_residual_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)pred, out, S, cells, two_dx, denom, clamp);
  MCEDM_LAUNCH_CHECK("darcy_residual_kernel");
  return MCEDM_OK;
}
