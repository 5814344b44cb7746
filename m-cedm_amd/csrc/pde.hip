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
#include "edm.hpp"

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

// ---- guidance gradients: the return_d=True branches (models/pde_loss.py:231-242, 60-75) ---------------------------------
// Analytic adjoints of the stencils above (the reference differentiates them with torch.autograd).  Fields are addressed
// through element strides so that the same kernels serve the metric layout (b, t, x, 2) and the sampler's NCHW denoised
// state; `sub` / `div` un-normalise on the fly (x * div + sub, models/normalizer.py:24-27; 0 / 1 for physical inputs).
struct FieldView {
  const float* p[2];             // the two fields (h, u) / (a, u): they may live in different tensors (cond and state)
  long sb[2], st, sx;            // element strides of (batch [per field], row, column)
  float sub[2], div[2];
  __device__ __forceinline__ float get(long b, int t, int x, int c) const { return p[c][b * sb[c] + t * st + x * sx] * div[c] + sub[c]; }
};
// where the gradient goes: both fields, or (mean != 0) their mean into plane 0 -- torch.mean(dim=1, keepdim=True) of the
// single-task models' get_dx_pde (models/ddim.py:1441-1448)
struct GradView {
  float* p[2];
  long sb[2], st, sx;
  int mean;
  __device__ __forceinline__ void put(long b, int t, int x, float g0, float g1) const {
    if (mean) { p[0][b * sb[0] + t * st + x * sx] = (g0 + g1) / 2.0f; return; }
    p[0][b * sb[0] + t * st + x * sx] = g0;
    p[1][b * sb[1] + t * st + x * sx] = g1;
  }
};

struct Swe6 { float dh[3], du[3]; };

// vector-Jacobian product of one FORCE cell update (swe_force_cell) w.r.t. its three input cells, evaluated operation by
// operation in reverse like autograd does (so NaN / inf propagate the same way before the final NaN -> 0)
__device__ __forceinline__ Swe6 swe_force_cell_vjp(const float (&hin)[3], const float (&uin)[3], float half_dt, float dx,
                                                   float gh, float gu) {
  const float eps = 1e-8f;
  float h[3], hu[3], upd[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    h[i] = hin[i];
    hu[i] = uin[i] * hin[i];
    upd[i] = hu[i] * hu[i] / (h[i] + eps) + 0.5f * (h[i] * h[i]);
  }
  float hm[2], hum[2], upd2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    hm[j] = 0.5f * (h[j] + h[j + 1]) - half_dt * (hu[j + 1] - hu[j]) / dx;
    hum[j] = 0.5f * (hu[j] + hu[j + 1]) - half_dt * (upd[j + 1] - upd[j]) / dx;
    upd2[j] = hum[j] * hum[j] / (hm[j] + eps) + 0.5f * (hm[j] * hm[j]);
  }
  const float oh = 0.5f * (hm[0] + hm[1]) - half_dt * (hum[1] - hum[0]) / dx;
  const float hun = 0.5f * (hum[0] + hum[1]) - half_dt * (upd2[1] - upd2[0]) / dx;
  const float c = half_dt / dx;
  // u = hun / (oh + eps); h = oh
  const float d_hun = gu / (oh + eps);
  const float d_oh = gh + gu * (-hun / ((oh + eps) * (oh + eps)));
  float d_hm[2], d_hum[2], d_upd2[2];
  d_hm[0] = 0.5f * d_oh; d_hm[1] = 0.5f * d_oh;
  d_hum[0] = c * d_oh + 0.5f * d_hun; d_hum[1] = -c * d_oh + 0.5f * d_hun;
  d_upd2[0] = c * d_hun; d_upd2[1] = -c * d_hun;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    d_hum[j] += d_upd2[j] * (2.0f * hum[j] / (hm[j] + eps));
    d_hm[j] += d_upd2[j] * (-(hum[j] * hum[j]) / ((hm[j] + eps) * (hm[j] + eps)) + hm[j]);
  }
  float d_h[3] = {0.f, 0.f, 0.f}, d_hu[3] = {0.f, 0.f, 0.f}, d_upd[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    d_h[j] += 0.5f * d_hm[j]; d_h[j + 1] += 0.5f * d_hm[j];
    d_hu[j] += c * d_hm[j]; d_hu[j + 1] -= c * d_hm[j];
    d_hu[j] += 0.5f * d_hum[j]; d_hu[j + 1] += 0.5f * d_hum[j];
    d_upd[j] += c * d_hum[j]; d_upd[j + 1] -= c * d_hum[j];
  }
  Swe6 o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    d_hu[i] += d_upd[i] * (2.0f * hu[i] / (h[i] + eps));
    d_h[i] += d_upd[i] * (-(hu[i] * hu[i]) / ((h[i] + eps) * (h[i] + eps)) + h[i]);
    o.du[i] = d_hu[i] * h[i];
    o.dh[i] = d_h[i] + d_hu[i] * uin[i];
  }
  return o;
}

// d mean(((with_ic(pred) - gt)^2 / scale2)) / d pred, with_ic = cat(pred[:, 0:1], step(pred)[:, :-1]) and NaNs of with_ic
// replaced by 0 before the residual (models/pde_loss.py:211-225); out in pred's layout.  One thread per cell: it
// re-evaluates the (at most three) cell updates of the next row that read it and keeps its own share of each VJP.
__global__ __launch_bounds__(256) void swe_fv_guidance_kernel(FieldView pred, FieldView gt, GradView out, int T,
                                                              int X, size_t cells, float half_dt, float dx, float scale2_h,
                                                              float scale2_u, float inv_n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const size_t r = i / X;
  const int x = (int)(i - r * X);
  const long b = (long)(r / T);
  const int t = (int)(r % T);
  float acc_h = 0.f, acc_u = 0.f;
  if (t == 0) {                               // the initial-condition row enters the residual as it is
    const float vh = pred.get(b, 0, x, 0), vu = pred.get(b, 0, x, 1);
    if (vh == vh) acc_h = 2.0f * (vh - gt.get(b, 0, x, 0)) / scale2_h * inv_n;
    if (vu == vu) acc_u = 2.0f * (vu - gt.get(b, 0, x, 1)) / scale2_u * inv_n;
  }
  if (t + 1 < T) {                            // row t is stepped and compared with gt row t + 1
    for (int xo = x - 1; xo <= x + 1; ++xo) {
      if (xo < 0 || xo >= X) continue;
      int xi[3];
      float hin[3], uin[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        int q = xo + k - 1;
        q = q < 0 ? 0 : (q > X - 1 ? X - 1 : q);
        xi[k] = q;
        hin[k] = pred.get(b, t, q, 0);
        uin[k] = pred.get(b, t, q, 1);
      }
      // forward value of this output cell -> upstream gradient of the residual (zero where the stepped value is NaN)
      float2 row3[3] = {make_float2(hin[0], uin[0]), make_float2(hin[1], uin[1]), make_float2(hin[2], uin[2])};
      const SweCell o = swe_force_cell(row3, 1, 3, half_dt, dx);
      const float gh = (o.h == o.h) ? 2.0f * (o.h - gt.get(b, t + 1, xo, 0)) / scale2_h * inv_n : 0.f;
      const float gu = (o.u == o.u) ? 2.0f * (o.u - gt.get(b, t + 1, xo, 1)) / scale2_u * inv_n : 0.f;
      const Swe6 v = swe_force_cell_vjp(hin, uin, half_dt, dx, gh, gu);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (xi[k] == x) { acc_h += v.dh[k]; acc_u += v.du[k]; }
    }
  }
  if (acc_h != acc_h) acc_h = 0.f;            // dloss[isnan] = 0 (:239-240)
  if (acc_u != acc_u) acc_u = 0.f;
  out.put(b, t, x, acc_h, acc_u);
}

// Darcy: G = d loss / d Du on the (S-4)^2 interior, loss = mean(L) or mean(log(2 (1 - sigmoid(1e5 L)) + 1e-12)), L = (Du - 1)^2
__global__ __launch_bounds__(256) void darcy_guidance_g_kernel(FieldView pred, float* __restrict__ G, int S, size_t cells,
                                                               float two_dx, int calc_prob, float inv_n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const int n = S - 4;
  const long b = (long)(i / ((size_t)n * n));
  const int rem = (int)(i - (size_t)b * n * n);
  const int oi = rem / n, oj = rem - oi * n;
  auto A = [&](int p, int q) { return pred.get(b, p, q, 0); };
  auto U = [&](int p, int q) { return pred.get(b, p, q, 1); };
  auto aux = [&](int p, int q) { return A(p + 1, q + 1) * ((U(p + 2, q + 1) - U(p, q + 1)) / two_dx); };
  auto auy = [&](int p, int q) { return A(p + 1, q + 1) * ((U(p + 1, q + 2) - U(p + 1, q)) / two_dx); };
  const float auxx = (aux(oi + 2, oj + 1) - aux(oi, oj + 1)) / two_dx;
  const float auyy = (auy(oi + 1, oj + 2) - auy(oi + 1, oj)) / two_dx;
  const float Du = -(auxx + auyy);
  const float L = (Du - 1.f) * (Du - 1.f);
  float gL = inv_n;
  if (calc_prob) {
    const float z = 1e5f * L;
    const float sg = 1.0f / (1.0f + expf(-z));
    const float inner = 2.0f * (1.0f - sg) + 1e-12f;
    gL = inv_n / inner * (2.0f * (-(sg * (1.0f - sg)))) * 1e5f;
  }
  G[i] = gL * (2.0f * (Du - 1.f));
}

// gather d a, d u for every grid point from G (zero outside the interior); see the index algebra in DESIGN.md
__global__ __launch_bounds__(256) void darcy_guidance_gather_kernel(FieldView pred, const float* __restrict__ G,
                                                                    GradView out, int S, size_t cells, float two_dx) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const long b = (long)(i / ((size_t)S * S));
  const int rem = (int)(i - (size_t)b * S * S);
  const int p = rem / S, q = rem - p * S;
  const int n = S - 4, m2 = S - 2;
  const float* Gb = G + (size_t)b * n * n;
  auto g = [&](int a, int c) -> float { return (a >= 0 && a < n && c >= 0 && c < n) ? Gb[(size_t)a * n + c] : 0.f; };
  auto A = [&](int r, int c) { return pred.get(b, r, c, 0); };
  auto U = [&](int r, int c) { return pred.get(b, r, c, 1); };
  // d loss / d aux(i, j), d auy(i, j) on the (S-2)^2 index space (auxx = (aux[i+2, j+1] - aux[i, j+1]) / two_dx, Du = -(auxx + auyy))
  auto daux = [&](int a, int c) -> float { return (a >= 0 && a < m2 && c >= 0 && c < m2) ? (g(a, c - 1) - g(a - 2, c - 1)) / two_dx : 0.f; };
  auto dauy = [&](int a, int c) -> float { return (a >= 0 && a < m2 && c >= 0 && c < m2) ? (g(a - 1, c) - g(a - 1, c - 2)) / two_dx : 0.f; };
  auto dux = [&](int a, int c) -> float { return (a >= 0 && a < m2 && c >= 0 && c < m2) ? daux(a, c) * A(a + 1, c + 1) : 0.f; };
  auto duy = [&](int a, int c) -> float { return (a >= 0 && a < m2 && c >= 0 && c < m2) ? dauy(a, c) * A(a + 1, c + 1) : 0.f; };
  float da = 0.f;
  if (p >= 1 && p <= S - 2 && q >= 1 && q <= S - 2) {
    const int a = p - 1, c = q - 1;
    const float ux = (U(a + 2, c + 1) - U(a, c + 1)) / two_dx, uy = (U(a + 1, c + 2) - U(a + 1, c)) / two_dx;
    da = daux(a, c) * ux + dauy(a, c) * uy;
  }
  float du = (dux(p - 2, q - 1) - dux(p, q - 1)) / two_dx + (duy(p - 1, q - 2) - duy(p - 1, q)) / two_dx;
  if (da != da) da = 0.f;
  if (du != du) du = 0.f;
  out.put(b, p, q, da, du);
}

int launch_swe_guidance(const GuideIO& io, int B, int T, int X, float half_dt, float dx, float scale2_h, float scale2_u,
                        hipStream_t s) {
  FieldView pv{{io.in[0], io.in[1]}, {io.in_sb[0], io.in_sb[1]}, io.st, io.sx, {io.sub[0], io.sub[1]}, {io.div[0], io.div[1]}};
  FieldView gv{{io.gt[0], io.gt[1]}, {io.in_sb[0], io.in_sb[1]}, io.st, io.sx, {io.sub[0], io.sub[1]}, {io.div[0], io.div[1]}};
  GradView ov{{io.out[0], io.out[1]}, {io.out_sb[0], io.out_sb[1]}, io.out_st, io.out_sx, io.mean};
  const size_t cells = (size_t)B * T * X;
  const float inv_n = 1.0f / (float)(2.0 * (double)cells);
  hipLaunchKernelGGL(swe_fv_guidance_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, pv, gv, ov, T, X, cells,
                     half_dt, dx, scale2_h, scale2_u, inv_n);
  MCEDM_LAUNCH_CHECK("swe_fv_guidance_kernel");
  return MCEDM_OK;
}

int launch_darcy_guidance(const GuideIO& io, float* scratch, int B, int S, float two_dx, int calc_prob, hipStream_t s) {
  FieldView pv{{io.in[0], io.in[1]}, {io.in_sb[0], io.in_sb[1]}, io.st, io.sx, {io.sub[0], io.sub[1]}, {io.div[0], io.div[1]}};
  GradView ov{{io.out[0], io.out[1]}, {io.out_sb[0], io.out_sb[1]}, io.out_st, io.out_sx, io.mean};
  const size_t inner = (size_t)B * (S - 4) * (S - 4), cells = (size_t)B * S * S;
  hipLaunchKernelGGL(darcy_guidance_g_kernel, dim3((unsigned)((inner + 255) / 256)), dim3(256), 0, s, pv, scratch, S, inner, two_dx,
                     calc_prob, 1.0f / (float)inner);
  MCEDM_LAUNCH_CHECK("darcy_guidance_g_kernel");
  hipLaunchKernelGGL(darcy_guidance_gather_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, pv, scratch, ov, S,
                     cells, two_dx);
  MCEDM_LAUNCH_CHECK("darcy_guidance_gather_kernel");
  return MCEDM_OK;
}

}  // namespace mcedm

using namespace mcedm;

// (b, t, x, 2) channel-last tensors: field c of cell (b, t, x) at ((b * T + t) * X + x) * 2 + c
static GuideIO channel_last_io(const float* pred, const float* gt, float* out, long rows, long cols) {
  GuideIO io{};
  io.in[0] = pred; io.in[1] = pred + 1; io.gt[0] = gt; io.gt[1] = gt ? gt + 1 : nullptr;
  io.in_sb[0] = io.in_sb[1] = rows * cols * 2; io.st = cols * 2; io.sx = 2;
  io.out[0] = out; io.out[1] = out + 1; io.out_sb[0] = io.out_sb[1] = rows * cols * 2; io.out_st = cols * 2; io.out_sx = 2;
  io.sub[0] = io.sub[1] = 0.f; io.div[0] = io.div[1] = 1.f; io.mean = 0;
  return io;
}

extern "C" int mcedm_swe_fv_guidance(const float* pred, const float* gt, float* out, int B, int T, int X, float half_dt, float dx,
                                     float scale2_h, float scale2_u, void* stream) {
  MCEDM_REQUIRE(pred && gt && out && B > 0 && T > 0 && X > 0, "swe_fv_guidance: bad arguments");
  return launch_swe_guidance(channel_last_io(pred, gt, out, T, X), B, T, X, half_dt, dx, scale2_h, scale2_u, (hipStream_t)stream);
}

extern "C" int mcedm_darcy_guidance(const float* pred, float* out, float* scratch, int B, int S, float two_dx, int calc_prob,
                                    void* stream) {
  MCEDM_REQUIRE(pred && out && scratch && B > 0 && S > 4, "darcy_guidance: needs a grid larger than 4 x 4 and a scratch buffer");
  return launch_darcy_guidance(channel_last_io(pred, nullptr, out, S, S), scratch, B, S, two_dx, calc_prob, (hipStream_t)stream);
}

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
