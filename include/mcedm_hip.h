/* mcedm_hip.h -- C ABI of libmcedm_hip.so, the MI355X (gfx950) implementation of the
 * m-cedm EDM hot path: ADM/EDM U-Net forward (+backward), EDM preconditioning, the
 * deterministic/stochastic Heun sampler, the masked EDM loss and the Adam+EMA step.
 *
 * The reference (katehai/m-cedm) is pure Python and has no FFI of its own; the seam it
 * exposes is Python-level.  Each entry point below names the
 * reference function whose device work it replaces (paths relative to the reference
 * checkout).  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only; no torch / C++ types.
 *  - Every entry point returns 0 on success and a negative mcedm_status on failure;
 *    mcedm_last_error() returns a thread-local message for the last failure.
 *  - All device buffers are owned by the caller (parameters, packed weights, workspace,
 *    inputs, outputs).  The library allocates only the host-side plan object.
 *  - Kernels are enqueued on the caller's HIP stream (passed as void* = hipStream_t);
 *    nothing synchronises the device, so calls are graph-capturable.
 *  - Tensors are dense NCHW fp32 unless stated; the sampler state is fp64 like the reference.
 */
#ifndef MCEDM_HIP_H_
#define MCEDM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCEDM_ABI_VERSION 4   /* 4: per-plan kernel variants (mcedm_*_plan_set_variant), mcedm_heun_sample_rng;
                               * 3: mcedm_edm_loss / mcedm_sqnorm take caller-owned reduction scratch; mcedm_ddim_timesteps */
/* Device scratch of one grid-wide fixed-order reduction (mcedm_edm_loss, mcedm_sqnorm): 8-byte aligned, contents
 * irrelevant on entry, private to the call until it has completed on its stream.  Two calls that may run concurrently (two
 * plans on two streams of one device) need two scratch areas; calls ordered on one stream can share one. */
#define MCEDM_REDUCE_SCRATCH_BYTES (4096 * 8 + 64)
#define MCEDM_MAX_LEVELS 8

typedef enum {
  MCEDM_OK = 0,
  MCEDM_ERR_INVALID = -1,      /* bad argument / unsupported shape (rejected before any launch) */
  MCEDM_ERR_UNSUPPORTED = -2,  /* configuration outside the hot path (e.g. cond_enc, self_cond) */
  MCEDM_ERR_WORKSPACE = -3,    /* workspace / packed buffer too small */
  MCEDM_ERR_HIP = -4           /* a HIP runtime call or launch failed */
} mcedm_status;

/* Architecture of DhariwalUNet as read from hparams.model (models/adm_blocks.py:203-317,
 * configs/model/adm_edm_mcedm_res32.yaml:4-28).  cat_cond=True, self_cond=False, label_dim=augment_dim=0,
 * dropout=0; dx_cond (network conditioning on the PDE-residual gradient, adm_blocks.py:233-280, 334-362) in both of
 * the reference's forms: dx_mode. */
#define MCEDM_DX_NONE 0   /* dx_cond False */
#define MCEDM_DX_CAT 1    /* dx_cond True, cat_dx True: conv_in reads cat(cond, x, dx) (adm_blocks.py:238, 334-339) */
#define MCEDM_DX_ENC 2    /* dx_cond True, cat_dx False: x_feat = combine_enc(cat(conv_in(.), dx_enc(dx))), dx_enc =
                             Conv3x3 -> GELU -> Conv3x3; dx None -> zero features (adm_blocks.py:266-280, 352-362) */
typedef struct {
  int32_t in_channels;       /* state channels (h_ch + u_ch), 2 */
  int32_t cond_channels;     /* concatenated conditioning channels, 2 (0 = none) */
  int32_t out_channels;      /* out_ch */
  int32_t ch;                /* base width; emb_channels == ch */
  int32_t n_levels;          /* len(ch_mult) */
  int32_t ch_mult[MCEDM_MAX_LEVELS];
  int32_t num_res_blocks;
  int32_t resolution;        /* label only: level names are resolution >> level */
  int32_t n_attn_resolutions;
  int32_t attn_resolutions[MCEDM_MAX_LEVELS];
  int32_t channels_per_head; /* 64 */
  float eps;                 /* GroupNorm eps, 1e-5 */
  int32_t dx_channels;       /* channels of the dx input = hparams.model.in_channels when dx_cond, else 0 */
  int32_t dx_mode;           /* MCEDM_DX_* */
} mcedm_unet_desc;

/* Sampler parameters (configs/diff_sampler/edm_sampler.yaml:1-20; fields read by
 * PlMcedm.sample_edm, models/mcedm.py:570-638). */
typedef struct {
  int32_t timesteps;
  double sigma_min, sigma_max, rho;
  double S_churn, S_min, S_max, S_noise;
  double w;                  /* classifier-free guidance weight; |w| < 1e-3 == off (mcedm.py:453) */
  double sigma_data;         /* 1.0 (mcedm.py:47) */
  double net_sigma_min, net_sigma_max; /* 0.002 / 80 (mcedm.py:49-50) */
} mcedm_sampler_desc;

typedef struct mcedm_plan mcedm_plan;

int mcedm_version(void);
const char* mcedm_last_error(void);

/* ---- plan: the static block list of the network ------------------------------------ */
int mcedm_unet_plan_create(const mcedm_unet_desc* desc, mcedm_plan** out);
void mcedm_unet_plan_destroy(mcedm_plan* plan);

/* Kernel variants of ONE plan (re-entrant per plan).  Every kernel family that exists in two forms is chosen per
 * call in three steps, first hit wins: the plan's own setting (this function; a field of the plan, in force while one of the
 * plan's entry points executes on the calling thread), the process-wide test hooks mcedm_op_set_* below (kernel-level calls
 * have no plan), the environment variable.  value: 1 on, 0 off, -1 back to the process default.  Two plans in one process --
 * on two threads or two streams -- cannot flip each other's kernels.  MCEDM_VARIANT_CONV_WINO also decides how a plan lays
 * out its workspace (whether a block's 1x1 skip projection is folded into conv1): set it before the first
 * mcedm_unet_workspace_bytes / forward of the plan and leave it. */
#define MCEDM_VARIANT_CONV_WINO 0       /* Winograd F(2x2, 3x3) forward / data-gradient convs (env MCEDM_WINOGRAD, default 1) */
#define MCEDM_VARIANT_CONV_WINO1 1      /* its one-wave-per-SIMD form for 128-channel shapes (env MCEDM_WINO1, default 0) */
#define MCEDM_VARIANT_CONV_RESIDENT 2   /* input-resident conv kernels at <= 32 x 32 (env MCEDM_CONV_RESIDENT, default 1) */
#define MCEDM_VARIANT_CONV8 3           /* experimental 8-wave direct conv (env MCEDM_CONV8, default 0) */
#define MCEDM_VARIANT_ATTN_FUSED 4      /* single-launch attention block at 8 x 8 x 64 (env MCEDM_ATTN_FUSED, default 1) */
#define MCEDM_VARIANT_WGRAD_WINO 5      /* Winograd F(3x3, 2x2) weight gradient (env MCEDM_WGRAD_WINO, default 1) */
#define MCEDM_VARIANT_CONV1X1_REG 6     /* register-direct GEMM for un-transformed 1x1 convs at >= 32 x 32 (env MCEDM_CONV1X1_REG, default 1) */
int mcedm_unet_plan_set_variant(mcedm_plan* plan, int which, int value);

/* Parameter table in DhariwalUNet.state_dict() order (parameters only, no buffers).
 * name is owned by the plan. */
int mcedm_unet_param_count(const mcedm_plan* plan);
int mcedm_unet_param_info(const mcedm_plan* plan, int index, const char** name, int64_t* numel,
                          int32_t* ndim, int64_t shape[4]);

/* Derived ("packed") weights: MFMA-friendly copies of the conv / linear weights.  Must be
 * re-run whenever a parameter changes.  params[i] is the device pointer of parameter i. */
int mcedm_unet_packed_bytes(const mcedm_plan* plan, size_t* bytes);
int mcedm_unet_pack_weights(const mcedm_plan* plan, const float* const* params, void* packed, void* stream);

/* Workspace needed by one forward at batch B and spatial size HxW (H, W multiples of
 * 2^(n_levels-1)).  training != 0 keeps every activation the backward needs. */
int mcedm_unet_workspace_bytes(const mcedm_plan* plan, int B, int H, int W, int training, size_t* bytes);

/* DhariwalUNet.forward (models/adm_blocks.py:364-404): out = F(cat(cond, x_scale * x), noise_labels).
 *  x            [B, in_channels, H, W]
 *  cond         [B, cond_channels, H, W] or NULL (treated as zeros, adm_blocks.py:328-331)
 *  x_scale      [n_noise] per-sample factor applied to x while it is read (EDM c_in), or NULL (= 1)
 *  noise_labels [n_noise], n_noise == 1 (broadcast, sampling) or == B (training)
 *  out          [B, out_channels, H, W] */
int mcedm_unet_forward(const mcedm_plan* plan, const void* packed, const float* x, const float* cond,
                       const float* x_scale, const float* noise_labels, int n_noise, float* out,
                       void* workspace, size_t workspace_bytes, int B, int H, int W, int training,
                       void* stream);

/* PlMcedm.model_precond / get_denoised with w == 0 (models/mcedm.py:199-211, 443-461):
 * D = c_skip*x + c_out*F(c_in*x, ln(sigma)/4, cond).  sigma is a device array of n_sigma (1 or B)
 * fp32 values.  F_out may be NULL. */
int mcedm_edm_denoise(const mcedm_plan* plan, const void* packed, const float* x, const float* sigma,
                      int n_sigma, const float* cond, float* D_out, float* F_out, void* workspace,
                      size_t workspace_bytes, int B, int H, int W, int training, double sigma_data,
                      void* stream);

/* PlMcedm.sample_edm (models/mcedm.py:570-638), guide_dx False / dx_cond False.
 *  cond       [B, cond_channels, H, W] fp32; its first in_channels channels are hu_known
 *  mask       [B, in_channels, H, W] fp32, 1 = missing; NULL = no mask: the unmasked single-task sampler of
 *             PlCondEdm.sample_edm (models/ddim.py:1532-1601), where cond is pure conditioning (may be NULL)
 *  init_noise [B, in_channels, H, W] fp32  (the reference's randn_like(hu), mcedm.py:576)
 *  step_noise [timesteps, B, in_channels, H, W] fp64 or NULL (the per-step randn_like(x_cur) of
 *             mcedm.py:608, fp64 like x_cur; NULL is only legal when no step has gamma > 0)
 *  out        fp64, [B, 1, H, W, C] if return_last else [B, timesteps+1, H, W, C]
 *  workspace must hold mcedm_sampler_workspace_bytes(). */
int mcedm_sampler_workspace_bytes(const mcedm_plan* plan, int B, int H, int W, size_t* bytes);
int mcedm_heun_sample(const mcedm_plan* plan, const void* packed, const mcedm_sampler_desc* sp,
                      const float* cond, const float* mask, const float* init_noise,
                      const double* step_noise, double* out, int return_last, void* workspace,
                      size_t workspace_bytes, int B, int H, int W, void* stream);

/* mcedm_heun_sample with the per-step churn noise drawn ON THE DEVICE: models/mcedm.py:604-608 draws randn_like(x_cur) in every
 * step; here the kernel that applies `x_hat = x_cur + sqrt(t_hat^2 - t_cur^2) * S_noise * eps * mask` generates eps itself
 * (Philox4x32-10 + Box-Muller in fp64; key = the 64-bit seed at *rng_seed in DEVICE memory, counter = (element pair, step index)),
 * so there is no [timesteps][B][C][H][W] fp64 step_noise tensor and a captured HIP graph replays with fresh noise once the host
 * has written a new seed.  Same moments as, but not the stream of, torch's generator; mcedm_normal_fill(out, n, rng_seed, draw = i)
 * writes step i's draw out, and feeding those tensors to mcedm_heun_sample reproduces this call bit for bit. */
int mcedm_heun_sample_rng(const mcedm_plan* plan, const void* packed, const mcedm_sampler_desc* sp, const float* cond,
                          const float* mask, const float* init_noise, const uint64_t* rng_seed, double* out, int return_last,
                          void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);
/* PDE guidance inside the single-task sampler (PlCondEdm.sample_edm with guide_dx=True, models/ddim.py:1532-1601:
 * get_dx_log_prob :641-650 -> get_dx_pde :1424-1450, used at :1577-1579 and :1589-1591): after every denoiser call
 * dx = mean over the two fields of d residual(x_unnorm) / d x_unnorm, x_unnorm = (h from cond[:, 0], u = denoised state),
 * and d = (x - D) / t - weight * dx / t_hat.  system 1 = SweFvLoss (FORCE finite-volume residual along W), system 2 =
 * DarcyLoss in its log-probability form (calc_prob=True).  sub_* / div_* are the normalisers' statistics (scalars):
 * x_unnorm = x * div + sub.  mask must be NULL, in_channels 1, cond_channels >= 1.  (The joint model's hook,
 * models/mcedm.py:500-518, slices the wrong axis and raises in the reference; it is not built.) */
typedef struct {
  int32_t system;
  float half_dt, dx;         /* SWE: (float)(0.5 * Tn / n_times) and x[1] - x[0] of SweFvLoss.gen_x (fp32), as for mcedm_swe_fv_residual */
  float two_dx;              /* Darcy: (float)(2 * D / s), as for mcedm_darcy_residual */
  float sub_h, div_h, sub_u, div_u;
  double weight;             /* 5.0 (mcedm.py:616) */
} mcedm_guidance_desc;
int mcedm_heun_sample_guided(const mcedm_plan* plan, const void* packed, const mcedm_sampler_desc* sp,
                             const mcedm_guidance_desc* gd, const float* cond, const float* mask,
                             const float* init_noise, const double* step_noise, double* out, int return_last,
                             void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);
/* ---- dx_cond: the network conditioned on the PDE-residual gradient (plans with dx_mode != MCEDM_DX_NONE) ----------
 * The same three calls with the extra network input dx [B, dx_channels, H, W] (DhariwalUNet.forward(..., dx=dx),
 * adm_blocks.py:364-388; model_precond / get_denoised with dx, models/ddim.py:1661-1666, 1745-1763).  dx == NULL is the
 * reference's dx=None: zeros concatenated (cat_dx) or zero dx features (dx_enc).  The entry points without _dx accept such
 * plans too and mean dx = NULL.  dx carries no gradient (torch.autograd.grad without create_graph, models/pde_loss.py:233;
 * dx_detach); the backward returns the gradients of dx_enc / combine_enc like every other parameter.  n_buckets == 0: no
 * bucket events. */
int mcedm_unet_forward_dx(const mcedm_plan* plan, const void* packed, const float* x, const float* dx, const float* cond,
                          const float* x_scale, const float* noise_labels, int n_noise, float* out, void* workspace,
                          size_t workspace_bytes, int B, int H, int W, int training, void* stream);
int mcedm_edm_denoise_dx(const mcedm_plan* plan, const void* packed, const float* x, const float* dx, const float* sigma,
                         int n_sigma, const float* cond, float* D_out, float* F_out, void* workspace,
                         size_t workspace_bytes, int B, int H, int W, int training, double sigma_data, void* stream);
int mcedm_edm_denoise_backward_dx(const mcedm_plan* plan, const void* packed, const float* const* params, const float* x,
                                  const float* dx, const float* sigma, int n_sigma, const float* cond, const float* dD,
                                  float* const* grads, void* workspace, size_t workspace_bytes, int B, int H, int W,
                                  double sigma_data, int n_buckets, const int32_t* bucket_first_param,
                                  void* const* bucket_events, void* stream);
/* PlCondEdm.sample_edm of a dx_cond model (models/ddim.py:1532-1601): before EVERY denoiser call
 * dx_in = get_dx_input(h, x) (:601-639 with dx_norm == 'prob', the only normalisation the single-task get_dx_pde :1424-1450
 * can feed: its calc_prob=False result is 3-D and the other branches fail to unpack it) = the mean over the two fields of
 * the log-probability residual gradient at x_unnorm = (h from cond[:, 0], u = the CURRENT NOISY state cast to fp32), fed to
 * the network as dx.  dxc describes that residual (same fields as for mcedm_heun_sample_guided; weight unused); gd != NULL
 * additionally applies guide_dx to the denoised state as in mcedm_heun_sample_guided.  With sp->w != 0 the unconditional
 * branch runs without cond AND without dx (:1759-1760). */
int mcedm_heun_sample_dxcond(const mcedm_plan* plan, const void* packed, const mcedm_sampler_desc* sp,
                             const mcedm_guidance_desc* dxc, const mcedm_guidance_desc* gd, const float* cond,
                             const float* init_noise, const double* step_noise, double* out, int return_last,
                             void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);
/* Host helper: the float64 sigma schedule of mcedm.py:584-588 (timesteps+1 values, last = 0). */
int mcedm_edm_t_steps(const mcedm_sampler_desc* sp, double* t_steps);

/* ---- training --------------------------------------------------------------------- */
/* Masked, weighted EDM loss of training_step (models/mcedm.py:266-278, models/losses.py:48-59)
 * and its gradient w.r.t. D:  loss = mean_b sum_chw w_b (D*m - x*m)^2,  w_b = (s^2+sd^2)/(s*sd)^2.
 *  loss_out: 1 fp32 (device), accumulated from zero by this call;  dD_out [B,C,H,W] or NULL.
 *  mask == NULL: unmasked loss of PlCondEdm.training_step (models/ddim.py:1727).
 *  scratch: MCEDM_REDUCE_SCRATCH_BYTES of device memory (the per-block partial sums and the ticket of the fixed-order,
 *  atomic-free sum that makes the loss bitwise reproducible): the library keeps no device state of its own. */
int mcedm_edm_loss(const float* D, const float* x, const float* mask, const float* sigma, int B, int C,
                   int H, int W, double sigma_data, float* loss_out, float* dD_out, void* scratch, size_t scratch_bytes,
                   void* stream);

/* x_noise = x + mask*noise*sigma (mcedm.py:216; mask == NULL: x + noise*sigma, :218), sigma = exp(rnd_normal*P_std + P_mean)
 * (mcedm.py:271). */
int mcedm_edm_noise_inputs(const float* x, const float* mask, const float* noise, const float* rnd_normal,
                           int B, int C, int H, int W, double P_mean, double P_std, float* x_noise,
                           float* sigma_out, void* stream);

/* Backward of mcedm_edm_denoise through the U-Net: consumes the activations kept in
 * `workspace` by the matching forward (training != 0).  grads[i] receives dLoss/dparam_i
 * (overwritten).  dD is the gradient w.r.t. D_out. */
int mcedm_edm_denoise_backward(const mcedm_plan* plan, const void* packed, const float* const* params,
                               const float* x, const float* sigma, int n_sigma, const float* cond,
                               const float* dD, float* const* grads, void* workspace,
                               size_t workspace_bytes, int B, int H, int W, double sigma_data, void* stream);

/* Overlap hook for the data-parallel gradient exchange (the DDP bucketed all-reduce behind
 * configs/trainer/trainer_ddim.yaml:7 `strategy: ddp`).  Same as
 * mcedm_edm_denoise_backward, and additionally records the caller's HIP events on `stream`: bucket_events[k]
 * (hipEvent_t) is recorded as soon as the gradient of EVERY parameter with index >= bucket_first_param[k] has been
 * enqueued.  The backward finishes parameters from the last (out_conv) to the first (map_layer0), so the caller can
 * all-reduce bucket k = parameters [bucket_first_param[k], bucket_first_param[k-1]) on another stream (after
 * hipStreamWaitEvent) while the rest of the backward still runs.  bucket_first_param must decrease, end with 0 and
 * name first parameters of blocks; mcedm_unet_grad_buckets proposes such a split into at most max_buckets
 * roughly equal parts.  The library itself stays free of any communication dependency: the collective is the
 * caller's (RCCL through torch.distributed in m-cedm_amd/train.py). */
int mcedm_unet_grad_buckets(const mcedm_plan* plan, int max_buckets, int32_t* first_param, int* n_buckets);
int mcedm_edm_denoise_backward_bucketed(const mcedm_plan* plan, const void* packed, const float* const* params,
                                        const float* x, const float* sigma, int n_sigma, const float* cond,
                                        const float* dD, float* const* grads, void* workspace,
                                        size_t workspace_bytes, int B, int H, int W, double sigma_data,
                                        int n_buckets, const int32_t* bucket_first_param,
                                        void* const* bucket_events, void* stream);

/* Squared L2 norm of a flat fp32 buffer, accumulated in fp64 into *sqnorm_out (device, overwritten); fixed summation
 * order (bitwise reproducible).  scratch: MCEDM_REDUCE_SCRATCH_BYTES of device memory, as for mcedm_edm_loss. */
int mcedm_sqnorm(const float* g, size_t n, double* sqnorm_out, void* scratch, size_t scratch_bytes, void* stream);

/* Fused grad-clip + Adam + EMA on flat fp32 buffers (models/mcedm.py:139-168,
 * models/ddim_blocks.py:44-54, configs/trainer/trainer_ddim.yaml:8-9).
 *  clip = min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) is computed on device from sqnorm (NULL = no clip);
 *  grad_scale multiplies grads first (1/world_size after a sum all-reduce). `step` counts from 1. */
int mcedm_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                        size_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                        const double* sqnorm, double max_norm, double grad_scale, double ema_beta,
                        int64_t step, void* stream);

/* ---- kernel-level entry points ----------------------------------------------------------
 * The building blocks the schedules above are made of, exported so that each kernel can be
 * parity-tested against the oracle and timed on its own (bench.py roofline leg). */

/* Per-(sample, channel) input transform a conv applies while staging: v' = act((v-mean)*scale+offset). */
typedef struct { float mean, scale, offset, pad; } mcedm_coef;

/* Packed-weight size (floats) of a [Cout, Cin, k, k] conv, k in {1, 3}; bias_pk needs ceil32(Cout) floats. */
size_t mcedm_op_conv_packed_floats(int Cout, int Cin, int k);
/* Conv2d weights -> MFMA slab order.  qkv_heads > 0 re-orders output rows from the reference's
 * (head, c, {q,k,v}) interleave (adm_blocks.py:175) to (head, {q,k,v}, c).  dgrad != 0 packs the
 * transposed, tap-mirrored weights of the data gradient instead (then wpk has Cin output rows). */
int mcedm_op_pack_conv(const float* w, const float* b, int Cout, int Cin, int k, int qkv_heads, int dgrad,
                       float* wpk, float* bias_pk, void* stream);
/* GroupNorm statistics of cat(xa, xb) [B, Ca+Cb, HW] -> coef table [B][C] (models/adm_blocks.py:86-97
 * fused with the FiLM of :163-166 when film != NULL: row n = film[n*film_stride + (scale[0..C) | shift[C..2C))]).
 * groups = min(32, C/4).  stats_out [B][groups][2] (mean, rstd) may be NULL. */
int mcedm_op_gn_coef(const float* xa, const float* xb, int Ca, int Cb, int B, int HW, const float* gamma,
                     const float* beta, const float* film, int film_batch, int film_stride, float eps,
                     mcedm_coef* coef_out, float* stats_out, void* stream);
/* out = conv_k(resample(act(coef(cat(xa, xb))))) + bias + resample(res)   (models/adm_blocks.py:57-82 with the
 * pointwise ops of :161,166,171,179 fused).  resample / res_mode: 0 none, 1 nearest-2x up, 2 2x2-mean down.
 * resample 3 (k = 3 only, no residual): the conv itself has stride 2 over the source padded by one zero row / column at
 * the bottom / right -- the DDPM Downsample of models/ddim_blocks.py:85-104; then (Hs, Ws) = (2H, 2W).
 * (Hs, Ws) is the source size, (H, W) the conv size. */
int mcedm_op_conv(const float* xa, const float* xb, int Ca, int Cb, const mcedm_coef* coef, int coef_batch, int act,
                  int resample, int Hs, int Ws, int H, int W, const float* wpk, const float* bias_pk,
                  const float* res, int res_mode, float* out, int Cout, int B, int k, void* stream);
/* The same convolution (3x3, pad 1) in Winograd F(2x2, 3x3) form: 4/9 of the matrix instructions of mcedm_op_conv.
 * Serves Cout % 64 == 0, H % 8 == 0, W % 16 == 0, Ca % 8 == 0, Cb % 8 == 0, resample 0 (none) or 1 (nearest-2x up: the
 * source is [.., H/2, W/2]), res_mode 0, 1 (residual [.., H/2, W/2]) or 2 (2x2 mean of a residual [.., 2H, 2W]); results agree with mcedm_op_conv to fp32 rounding (a different
 * summation), not bit for bit.  w [Cout][Cin][3][3] -> wino (mcedm_op_conv_wino_packed_floats floats); bias [Cout] in
 * natural order or NULL; (H, W) is the conv (= output) size. */
size_t mcedm_op_conv_wino_packed_floats(int Cout, int Cin);
int mcedm_op_pack_conv_wino(const float* w, int Cout, int Cin, float* wino, void* stream);
int mcedm_op_conv_wino(const float* xa, const float* xb, int Ca, int Cb, const mcedm_coef* coef, int coef_batch, int act,
                       int resample, int H, int W, const float* wino, const float* bias, const float* res, int res_mode,
                       float* out, int Cout, int B, void* stream);
/* sigma-embedding MLP + every block's FiLM rows in one launch (models/adm_blocks.py:192-199 PositionalEmbedding,
 * :367-379 mapping MLP with SiLU, :143,163-165 per-block affine):  emb = silu(W1 silu(W0 pe(labels) + b0) + b1),
 * film[n] = Waff emb[n] + baff.  labels [n]; w0, w1 [ch][ch] (out, in); waff [rows][ch] = the blocks' affine weights
 * concatenated; freqs_scratch [ch/2]; emb_out [n][ch] (may be NULL); film_out [n][rows]. */
int mcedm_op_embedding(const float* labels, int n, int ch, const float* w0, const float* b0, const float* w1,
                       const float* b1, const float* waff, const float* baff, int rows, float* freqs_scratch,
                       float* emb_out, float* film_out, void* stream);
/* a = softmax(q^T k / sqrt(64)) v per (sample, head) (models/adm_blocks.py:103-109,174-178);
 * qkv is [B][heads][3][64][T] (the packed qkv conv's output), out [B][heads*64][T]. */
int mcedm_op_attention(const float* qkv, float* out, int B, int heads, int T, void* stream);
/* Backward building blocks.
 * conv weight / bias gradient: dw [Cout, Cin, k, k], db [Cout] (may be NULL) for the conv described as in mcedm_op_conv
 * (the transformed input is rebuilt into the scratch first); scratch holds mcedm_op_wgrad_scratch_floats() floats; qkv_heads > 0:
 * dy rows are in packed qkv order.  Data gradient = mcedm_op_conv on weights packed with dgrad = 1. */
size_t mcedm_op_wgrad_scratch_floats(int Cout, int Cin, int k, int B, int H, int W);
int mcedm_op_conv_wgrad(const float* dy, const float* xa, const float* xb, int Ca, int Cb, const mcedm_coef* coef,
                        int coef_batch, int act, int resample, int Hs, int Ws, int H, int W, int Cout, int B, int k,
                        int qkv_heads, float* scratch, float* dw, float* db, void* stream);
/* Backward of resample(act(film(group_norm(cat(xa, xb))))): dact is the gradient w.r.t. the conv input (conv
 * resolution), coef / stats come from mcedm_op_gn_coef of the forward.  Writes (or accumulates into) dxa / dxb, adds
 * `add` (add_mode 1: source resolution [B, C, Hs, Ws]; 2: conv resolution, mapped back through the resampling),
 * and returns dgamma, dbeta [C] and, when film != NULL, dfilm rows (d scale | d shift) with stride dfilm_stride.
 * ab is a [B][C][2] scratch. */
int mcedm_op_gn_bwd(const float* dact, int resample, const float* xa, const float* xb, int Ca, int Cb, int Hs, int Ws,
                    int B, const mcedm_coef* coef, const float* stats, const float* gamma, const float* beta,
                    const float* film, int film_batch, int film_stride, int act, float* dxa, float* dxb, int accumulate,
                    const float* add, int add_mode, float* ab, float* dgamma, float* dbeta, float* dfilm,
                    int dfilm_stride, void* stream);
/* The same with `sync`: MCEDM_GN_SYNC_WORDS * B * groups 32-bit words (groups = min(32, C / 4)), ZERO on entry and left
 * zero on exit.  With them a (sample, group) slab of 64 KB or more is cut into pieces of 4096 elements whose workgroups keep them in
 * LDS between the two passes and exchange their partial sums through `sync` (round 5; without it such slabs take the two-pass
 * kernel).  What the plan's backward (mcedm_denoise_backward) passes from its workspace. */
#define MCEDM_GN_SYNC_WORDS 130
int mcedm_op_gn_bwd_sync(const float* dact, int resample, const float* xa, const float* xb, int Ca, int Cb, int Hs, int Ws,
                         int B, const mcedm_coef* coef, const float* stats, const float* gamma, const float* beta,
                         const float* film, int film_batch, int film_stride, int act, float* dxa, float* dxb, int accumulate,
                         const float* add, int add_mode, float* ab, float* dgamma, float* dbeta, float* dfilm,
                         int dfilm_stride, unsigned int* sync, void* stream);
/* Attention backward (models/adm_blocks.py:111-118): qkv / dqkv packed [B][heads][3][64][T]; a, da [B][heads*64][T];
 * lse_scratch holds B*heads*T*2 floats. */
int mcedm_op_attention_bwd(const float* qkv, const float* a, const float* da, float* dqkv, float* lse_scratch, int B,
                           int heads, int T, void* stream);
/* Test hook: force the conv tile (channel tile mt in {32,64,128}, pixel tile ph x pw in {8x32,8x16,16x16,8x8};
 * (128,16,32) = the 8-wave kernel, 3x3 only);
 * (0,0,0) restores the size heuristic.  Process-global, not thread-safe. */
int mcedm_op_set_conv_tile(int mt, int ph, int pw);
/* Selects the experimental 8-wave conv kernel (one 512-thread workgroup per CU, double-buffered LDS slabs) for the
 * 3x3 layers large enough to give every CU a workgroup: 1 on, 0 off, -1 back to the default (env MCEDM_CONV8, else
 * off).  Results are bit-identical to the default kernel.  Process-global, not thread-safe. */
int mcedm_op_set_conv8(int enable);
/* The input-resident conv kernels that serve <= 32 x 32 images (conv_resident.hip: the tile's K extent in LDS by DMA,
 * weights streamed under the MFMAs): 1 on, 0 off (every conv takes conv_mfma_kernel), -1 back to the default (env
 * MCEDM_CONV_RESIDENT, else on).  Bit-identical to conv_mfma_kernel per tile configuration except the K-split tile used
 * at <= 8 x 8 (same result for a given shape, grouped differently).  Process-global, not thread-safe. */
int mcedm_op_set_conv_resident(int enable);
/* The Winograd F(2x2, 3x3) kernels (conv_wino.hip) in the network paths: 1 on, 0 off (direct kernels everywhere), -1 back to
 * the default (env MCEDM_WINOGRAD, else on).  Read when a plan lays out its workspace (whether a block's 1x1 skip projection
 * is folded into conv1 depends on it) and at every launch: set it before mcedm_unet_plan_create / the first forward.
 * Process-global, not thread-safe. */
int mcedm_op_set_conv_wino(int enable);
/* Which of the two Winograd kernels serves the 128-channel shapes: 1 = the one-wave-per-SIMD kernel (conv_wino1.hip: all 16
 * positions of a 32-channel block in one wave's AccVGPRs), 0 = the two-waves-per-SIMD kernel (conv_wino.hip), -1 = the default
 * (env MCEDM_WINO1, else 0: the one-wave kernel measured 6-9 % slower and stays off).  The two are bit-identical (statistics
 * included).  Read at every launch.  Process-global. */
int mcedm_op_set_conv_wino1(int enable);
/* The Winograd F(3x3, 2x2) weight-gradient kernel (wgrad_wino.hip) for the un-resampled 3x3 convs whose channel counts are
 * multiples of 128 on images with W % 32 == 0: 1 on, 0 off (the direct split-K kernel everywhere), -1 back to the default (env
 * MCEDM_WGRAD_WINO, else on).  Read at every launch; the scratch size does not depend on it.  Process-global. */
int mcedm_op_set_wgrad_wino(int enable);
/* The register-direct GEMM kernel (conv1x1_reg.hip) for 1x1 convs without input transform (the decoder blocks' skip projections and
 * their data gradients; Cout % 128 == 0, Cin % 16 == 0, Cin <= 256, H * W % 512 == 0): 1 on, 0 off (conv_mfma_kernel), -1 back to the
 * default (env MCEDM_CONV1X1_REG, else on).  Results differ from conv_mfma_kernel's in the last bits only through ... nothing: both sum
 * over ci in ascending pairs; tests hold them to rtol 1e-5.  Read at every launch.  Process-global. */
int mcedm_op_set_conv1x1_reg(int enable);
/* The single-launch attention part of a UNetBlock at 8 x 8 x 64 channels (attn_fused.hip; inference only): 1 on, 0 off
 * (qkv conv + attention kernel + proj conv), -1 back to the default (env MCEDM_ATTN_FUSED, else on).  Process-global. */
int mcedm_op_set_attn_fused(int enable);
/* Diagnostics: when buf != NULL every conv workgroup writes 16 x u64 at buf[16*blockIdx]: [0..3] timestamps (10 ns
 * units) at start / first chunk / end of K loop / end of epilogue, [4] (XCC id << 32 | HW_ID), [5..6] shader-clock
 * counter at the K loop's ends, [8..] per-phase cycle sums (MCEDM_CONV_TIMELINE builds) or per-wave HW_ID (8-wave
 * kernel).  NULL switches it off. */
int mcedm_op_set_conv_debug(unsigned long long* buf);

/* ---- RePaint-style EDM sampling on the DDPM U-Net ---------------------------------------
 * The joint-DDPM baseline PlDdim (models/ddim.py) sampled with the EDM Heun sampler and RePaint-style resampling:
 * `sample_edm` (ddim.py:959-1051: n_repeat inner loops per step, the known region re-noised to the current level
 * after every loop), `get_denoised` (:915-947, VP preconditioning c_skip 1, c_out -sigma, c_noise = the nearest DDPM
 * timestep), `round_sigma` (:949-957), `compute_alpha` (:700-704), on the ermongroup/ddim U-Net `Model`
 * (models/ddim_blocks.py:222-470; configs/model/ddim_res32.yaml).  Inference only, one noise level per call,
 * cond = None, x_self_cond = None (zeros), dx = None -- exactly what sample_edm evaluates. */
typedef struct {
  int32_t in_channels;       /* hparams.model.in_channels (state channels h_ch + u_ch), 2 */
  int32_t out_channels;      /* out_ch */
  int32_t ch;                /* base width (multiple of 32); temb width is 4 * ch */
  int32_t n_levels;
  int32_t ch_mult[MCEDM_MAX_LEVELS];
  int32_t num_res_blocks;
  int32_t resolution;        /* the network asserts input size == resolution (ddim_blocks.py:411); levels are resolution >> l */
  int32_t n_attn_resolutions;
  int32_t attn_resolutions[MCEDM_MAX_LEVELS];
  int32_t self_cond;         /* 1: conv_in takes cat(x_self_cond, x) (ddim_blocks.py:262, 366-370) */
  float eps;                 /* GroupNorm eps, 1e-6 (ddim_blocks.py:62-63) */
} mcedm_ddpm_desc;

/* configs/diff_sampler/edm_sampler_inv.yaml + the diffusion schedule tables PlDdim derives from `betas`
 * (host pointers, fp32 as the reference holds them): edm_steps[n] = get_edm_steps() (ddim.py:131-137, largest sigma
 * first), alphas_cumprod_ext[n + 1] = cumprod(1 - cat(0, betas)) (the table compute_alpha indexes with t + 1). */
typedef struct {
  int32_t timesteps;
  double sigma_min, sigma_max, rho;
  double S_churn, S_min, S_max, S_noise;
  double w;                  /* must be 0 on this path (cond is None) */
  int32_t n_repeat;          /* resampling loops per step */
  int32_t n_time_h, n_time_u;/* rows [0, n_time_*) of the h / u channels are KNOWN (conditioning) */
  int32_t h_ch, u_ch;
  int32_t num_diffusion_timesteps;
  const float* edm_steps;
  const float* alphas_cumprod_ext;
} mcedm_repaint_desc;

typedef struct mcedm_ddpm_plan mcedm_ddpm_plan;
int mcedm_ddpm_plan_create(const mcedm_ddpm_desc* desc, mcedm_ddpm_plan** out);
void mcedm_ddpm_plan_destroy(mcedm_ddpm_plan* plan);
int mcedm_ddpm_plan_set_variant(mcedm_ddpm_plan* plan, int which, int value);      /* as mcedm_unet_plan_set_variant */
/* Parameter table in Model.state_dict() order (names as the reference's: "temb.dense.0.weight", "down.0.block.0.norm1.weight", ...). */
int mcedm_ddpm_param_count(const mcedm_ddpm_plan* plan);
int mcedm_ddpm_param_info(const mcedm_ddpm_plan* plan, int index, const char** name, int64_t* numel, int32_t* ndim,
                          int64_t shape[4]);
int mcedm_ddpm_packed_bytes(const mcedm_ddpm_plan* plan, size_t* bytes);
/* temb_freqs: device array [ch / 2] = exp(arange(ch/2) * -(ln 10000 / (ch/2 - 1))) exactly as get_timestep_embedding
 * builds it (ddim_blocks.py:22-24); the caller owns that expression because t * freqs reaches ~1000 and a 1-ulp
 * difference in a frequency is a 1e-4 difference in sin / cos. */
int mcedm_ddpm_pack_weights(const mcedm_ddpm_plan* plan, const float* const* params, const float* temb_freqs, void* packed,
                            void* stream);
int mcedm_ddpm_workspace_bytes(const mcedm_ddpm_plan* plan, int B, size_t* bytes);
/* Model.forward(x, t) (ddim_blocks.py:410-470), one timestep t for the whole batch; x [B, in_channels, R, R]. */
int mcedm_ddpm_forward(const mcedm_ddpm_plan* plan, const void* packed, const float* x, float t, float* out, void* workspace,
                       size_t workspace_bytes, int B, void* stream);
/* PlDdim.get_denoised at a scalar sigma: D = x - sigma * F(x / sqrt(sigma^2 + 1), c_noise); c_noise is the timestep
 * num_timesteps - 1 - round_sigma(sigma, return_index) the caller (or mcedm_repaint_sample) derives.  F_out may be NULL. */
int mcedm_ddpm_denoise(const mcedm_ddpm_plan* plan, const void* packed, const float* x, float sigma, float c_noise,
                       float* D_out, float* F_out, void* workspace, size_t workspace_bytes, int B, void* stream);
/* Host helper: the rounded sigma schedule of ddim.py:982-986 (timesteps + 1 values, last = 0). */
int mcedm_repaint_schedule(const mcedm_repaint_desc* sp, double* t_steps);
/* PlDdim.sample_edm (ddim.py:959-1051), guide_dx False.
 *  hu           [B, C, R, R] fp32 normalised joint state ('b h w c' of cat(h, u) rearranged to NCHW, ddim.py:967-968)
 *  init_noise   [B, C, R, R] fp32            randn_like(hu) (:969), also the noise of every known-region re-noising
 *  step_noise   [timesteps][B, C, R, R] fp64  the per-step draw (:1004); may be NULL when no step raises sigma
 *  repeat_noise [timesteps][n_repeat - 1][B, C, R, R] fp64  the draw between inner loops (:1037); NULL iff n_repeat == 1
 *  out          fp64 [B, 1 or timesteps + 1, R, R, C] */
int mcedm_repaint_workspace_bytes(const mcedm_ddpm_plan* plan, int B, size_t* bytes);
int mcedm_repaint_sample(const mcedm_ddpm_plan* plan, const void* packed, const mcedm_repaint_desc* sp, const float* hu,
                         const float* init_noise, const double* step_noise, const double* repeat_noise, double* out,
                         int return_last, void* workspace, size_t workspace_bytes, int B, void* stream);
/* The same sampler with the per-step and per-loop noise GENERATED ON THE DEVICE instead of read from tensors (the
 * reference draws timesteps * n_repeat randn_like tensors per call, ddim.py:1004, 1037: 576 at BASELINE config 5):
 * Philox4x32-10 keyed by the 64-bit seed at *rng_seed (DEVICE memory, read by the kernels when they run, so one captured
 * HIP graph replays with fresh noise after the host rewrites the seed), counter = (element pair, draw index), Box-Muller
 * on 53-bit uniforms.  Draw index of step i: i * n_repeat (the :1004 draw), i * n_repeat + 1 + k (the :1037 draw after
 * inner loop k).  Statistically equivalent to, not the same stream as, torch.randn_like. */
int mcedm_repaint_sample_rng(const mcedm_ddpm_plan* plan, const void* packed, const mcedm_repaint_desc* sp, const float* hu,
                             const float* init_noise, const uint64_t* rng_seed, double* out, int return_last,
                             void* workspace, size_t workspace_bytes, int B, void* stream);
/* out[0 .. n) = the N(0, 1) values of draw `draw` of that generator (fp64). */
int mcedm_normal_fill(double* out, size_t n, const uint64_t* rng_seed, uint64_t draw, void* stream);

/* Model.forward(x, t, x_self_cond) with the self-conditioning tensor given (ddim_blocks.py:417-420; NULL = zeros, i.e.
 * mcedm_ddpm_forward).  x_self_cond [B, in_channels, R, R]. */
int mcedm_ddpm_forward_sc(const mcedm_ddpm_plan* plan, const void* packed, const float* x, const float* x_self_cond, float t,
                          float* out, void* workspace, size_t workspace_bytes, int B, void* stream);
/* PlDdim.sample_with_repeat (models/ddim.py:808-913): the DDIM sampler with RePaint-style inner loops, guide_dx False,
 * dx_cond False.  Everything is fp32 like the reference.
 *  timesteps, skip_type (0 uniform, 1 quad: ddim.py:823-830), eta, n_repeat, n_time_h / n_time_u / h_ch / u_ch as in
 *  mcedm_repaint_desc; alphas_cumprod_ext: host table cumprod(1 - cat(0, betas)) with num_diffusion_timesteps + 1 entries.
 *  hu, init_noise  [B, C, R, R]      normalised joint state; randn_like(hu) (:832)
 *  eta_noise       [timesteps][B, C, R, R] or NULL: the torch.rand_like draws of :893 (UNIFORM in the reference), eta != 0 only
 *  self_cond       1: the previous x0 prediction is fed back as x_self_cond (hparams.model.self_cond), 0: never
 *  xs_out, x0_out  fp32 [B, 1 or timesteps + 1, R, R, C] / [B, 1 or timesteps, R, R, C] ('b t h w c', as the reference returns) */
typedef struct mcedm_ddim_desc {
  int32_t timesteps, skip_type;
  double eta;
  int32_t n_repeat, n_time_h, n_time_u, h_ch, u_ch, num_diffusion_timesteps, self_cond;
  const float* alphas_cumprod_ext;
} mcedm_ddim_desc;
int mcedm_ddim_workspace_bytes(const mcedm_ddpm_plan* plan, int B, size_t* bytes);
/* Host helper: the timestep sequence the sampler walks (models/ddim.py:823-830), exactly as the reference builds it --
 * range(0, n, n // timesteps) for uniform, [int(s) for s in np.linspace(0, sqrt(0.8 n), timesteps) ** 2] for quad, with
 * numpy.linspace's own arithmetic (arange * step, last sample pinned to the stop value).  *count = entries (uniform may
 * exceed `timesteps`); seq may be NULL to query the count, else capacity >= *count. */
int mcedm_ddim_timesteps(int num_diffusion_timesteps, int timesteps, int skip_type, int* seq, int capacity, int* count);
int mcedm_ddim_repaint_sample(const mcedm_ddpm_plan* plan, const void* packed, const mcedm_ddim_desc* sp, const float* hu,
                              const float* init_noise, const float* eta_noise, float* xs_out, float* x0_out, int return_last,
                              void* workspace, size_t workspace_bytes, int B, void* stream);

/* ---- PDE residuals (forward) ----------------------------------------------
 * Replace the tensor-op bodies of models/pde_loss.py; results are bit-identical to the PyTorch CPU path.
 * All tensors are fp32, channel-last (b, t, x, 2) = (h, u) for SWE and (b, s, s, 2) = (a, u) for Darcy.
 *   half_dt = (float)(0.5 * Tn / n_times), dx = x[1] - x[0] of SweFvLoss.gen_x (models/pde_loss.py:102-118, fp32).
 * mcedm_swe_fv_step      SweFvLoss.f_t_swp1d      models/pde_loss.py:131-165   out (b, t, x, 2)
 * mcedm_swe_fv_residual  SweFvLoss.calculate_loss models/pde_loss.py:211-225 (+ clamp of forward, :245-247)
 *                        scale2_* = normalizer.divide ** 2 (get_scaling, :197-209); out (b, t, x, 2)
 * mcedm_darcy_residual   DarcyLoss.calculate_loss models/pde_loss.py:30-54 and the division / clamp of forward
 *                        (:80-86): two_dx = (float)(2 * D / s), denom = (s-4)^2; out (b, s-4, s-4) */
int mcedm_swe_fv_step(const float* s, float* out, int B, int T, int X, float half_dt, float dx, void* stream);
/* The return_d=True branches (the guidance gradients; analytic adjoints of the stencils, models/pde_loss.py:231-242 and
 * :60-75): mcedm_swe_fv_guidance = d mean(calculate_loss(pred, gt)) / d pred, out (b, t, x, 2);
 * mcedm_darcy_guidance = d mean(L) / d pred or, calc_prob != 0, d mean(log(2 (1 - sigmoid(1e5 L)) + 1e-12)) / d pred,
 * out (b, s, s, 2), scratch holds b * (s-4)^2 floats.  NaNs of the gradient are returned as 0 like the reference's. */
int mcedm_swe_fv_guidance(const float* pred, const float* gt, float* out, int B, int T, int X, float half_dt, float dx,
                          float scale2_h, float scale2_u, void* stream);
int mcedm_darcy_guidance(const float* pred, float* out, float* scratch, int B, int S, float two_dx, int calc_prob,
                         void* stream);
int mcedm_swe_fv_residual(const float* pred, const float* gt, float* out, int B, int T, int X, float half_dt, float dx,
                          float scale2_h, float scale2_u, int clamp, void* stream);
int mcedm_darcy_residual(const float* pred, float* out, int B, int S, float two_dx, float denom, int clamp, void* stream);

/* ---- measurement ---------------------------------------------------------------------------
 * Per-launch timing with HIP event pairs recorded on the launch stream (bench.py's roofline leg).
 * mcedm_prof_report waits for the events, then writes a JSON array
 *   [{"name", "launches", "total_ms", "flops", "bytes"}, ...]   (flops / bytes: algorithmic sums)
 * into buf and clears the records.  Not for use under graph capture. */
int mcedm_prof_enable(int on);
int mcedm_prof_report(char* buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* MCEDM_HIP_H_ */
