"""ctypes binding of libmcedm_hip.so (include/mcedm_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  Build it with ``python m-cedm_amd/build.py`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCEDM_LIB") or os.path.join(_HERE, "libmcedm_hip.so")   # MCEDM_LIB: A/B builds
MAX_LEVELS = 8

# every symbol include/mcedm_hip.h declares
EXPORTS = [
    "mcedm_version", "mcedm_last_error", "mcedm_unet_plan_create", "mcedm_unet_plan_destroy",
    "mcedm_unet_param_count", "mcedm_unet_param_info", "mcedm_unet_packed_bytes", "mcedm_unet_pack_weights",
    "mcedm_unet_workspace_bytes", "mcedm_unet_forward", "mcedm_edm_denoise", "mcedm_sampler_workspace_bytes",
    "mcedm_heun_sample", "mcedm_edm_t_steps", "mcedm_edm_loss", "mcedm_edm_noise_inputs",
    "mcedm_edm_denoise_backward", "mcedm_edm_denoise_backward_bucketed", "mcedm_unet_grad_buckets", "mcedm_sqnorm",
    "mcedm_adam_ema_step",
    "mcedm_swe_fv_step", "mcedm_swe_fv_residual", "mcedm_darcy_residual", "mcedm_swe_fv_guidance", "mcedm_darcy_guidance",
    "mcedm_heun_sample_guided",
    "mcedm_ddpm_plan_create", "mcedm_ddpm_plan_destroy", "mcedm_ddpm_param_count", "mcedm_ddpm_param_info",
    "mcedm_ddpm_packed_bytes", "mcedm_ddpm_pack_weights", "mcedm_ddpm_workspace_bytes", "mcedm_ddpm_forward",
    "mcedm_ddpm_denoise", "mcedm_repaint_schedule", "mcedm_repaint_workspace_bytes", "mcedm_repaint_sample",
    "mcedm_repaint_sample_rng", "mcedm_normal_fill", "mcedm_ddpm_forward_sc", "mcedm_ddim_workspace_bytes",
    "mcedm_ddim_repaint_sample", "mcedm_ddim_timesteps",
    "mcedm_unet_forward_dx", "mcedm_edm_denoise_dx", "mcedm_edm_denoise_backward_dx", "mcedm_heun_sample_dxcond",
    "mcedm_unet_plan_set_variant", "mcedm_ddpm_plan_set_variant", "mcedm_heun_sample_rng",
]
# kernel families that exist in two forms (include/mcedm_hip.h MCEDM_VARIANT_*)
GN_SYNC_WORDS = 130          # MCEDM_GN_SYNC_WORDS
VARIANTS = {"conv_wino": 0, "conv_wino1": 1, "conv_resident": 2, "conv8": 3, "attn_fused": 4, "wgrad_wino": 5, "conv1x1_reg": 6}


class UNetDesc(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("cond_channels", C.c_int32), ("out_channels", C.c_int32),
                ("ch", C.c_int32), ("n_levels", C.c_int32), ("ch_mult", C.c_int32 * MAX_LEVELS),
                ("num_res_blocks", C.c_int32), ("resolution", C.c_int32), ("n_attn_resolutions", C.c_int32),
                ("attn_resolutions", C.c_int32 * MAX_LEVELS), ("channels_per_head", C.c_int32), ("eps", C.c_float),
                ("dx_channels", C.c_int32), ("dx_mode", C.c_int32)]


DX_NONE, DX_CAT, DX_ENC = 0, 1, 2      # MCEDM_DX_* (include/mcedm_hip.h)
ABI_VERSION = 4                        # MCEDM_ABI_VERSION
REDUCE_SCRATCH_BYTES = 4096 * 8 + 64   # MCEDM_REDUCE_SCRATCH_BYTES


class SamplerDesc(C.Structure):
    _fields_ = [("timesteps", C.c_int32), ("sigma_min", C.c_double), ("sigma_max", C.c_double), ("rho", C.c_double),
                ("S_churn", C.c_double), ("S_min", C.c_double), ("S_max", C.c_double), ("S_noise", C.c_double),
                ("w", C.c_double), ("sigma_data", C.c_double), ("net_sigma_min", C.c_double),
                ("net_sigma_max", C.c_double)]


class GuidanceDesc(C.Structure):
    _fields_ = [("system", C.c_int32), ("half_dt", C.c_float), ("dx", C.c_float), ("two_dx", C.c_float), ("sub_h", C.c_float),
                ("div_h", C.c_float), ("sub_u", C.c_float), ("div_u", C.c_float), ("weight", C.c_double)]


class DdpmDesc(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("ch", C.c_int32), ("n_levels", C.c_int32),
                ("ch_mult", C.c_int32 * MAX_LEVELS), ("num_res_blocks", C.c_int32), ("resolution", C.c_int32),
                ("n_attn_resolutions", C.c_int32), ("attn_resolutions", C.c_int32 * MAX_LEVELS), ("self_cond", C.c_int32),
                ("eps", C.c_float)]


class RepaintDesc(C.Structure):
    _fields_ = [("timesteps", C.c_int32), ("sigma_min", C.c_double), ("sigma_max", C.c_double), ("rho", C.c_double),
                ("S_churn", C.c_double), ("S_min", C.c_double), ("S_max", C.c_double), ("S_noise", C.c_double),
                ("w", C.c_double), ("n_repeat", C.c_int32), ("n_time_h", C.c_int32), ("n_time_u", C.c_int32),
                ("h_ch", C.c_int32), ("u_ch", C.c_int32), ("num_diffusion_timesteps", C.c_int32),
                ("edm_steps", C.POINTER(C.c_float)), ("alphas_cumprod_ext", C.POINTER(C.c_float))]


class DdimDesc(C.Structure):
    _fields_ = [("timesteps", C.c_int32), ("skip_type", C.c_int32), ("eta", C.c_double), ("n_repeat", C.c_int32),
                ("n_time_h", C.c_int32), ("n_time_u", C.c_int32), ("h_ch", C.c_int32), ("u_ch", C.c_int32),
                ("num_diffusion_timesteps", C.c_int32), ("self_cond", C.c_int32), ("alphas_cumprod_ext", C.POINTER(C.c_float))]


_lib = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python m-cedm_amd/build.py` "
                           "(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32p, f64p, sz = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t
    lib.mcedm_version.restype = C.c_int
    if lib.mcedm_version() != ABI_VERSION:           # struct layouts below are those of include/mcedm_hip.h at this version
        raise RuntimeError(f"{LIB_PATH} implements ABI {lib.mcedm_version()}, this binding ABI {ABI_VERSION}: rebuild it with "
                           "`python m-cedm_amd/build.py`")
    lib.mcedm_last_error.restype = C.c_char_p
    lib.mcedm_unet_plan_create.argtypes = [C.POINTER(UNetDesc), C.POINTER(vp)]
    lib.mcedm_unet_plan_destroy.argtypes = [vp]
    lib.mcedm_unet_plan_destroy.restype = None
    lib.mcedm_unet_param_count.argtypes = [vp]
    lib.mcedm_unet_param_info.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int64 * 4)]
    lib.mcedm_unet_packed_bytes.argtypes = [vp, C.POINTER(sz)]
    lib.mcedm_unet_pack_weights.argtypes = [vp, C.POINTER(vp), vp, vp]
    lib.mcedm_unet_workspace_bytes.argtypes = [vp, i32, i32, i32, i32, C.POINTER(sz)]
    lib.mcedm_unet_forward.argtypes = [vp, vp, f32p, f32p, f32p, f32p, i32, f32p, vp, sz, i32, i32, i32, i32, vp]
    lib.mcedm_edm_denoise.argtypes = [vp, vp, f32p, f32p, i32, f32p, f32p, f32p, vp, sz, i32, i32, i32, i32,
                                      C.c_double, vp]
    lib.mcedm_sampler_workspace_bytes.argtypes = [vp, i32, i32, i32, C.POINTER(sz)]
    lib.mcedm_heun_sample.argtypes = [vp, vp, C.POINTER(SamplerDesc), f32p, f32p, f32p, f64p, f64p, i32, vp, sz,
                                      i32, i32, i32, vp]
    lib.mcedm_heun_sample_rng.argtypes = [vp, vp, C.POINTER(SamplerDesc), f32p, f32p, f32p, vp, f64p, i32, vp, sz,
                                          i32, i32, i32, vp]
    lib.mcedm_unet_plan_set_variant.argtypes = [vp, i32, i32]
    lib.mcedm_ddpm_plan_set_variant.argtypes = [vp, i32, i32]
    lib.mcedm_edm_t_steps.argtypes = [C.POINTER(SamplerDesc), C.POINTER(C.c_double)]
    lib.mcedm_edm_loss.argtypes = [f32p, f32p, f32p, f32p, i32, i32, i32, i32, C.c_double, f32p, f32p, vp, sz, vp]
    lib.mcedm_edm_noise_inputs.argtypes = [f32p, f32p, f32p, f32p, i32, i32, i32, i32, C.c_double, C.c_double, f32p,
                                           f32p, vp]
    lib.mcedm_edm_denoise_backward.argtypes = [vp, vp, C.POINTER(vp), f32p, f32p, i32, f32p, f32p, C.POINTER(vp), vp,
                                               sz, i32, i32, i32, C.c_double, vp]
    lib.mcedm_edm_denoise_backward_bucketed.argtypes = [vp, vp, C.POINTER(vp), f32p, f32p, i32, f32p, f32p, C.POINTER(vp),
                                                        vp, sz, i32, i32, i32, C.c_double, i32, C.POINTER(C.c_int32),
                                                        C.POINTER(vp), vp]
    lib.mcedm_unet_grad_buckets.argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int)]
    lib.mcedm_unet_forward_dx.argtypes = [vp, vp, f32p, f32p, f32p, f32p, f32p, i32, f32p, vp, sz, i32, i32, i32, i32, vp]
    lib.mcedm_edm_denoise_dx.argtypes = [vp, vp, f32p, f32p, f32p, i32, f32p, f32p, f32p, vp, sz, i32, i32, i32, i32,
                                         C.c_double, vp]
    lib.mcedm_edm_denoise_backward_dx.argtypes = [vp, vp, C.POINTER(vp), f32p, f32p, f32p, i32, f32p, f32p, C.POINTER(vp),
                                                  vp, sz, i32, i32, i32, C.c_double, i32, C.POINTER(C.c_int32),
                                                  C.POINTER(vp), vp]
    lib.mcedm_heun_sample_dxcond.argtypes = [vp, vp, C.POINTER(SamplerDesc), C.POINTER(GuidanceDesc), C.POINTER(GuidanceDesc),
                                             f32p, f32p, f64p, f64p, i32, vp, sz, i32, i32, i32, vp]
    lib.mcedm_sqnorm.argtypes = [f32p, sz, f64p, vp, sz, vp]
    lib.mcedm_adam_ema_step.argtypes = [f32p, f32p, f32p, f32p, f32p, sz, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.c_double, f64p, C.c_double, C.c_double, C.c_double, C.c_int64, vp]
    lib.mcedm_swe_fv_step.argtypes = [f32p, f32p, i32, i32, i32, C.c_float, C.c_float, vp]
    lib.mcedm_swe_fv_residual.argtypes = [f32p, f32p, f32p, i32, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float,
                                          i32, vp]
    lib.mcedm_darcy_residual.argtypes = [f32p, f32p, i32, i32, C.c_float, C.c_float, i32, vp]
    lib.mcedm_swe_fv_guidance.argtypes = [f32p, f32p, f32p, i32, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    lib.mcedm_darcy_guidance.argtypes = [f32p, f32p, f32p, i32, i32, C.c_float, i32, vp]
    lib.mcedm_heun_sample_guided.argtypes = [vp, vp, C.POINTER(SamplerDesc), C.POINTER(GuidanceDesc), f32p, f32p, f32p, f64p,
                                             f64p, i32, vp, sz, i32, i32, i32, vp]
    lib.mcedm_ddpm_plan_create.argtypes = [C.POINTER(DdpmDesc), C.POINTER(vp)]
    lib.mcedm_ddpm_plan_destroy.argtypes = [vp]
    lib.mcedm_ddpm_plan_destroy.restype = None
    lib.mcedm_ddpm_param_count.argtypes = [vp]
    lib.mcedm_ddpm_param_info.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int64 * 4)]
    lib.mcedm_ddpm_packed_bytes.argtypes = [vp, C.POINTER(sz)]
    lib.mcedm_ddpm_pack_weights.argtypes = [vp, C.POINTER(vp), f32p, vp, vp]
    lib.mcedm_ddpm_workspace_bytes.argtypes = [vp, i32, C.POINTER(sz)]
    lib.mcedm_ddpm_forward.argtypes = [vp, vp, f32p, C.c_float, f32p, vp, sz, i32, vp]
    lib.mcedm_ddpm_denoise.argtypes = [vp, vp, f32p, C.c_float, C.c_float, f32p, f32p, vp, sz, i32, vp]
    lib.mcedm_repaint_schedule.argtypes = [C.POINTER(RepaintDesc), C.POINTER(C.c_double)]
    lib.mcedm_repaint_workspace_bytes.argtypes = [vp, i32, C.POINTER(sz)]
    lib.mcedm_repaint_sample.argtypes = [vp, vp, C.POINTER(RepaintDesc), f32p, f32p, f64p, f64p, f64p, i32, vp, sz, i32, vp]
    lib.mcedm_repaint_sample_rng.argtypes = [vp, vp, C.POINTER(RepaintDesc), f32p, f32p, vp, f64p, i32, vp, sz, i32, vp]
    lib.mcedm_normal_fill.argtypes = [f64p, sz, vp, C.c_uint64, vp]
    lib.mcedm_ddpm_forward_sc.argtypes = [vp, vp, f32p, f32p, C.c_float, f32p, vp, sz, i32, vp]
    lib.mcedm_ddim_workspace_bytes.argtypes = [vp, i32, C.POINTER(sz)]
    lib.mcedm_ddim_repaint_sample.argtypes = [vp, vp, C.POINTER(DdimDesc), f32p, f32p, f32p, f32p, f32p, i32, vp, sz, i32, vp]
    lib.mcedm_ddim_timesteps.argtypes = [i32, i32, i32, C.POINTER(C.c_int), i32, C.POINTER(C.c_int)]
    for name in EXPORTS:
        fn = getattr(lib, name)          # AttributeError here == header/library drift
        if name not in ("mcedm_last_error", "mcedm_unet_plan_destroy", "mcedm_ddpm_plan_destroy"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mcedm_last_error().decode(errors="replace")
        raise RuntimeError(f"libmcedm_hip {what} failed ({rc}): {msg}")


def _ptr(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libmcedm_hip needs device tensors (got a CPU tensor); there is no CPU fallback")
    if t.dtype != dtype:
        raise RuntimeError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("expected a contiguous tensor")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def sampler_desc(sp, sigma_data=1.0, net_sigma_min=0.002, net_sigma_max=80.0) -> SamplerDesc:
    """Build the C sampler description from the reference's ``sparams`` (attribute access, DictConfig or DotDict)."""
    return SamplerDesc(int(sp.timesteps), float(sp.sigma_min), float(sp.sigma_max), float(sp.rho), float(sp.S_churn),
                       float(sp.S_min), float(sp.S_max), float(sp.S_noise), float(sp.w), float(sigma_data),
                       float(net_sigma_min), float(net_sigma_max))


def edm_t_steps(sd: SamplerDesc) -> List[float]:
    arr = (C.c_double * (sd.timesteps + 1))()
    check(load().mcedm_edm_t_steps(C.byref(sd), arr), "edm_t_steps")
    return list(arr)


class Workspace:
    """Grow-only byte buffer on one device (the library never allocates device memory itself)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = None
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


class Plan:
    """Host-side handle of one network architecture (mcedm_unet_plan_create)."""

    def __init__(self, in_channels: int, cond_channels: int, out_channels: int, ch: int, ch_mult: Sequence[int],
                 num_res_blocks: int, attn_resolutions: Sequence[int], resolution: int, channels_per_head: int = 64,
                 eps: float = 1e-5, dx_channels: int = 0, dx_mode: int = DX_NONE):
        lib = load()
        if len(ch_mult) > MAX_LEVELS or len(attn_resolutions) > MAX_LEVELS:
            raise RuntimeError("too many levels / attention resolutions")
        d = UNetDesc()
        d.in_channels, d.cond_channels, d.out_channels, d.ch = in_channels, cond_channels, out_channels, ch
        d.n_levels = len(ch_mult)
        for i, m in enumerate(ch_mult):
            d.ch_mult[i] = int(m)
        d.num_res_blocks, d.resolution = num_res_blocks, resolution
        d.n_attn_resolutions = len(attn_resolutions)
        for i, r in enumerate(attn_resolutions):
            d.attn_resolutions[i] = int(r)
        d.channels_per_head, d.eps = channels_per_head, eps
        d.dx_channels, d.dx_mode = int(dx_channels), int(dx_mode)
        self.dx_channels, self.dx_mode = int(dx_channels), int(dx_mode)
        self.desc = d
        h = C.c_void_p()
        check(lib.mcedm_unet_plan_create(C.byref(d), C.byref(h)), "plan_create")
        self._h = h
        self._lib = lib
        n = lib.mcedm_unet_param_count(h)
        self.param_names: List[str] = []
        self.param_shapes: List[tuple] = []
        for i in range(n):
            name, numel, ndim, shape = C.c_char_p(), C.c_int64(), C.c_int32(), (C.c_int64 * 4)()
            check(lib.mcedm_unet_param_info(h, i, C.byref(name), C.byref(numel), C.byref(ndim), C.byref(shape)))
            self.param_names.append(name.value.decode())
            self.param_shapes.append(tuple(shape[j] for j in range(ndim.value)))
        sz = C.c_size_t()
        check(lib.mcedm_unet_packed_bytes(h, C.byref(sz)))
        self.packed_bytes = sz.value
        self.in_channels, self.cond_channels, self.out_channels = in_channels, cond_channels, out_channels

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.mcedm_unet_plan_destroy(h)

    def set_variant(self, which: str, value: int = -1) -> None:
        """This plan's own choice for one kernel family (``VARIANTS``): 1 / 0, -1 = the process default (mcedm_op_set_* or the
        environment).  In force whenever one of THIS plan's entry points runs; other plans are not affected.  'conv_wino'
        also shapes the workspace layout: set it before the plan's first use."""
        self._lib.mcedm_unet_plan_set_variant.argtypes = [C.c_void_p, C.c_int, C.c_int]
        check(self._lib.mcedm_unet_plan_set_variant(self._h, VARIANTS[which], int(value)), "plan_set_variant")

    # ---- derived weights ---------------------------------------------------------------
    def pack(self, params: Dict[str, torch.Tensor], packed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Pack the named fp32 device parameters (keys = DhariwalUNet.state_dict() names)."""
        tens = []
        for name, shape in zip(self.param_names, self.param_shapes):
            t = params[name]
            if tuple(t.shape) != shape:
                raise RuntimeError(f"parameter {name}: shape {tuple(t.shape)} != {shape}")
            tens.append(t.detach())
        dev = tens[0].device
        if packed is None:
            packed = torch.empty(self.packed_bytes, dtype=torch.uint8, device=dev)
        arr = (C.c_void_p * len(tens))(*[_ptr(t) for t in tens])
        check(self._lib.mcedm_unet_pack_weights(self._h, arr, packed.data_ptr(), _stream()), "pack_weights")
        return packed

    # ---- sizes -----------------------------------------------------------------------------
    def workspace_bytes(self, B: int, H: int, W: int, training: bool = False) -> int:
        sz = C.c_size_t()
        check(self._lib.mcedm_unet_workspace_bytes(self._h, B, H, W, int(training), C.byref(sz)), "workspace_bytes")
        return sz.value

    def sampler_workspace_bytes(self, B: int, H: int, W: int) -> int:
        sz = C.c_size_t()
        check(self._lib.mcedm_sampler_workspace_bytes(self._h, B, H, W, C.byref(sz)), "sampler_workspace_bytes")
        return sz.value

    # ---- compute ---------------------------------------------------------------------------
    def _check_dx(self, x, dx):
        if dx is None:
            return
        if self.dx_mode == DX_NONE:
            raise RuntimeError("dx given to a plan without dx_cond")
        if tuple(dx.shape) != (x.shape[0], self.dx_channels, x.shape[2], x.shape[3]) or dx.dtype != torch.float32:
            raise RuntimeError(f"dx must be fp32 [B, {self.dx_channels}, H, W], got {tuple(dx.shape)} {dx.dtype}")

    def forward(self, packed, x, noise_labels, cond=None, x_scale=None, ws: Optional[Workspace] = None,
                training: bool = False, dx=None) -> torch.Tensor:
        self._check_dx(x, dx)
        B, _, H, W = x.shape
        n_noise = noise_labels.numel()
        ws = ws or Workspace()
        need = self.workspace_bytes(B, H, W, training)
        buf = ws.get(need, x.device)
        out = torch.empty((B, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        check(self._lib.mcedm_unet_forward_dx(self._h, packed.data_ptr(), _ptr(x), _ptr(dx), _ptr(cond), _ptr(x_scale),
                                              _ptr(noise_labels), n_noise, _ptr(out), buf.data_ptr(), buf.numel(), B, H, W,
                                              int(training), _stream()), "unet_forward")
        return out

    def denoise(self, packed, x, sigma, cond=None, ws: Optional[Workspace] = None, training: bool = False,
                sigma_data: float = 1.0, want_F: bool = False, dx=None):
        self._check_dx(x, dx)
        B, _, H, W = x.shape
        n_sigma = sigma.numel()
        ws = ws or Workspace()
        buf = ws.get(self.workspace_bytes(B, H, W, training), x.device)
        D = torch.empty((B, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        F = torch.empty_like(D) if want_F else None
        check(self._lib.mcedm_edm_denoise_dx(self._h, packed.data_ptr(), _ptr(x), _ptr(dx), _ptr(sigma), n_sigma, _ptr(cond),
                                             _ptr(D), _ptr(F), buf.data_ptr(), buf.numel(), B, H, W, int(training),
                                             float(sigma_data), _stream()), "edm_denoise")
        return (D, F) if want_F else D

    def grad_buckets(self, max_buckets: int) -> List[int]:
        """First parameter index of each gradient bucket, in the order the backward completes them (last one is 0)."""
        arr = (C.c_int32 * max(1, max_buckets))()
        n = C.c_int()
        check(self._lib.mcedm_unet_grad_buckets(self._h, int(max_buckets), arr, C.byref(n)), "grad_buckets")
        return [arr[i] for i in range(n.value)]

    def denoise_backward(self, packed, params: Dict[str, torch.Tensor], x, sigma, cond, dD, grads: Sequence[torch.Tensor],
                         ws: Workspace, sigma_data: float = 1.0, bucket_first: Optional[Sequence[int]] = None,
                         bucket_events: Optional[Sequence[torch.cuda.Event]] = None, dx=None) -> None:
        """Backward of denoise(..., training=True) on the SAME workspace: grads[i] <- dLoss/dparam_i (overwritten).
        With bucket_first / bucket_events the library records event k once every parameter >= bucket_first[k] is done."""
        self._check_dx(x, dx)
        B, _, H, W = x.shape
        buf = ws.get(self.workspace_bytes(B, H, W, True), x.device)
        parr = (C.c_void_p * len(self.param_names))(*[_ptr(params[n].detach()) for n in self.param_names])
        garr = (C.c_void_p * len(self.param_names))(*[_ptr(g) for g in grads])
        if dx is not None and bucket_first is None:
            check(self._lib.mcedm_edm_denoise_backward_dx(self._h, packed.data_ptr(), parr, _ptr(x), _ptr(dx), _ptr(sigma),
                                                          sigma.numel(), _ptr(cond), _ptr(dD), garr, buf.data_ptr(), buf.numel(),
                                                          B, H, W, float(sigma_data), 0, None, None, _stream()),
                  "edm_denoise_backward_dx")
            return
        if bucket_first is None:
            check(self._lib.mcedm_edm_denoise_backward(self._h, packed.data_ptr(), parr, _ptr(x), _ptr(sigma), sigma.numel(),
                                                       _ptr(cond), _ptr(dD), garr, buf.data_ptr(), buf.numel(), B, H, W,
                                                       float(sigma_data), _stream()), "edm_denoise_backward")
            return
        nb = len(bucket_first)
        firsts = (C.c_int32 * nb)(*[int(f) for f in bucket_first])
        handles = [int(e.cuda_event) for e in bucket_events]
        if len(handles) != nb or not all(handles):
            raise RuntimeError("denoise_backward: one created (recorded at least once) torch.cuda.Event per bucket is needed")
        evs = (C.c_void_p * nb)(*handles)
        check(self._lib.mcedm_edm_denoise_backward_dx(self._h, packed.data_ptr(), parr, _ptr(x), _ptr(dx), _ptr(sigma),
                                                      sigma.numel(), _ptr(cond), _ptr(dD), garr, buf.data_ptr(),
                                                      buf.numel(), B, H, W, float(sigma_data), nb, firsts, evs,
                                                      _stream()), "edm_denoise_backward_dx")

    def sample(self, packed, sd: SamplerDesc, cond, mask, init_noise, step_noise=None, return_last: bool = True,
               ws: Optional[Workspace] = None, out: Optional[torch.Tensor] = None,
               guidance: Optional["GuidanceDesc"] = None, dx_input: Optional["GuidanceDesc"] = None,
               rng_seed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dx_input: the residual whose gradient at the current state is the network's dx input (dx_cond plans).
        rng_seed: a one-element int64 DEVICE tensor -- the churn noise of every step is then generated inside the kernel that
        applies it (mcedm_heun_sample_rng) instead of being read from step_noise [N, B, C, H, W] float64."""
        B, _, H, W = init_noise.shape
        ws = ws or Workspace()
        buf = ws.get(self.sampler_workspace_bytes(B, H, W), init_noise.device)
        T = 1 if return_last else sd.timesteps + 1
        if out is None:
            out = torch.empty((B, T, H, W, self.in_channels), dtype=torch.float64, device=init_noise.device)
        elif tuple(out.shape) != (B, T, H, W, self.in_channels):
            raise RuntimeError(f"sample: out has shape {tuple(out.shape)}, expected {(B, T, H, W, self.in_channels)}")
        if dx_input is not None:
            if mask is not None:
                raise RuntimeError("sample: dx_cond sampling is the unmasked single-task sampler")
            check(self._lib.mcedm_heun_sample_dxcond(self._h, packed.data_ptr(), C.byref(sd), C.byref(dx_input),
                                                     C.byref(guidance) if guidance is not None else None, _ptr(cond),
                                                     _ptr(init_noise), _ptr(step_noise, torch.float64), _ptr(out, torch.float64),
                                                     int(return_last), buf.data_ptr(), buf.numel(), B, H, W, _stream()),
                  "heun_sample_dxcond")
            return out
        if guidance is not None:
            check(self._lib.mcedm_heun_sample_guided(self._h, packed.data_ptr(), C.byref(sd), C.byref(guidance), _ptr(cond),
                                                     _ptr(mask), _ptr(init_noise), _ptr(step_noise, torch.float64),
                                                     _ptr(out, torch.float64), int(return_last), buf.data_ptr(), buf.numel(),
                                                     B, H, W, _stream()), "heun_sample_guided")
            return out
        if rng_seed is not None:
            if step_noise is not None:
                raise RuntimeError("sample: give step_noise (materialised draws) or rng_seed (device-side draws), not both")
            if rng_seed.dtype != torch.int64 or rng_seed.numel() != 1 or rng_seed.device != init_noise.device:
                raise RuntimeError("sample: rng_seed must be a one-element int64 tensor on the sampler's device")
            check(self._lib.mcedm_heun_sample_rng(self._h, packed.data_ptr(), C.byref(sd), _ptr(cond), _ptr(mask),
                                                  _ptr(init_noise), rng_seed.data_ptr(), _ptr(out, torch.float64),
                                                  int(return_last), buf.data_ptr(), buf.numel(), B, H, W, _stream()),
                  "heun_sample_rng")
            return out
        check(self._lib.mcedm_heun_sample(self._h, packed.data_ptr(), C.byref(sd), _ptr(cond), _ptr(mask),
                                          _ptr(init_noise), _ptr(step_noise, torch.float64), _ptr(out, torch.float64),
                                          int(return_last), buf.data_ptr(), buf.numel(), B, H, W, _stream()),
              "heun_sample")
        return out


def repaint_desc(sp, edm_steps: torch.Tensor, alphas_ext: torch.Tensor, h_ch: int, u_ch: int):
    """C description of PlDdim.sample_edm's parameters (configs/diff_sampler/edm_sampler_inv.yaml) + the schedule tables
    (host fp32 tensors: get_edm_steps(), cumprod(1 - cat(0, betas))).  Returns (desc, keepalive)."""
    es = edm_steps.detach().to("cpu", torch.float32).contiguous()
    ae = alphas_ext.detach().to("cpu", torch.float32).contiguous()
    if ae.numel() != es.numel() + 1:
        raise RuntimeError("alphas_ext must have one more entry than edm_steps")
    d = RepaintDesc(int(sp.timesteps), float(sp.sigma_min), float(sp.sigma_max), float(sp.rho), float(sp.S_churn),
                    float(sp.S_min), float(sp.S_max), float(sp.S_noise), float(sp.w), int(sp.n_repeat), int(sp.n_time_h),
                    int(sp.n_time_u), int(h_ch), int(u_ch), int(es.numel()),
                    C.cast(es.data_ptr(), C.POINTER(C.c_float)), C.cast(ae.data_ptr(), C.POINTER(C.c_float)))
    return d, (es, ae)


def ddim_desc(sp, alphas_ext: torch.Tensor, h_ch: int, u_ch: int, self_cond: bool):
    """C description of PlDdim.sample_with_repeat's parameters (configs/diff_sampler/ddim_sampler*.yaml).  Returns (desc, keepalive)."""
    ae = alphas_ext.detach().to("cpu", torch.float32).contiguous()
    skip = {"uniform": 0, "quad": 1}.get(str(sp.skip_type))
    if skip is None:
        raise NotImplementedError(f"skip_type {sp.skip_type}")             # models/ddim.py:829-830
    d = DdimDesc(int(sp.timesteps), skip, float(sp.eta), int(sp.n_repeat), int(sp.n_time_h), int(sp.n_time_u), int(h_ch), int(u_ch),
                 int(ae.numel() - 1), int(bool(self_cond)), C.cast(ae.data_ptr(), C.POINTER(C.c_float)))
    return d, ae


def ddim_timesteps(num_diffusion_timesteps: int, timesteps: int, skip_type) -> List[int]:
    """The timestep sequence of PlDdim.sample_with_repeat (models/ddim.py:823-830) as the device sampler walks it."""
    skip = {"uniform": 0, "quad": 1, 0: 0, 1: 1}.get(skip_type)
    if skip is None:
        raise NotImplementedError(f"skip_type {skip_type}")
    cnt = C.c_int()
    check(load().mcedm_ddim_timesteps(num_diffusion_timesteps, timesteps, skip, None, 0, C.byref(cnt)), "ddim_timesteps")
    arr = (C.c_int * cnt.value)()
    check(load().mcedm_ddim_timesteps(num_diffusion_timesteps, timesteps, skip, arr, cnt.value, C.byref(cnt)), "ddim_timesteps")
    return list(arr)


def repaint_schedule(rd: RepaintDesc) -> List[float]:
    arr = (C.c_double * (rd.timesteps + 1))()
    check(load().mcedm_repaint_schedule(C.byref(rd), arr), "repaint_schedule")
    return list(arr)


class DdpmPlan:
    """Host-side handle of one DDPM U-Net (models/ddim_blocks.py Model; mcedm_ddpm_plan_create)."""

    def __init__(self, in_channels: int, out_channels: int, ch: int, ch_mult: Sequence[int], num_res_blocks: int,
                 attn_resolutions: Sequence[int], resolution: int, self_cond: bool = True, eps: float = 1e-6):
        lib = load()
        if len(ch_mult) > MAX_LEVELS or len(attn_resolutions) > MAX_LEVELS:
            raise RuntimeError("too many levels / attention resolutions")
        d = DdpmDesc()
        d.in_channels, d.out_channels, d.ch, d.n_levels = in_channels, out_channels, ch, len(ch_mult)
        for i, m in enumerate(ch_mult):
            d.ch_mult[i] = int(m)
        d.num_res_blocks, d.resolution, d.n_attn_resolutions = num_res_blocks, resolution, len(attn_resolutions)
        for i, r in enumerate(attn_resolutions):
            d.attn_resolutions[i] = int(r)
        d.self_cond, d.eps = int(bool(self_cond)), eps
        self.desc = d
        h = C.c_void_p()
        check(lib.mcedm_ddpm_plan_create(C.byref(d), C.byref(h)), "ddpm_plan_create")
        self._h, self._lib = h, lib
        self.param_names: List[str] = []
        self.param_shapes: List[tuple] = []
        for i in range(lib.mcedm_ddpm_param_count(h)):
            name, numel, ndim, shape = C.c_char_p(), C.c_int64(), C.c_int32(), (C.c_int64 * 4)()
            check(lib.mcedm_ddpm_param_info(h, i, C.byref(name), C.byref(numel), C.byref(ndim), C.byref(shape)))
            self.param_names.append(name.value.decode())
            self.param_shapes.append(tuple(shape[j] for j in range(ndim.value)))
        sz = C.c_size_t()
        check(lib.mcedm_ddpm_packed_bytes(h, C.byref(sz)))
        self.packed_bytes = sz.value
        self.in_channels, self.out_channels, self.resolution, self.ch = in_channels, out_channels, resolution, ch

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.mcedm_ddpm_plan_destroy(h)

    def set_variant(self, which: str, value: int = -1) -> None:
        """As ``Plan.set_variant``: this plan's own kernel choice, -1 = the process default."""
        self._lib.mcedm_ddpm_plan_set_variant.argtypes = [C.c_void_p, C.c_int, C.c_int]
        check(self._lib.mcedm_ddpm_plan_set_variant(self._h, VARIANTS[which], int(value)), "ddpm_plan_set_variant")

    def pack(self, params: Dict[str, torch.Tensor], temb_freqs: torch.Tensor, packed: Optional[torch.Tensor] = None):
        """params keyed like Model.state_dict(); temb_freqs [ch/2] device fp32, built by the caller exactly as
        get_timestep_embedding does (models/ddim_blocks.py:22-24)."""
        tens = []
        for name, shape in zip(self.param_names, self.param_shapes):
            t = params[name]
            if tuple(t.shape) != shape:
                raise RuntimeError(f"parameter {name}: shape {tuple(t.shape)} != {shape}")
            tens.append(t.detach())
        if temb_freqs.numel() != self.ch // 2:
            raise RuntimeError("temb_freqs must have ch / 2 entries")
        if packed is None:
            packed = torch.empty(self.packed_bytes, dtype=torch.uint8, device=tens[0].device)
        arr = (C.c_void_p * len(tens))(*[_ptr(t) for t in tens])
        check(self._lib.mcedm_ddpm_pack_weights(self._h, arr, _ptr(temb_freqs), packed.data_ptr(), _stream()), "ddpm_pack_weights")
        return packed

    def workspace_bytes(self, B: int) -> int:
        sz = C.c_size_t()
        check(self._lib.mcedm_ddpm_workspace_bytes(self._h, B, C.byref(sz)), "ddpm_workspace_bytes")
        return sz.value

    def repaint_workspace_bytes(self, B: int) -> int:
        sz = C.c_size_t()
        check(self._lib.mcedm_repaint_workspace_bytes(self._h, B, C.byref(sz)), "repaint_workspace_bytes")
        return sz.value

    def _check_x(self, x):
        if tuple(x.shape[1:]) != (self.in_channels, self.resolution, self.resolution):
            raise RuntimeError(f"input {tuple(x.shape)} != [B, {self.in_channels}, {self.resolution}, {self.resolution}] "
                               "(the network asserts input size == resolution, ddim_blocks.py:411)")

    def forward(self, packed, x, t: float, ws: Optional[Workspace] = None, x_self_cond: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._check_x(x)
        B = x.shape[0]
        ws = ws or Workspace()
        buf = ws.get(self.workspace_bytes(B), x.device)
        out = torch.empty((B, self.out_channels, self.resolution, self.resolution), dtype=torch.float32, device=x.device)
        if x_self_cond is not None:
            if tuple(x_self_cond.shape) != tuple(x.shape):
                raise RuntimeError("x_self_cond must have the shape of x")
            check(self._lib.mcedm_ddpm_forward_sc(self._h, packed.data_ptr(), _ptr(x), _ptr(x_self_cond), float(t), _ptr(out),
                                                  buf.data_ptr(), buf.numel(), B, _stream()), "ddpm_forward_sc")
            return out
        check(self._lib.mcedm_ddpm_forward(self._h, packed.data_ptr(), _ptr(x), float(t), _ptr(out), buf.data_ptr(),
                                           buf.numel(), B, _stream()), "ddpm_forward")
        return out

    def ddim_workspace_bytes(self, B: int) -> int:
        sz = C.c_size_t()
        check(self._lib.mcedm_ddim_workspace_bytes(self._h, B, C.byref(sz)), "ddim_workspace_bytes")
        return sz.value

    def ddim_repaint_sample(self, packed, dd: "DdimDesc", hu, init_noise, eta_noise=None, return_last: bool = True,
                            ws: Optional[Workspace] = None):
        """PlDdim.sample_with_repeat on the device -> (xs, x0_preds), both fp32 'b t h w c'."""
        self._check_x(hu)
        B = hu.shape[0]
        ws = ws or Workspace()
        buf = ws.get(self.ddim_workspace_bytes(B), hu.device)
        n = dd.num_diffusion_timesteps
        S = len(ddim_timesteps(n, dd.timesteps, dd.skip_type))
        R = self.resolution
        xs = torch.empty((B, 1 if return_last else S + 1, R, R, self.in_channels), dtype=torch.float32, device=hu.device)
        x0 = torch.empty((B, 1 if return_last else S, R, R, self.in_channels), dtype=torch.float32, device=hu.device)
        check(self._lib.mcedm_ddim_repaint_sample(self._h, packed.data_ptr(), C.byref(dd), _ptr(hu), _ptr(init_noise),
                                                  _ptr(eta_noise), _ptr(xs), _ptr(x0), int(return_last), buf.data_ptr(),
                                                  buf.numel(), B, _stream()), "ddim_repaint_sample")
        return xs, x0

    def denoise(self, packed, x, sigma: float, c_noise: float, ws: Optional[Workspace] = None, want_F: bool = False):
        self._check_x(x)
        B = x.shape[0]
        ws = ws or Workspace()
        buf = ws.get(self.workspace_bytes(B), x.device)
        D = torch.empty((B, self.out_channels, self.resolution, self.resolution), dtype=torch.float32, device=x.device)
        F = torch.empty_like(D) if want_F else None
        check(self._lib.mcedm_ddpm_denoise(self._h, packed.data_ptr(), _ptr(x), float(sigma), float(c_noise), _ptr(D), _ptr(F),
                                           buf.data_ptr(), buf.numel(), B, _stream()), "ddpm_denoise")
        return (D, F) if want_F else D

    def repaint_sample(self, packed, rd: RepaintDesc, hu, init_noise, step_noise=None, repeat_noise=None,
                       return_last: bool = True, ws: Optional[Workspace] = None, out: Optional[torch.Tensor] = None,
                       rng_seed: Optional[torch.Tensor] = None):
        """rng_seed (device int64 [1]): the per-step / per-loop noise is generated on the device from that seed
        (mcedm_repaint_sample_rng) instead of being read from step_noise / repeat_noise."""
        self._check_x(hu)
        B = hu.shape[0]
        ws = ws or Workspace()
        buf = ws.get(self.repaint_workspace_bytes(B), hu.device)
        T = 1 if return_last else rd.timesteps + 1
        shape = (B, T, self.resolution, self.resolution, self.in_channels)
        if out is None:
            out = torch.empty(shape, dtype=torch.float64, device=hu.device)
        elif tuple(out.shape) != shape:
            raise RuntimeError(f"repaint_sample: out has shape {tuple(out.shape)}, expected {shape}")
        if rng_seed is not None:
            if step_noise is not None or repeat_noise is not None:
                raise RuntimeError("repaint_sample: give either noise tensors or rng_seed")
            check(self._lib.mcedm_repaint_sample_rng(self._h, packed.data_ptr(), C.byref(rd), _ptr(hu), _ptr(init_noise),
                                                     _ptr(rng_seed, torch.int64), _ptr(out, torch.float64), int(return_last),
                                                     buf.data_ptr(), buf.numel(), B, _stream()), "repaint_sample_rng")
            return out
        check(self._lib.mcedm_repaint_sample(self._h, packed.data_ptr(), C.byref(rd), _ptr(hu), _ptr(init_noise),
                                             _ptr(step_noise, torch.float64), _ptr(repeat_noise, torch.float64),
                                             _ptr(out, torch.float64), int(return_last), buf.data_ptr(), buf.numel(), B,
                                             _stream()), "repaint_sample")
        return out


def normal_fill(out: torch.Tensor, rng_seed: torch.Tensor, draw: int) -> torch.Tensor:
    """out (fp64, contiguous) <- draw number `draw` of the device generator keyed by rng_seed (int64 [1] on the device)."""
    check(load().mcedm_normal_fill(_ptr(out, torch.float64), out.numel(), _ptr(rng_seed, torch.int64), int(draw), _stream()),
          "normal_fill")
    return out


class _PinnedWorkspace:
    """Workspace view with a FIXED buffer: a captured graph bakes the pointer in, so the buffer must neither move nor be
    freed while the graph lives (the owner of the graph holds this object)."""

    def __init__(self, ws: Optional[Workspace], nbytes: int, device):
        self.buf = (ws or Workspace()).get(nbytes, device)

    def get(self, nbytes: int, device) -> torch.Tensor:
        if nbytes > self.buf.numel() or self.buf.device != torch.device(device):
            raise RuntimeError("graphed call: workspace request differs from the captured one")
        return self.buf


def _capture(run, dev) -> "torch.cuda.CUDAGraph":
    """Warm ``run`` up once off the capture (lazily initialised library state settles), then capture it into a HIP graph.
    capture_error_mode='thread_local': HIP calls of OTHER threads (DataLoader pin-memory thread, RCCL watchdog, logger
    hooks) during the long capture do not invalidate it."""
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        run()
    return graph


class GraphedSampler:
    """The whole Heun sampling call (every U-Net evaluation and state update of mcedm_heun_sample: ~4000 launches at
    18 steps) captured ONCE into a HIP graph and replayed.  The library never allocates or synchronises and the sigma
    schedule is host-side arithmetic baked into kernel arguments, so a replay is exact; inputs are copied into static
    buffers first.  Shapes, sampler parameters and the packed-weight buffer are fixed per instance (re-packing weights
    in place into the same buffer is fine).  ``ws``: the caller's workspace to borrow (replays and eager calls of one
    module are serial on one stream, so one buffer serves both); the instance keeps the buffer it captured with alive."""

    def __init__(self, plan: "Plan", packed: torch.Tensor, sd: SamplerDesc, B: int, H: int, W: int, masked: bool = True,
                 has_cond: bool = True, churn: bool = False, return_last: bool = True, ws: Optional[Workspace] = None,
                 guidance: Optional["GuidanceDesc"] = None, dx_input: Optional["GuidanceDesc"] = None, device_noise: bool = False):
        """churn with device_noise: the per-step draws are generated inside the sampler's kernels from ``self.seed`` (an int64
        device scalar the call rewrites before each replay) -- no [N, B, C, H, W] float64 buffer, fresh noise per replay."""
        dev = packed.device
        self.plan, self.packed, self.sd, self.return_last = plan, packed, sd, return_last
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev) if (churn and device_noise) else None
        self.guidance, self.dx_input = guidance, dx_input      # host-side descriptions, baked into the captured kernel arguments
        C = plan.in_channels
        self.cond = torch.zeros((B, plan.cond_channels, H, W), device=dev) if has_cond else None
        self.mask = torch.zeros((B, C, H, W), device=dev) if masked else None
        self.init = torch.zeros((B, C, H, W), device=dev)
        self.step_noise = (torch.zeros((sd.timesteps, B, C, H, W), dtype=torch.float64, device=dev)
                           if churn and not device_noise else None)
        T = 1 if return_last else sd.timesteps + 1
        self.out = torch.empty((B, T, H, W, C), dtype=torch.float64, device=dev)
        self.ws = _PinnedWorkspace(ws, plan.sampler_workspace_bytes(B, H, W), dev)
        self.graph = _capture(self._run, dev)

    def _run(self):
        self.plan.sample(self.packed, self.sd, self.cond, self.mask, self.init, self.step_noise, self.return_last, self.ws,
                         out=self.out, guidance=self.guidance, dx_input=self.dx_input, rng_seed=self.seed)

    def __call__(self, cond, mask, init_noise, step_noise=None, seed=None) -> torch.Tensor:
        """Returns the instance's static output tensor (overwritten by the next call).  seed (device-noise instances): python
        int or int64 tensor, the key of this call's churn draws."""
        if (self.seed is None) != (seed is None):
            raise RuntimeError("GraphedSampler: 'seed' goes with device_noise=True instances (and only with them)")
        if seed is not None:
            if torch.is_tensor(seed):
                self.seed.copy_(seed.reshape(1))
            else:
                self.seed.fill_(int(seed))
        for dst, src, name in ((self.cond, cond, "cond"), (self.mask, mask, "mask"), (self.init, init_noise, "init_noise"),
                               (self.step_noise, step_noise, "step_noise")):
            if (dst is None) != (src is None):
                raise RuntimeError(f"GraphedSampler: '{name}' presence differs from the captured call")
            if dst is not None:
                dst.copy_(src)
        self.graph.replay()
        return self.out


class GraphedRepaint:
    """mcedm_repaint_sample_rng captured once and replayed: the whole RePaint call (timesteps x n_repeat Heun updates:
    ~80 000 launches at BASELINE config 5) is one HIP graph.  The noise is generated on the device from the seed in
    ``self.seed``, which the host rewrites before each replay, so every replay draws fresh noise."""

    def __init__(self, plan: "DdpmPlan", packed: torch.Tensor, rd: RepaintDesc, keep, B: int, return_last: bool = True,
                 ws: Optional[Workspace] = None):
        dev = packed.device
        self.plan, self.packed, self.rd, self._keep, self.return_last = plan, packed, rd, keep, return_last
        S, Cc = plan.resolution, plan.in_channels
        self.hu = torch.zeros((B, Cc, S, S), device=dev)
        self.init = torch.zeros((B, Cc, S, S), device=dev)
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
        T = 1 if return_last else rd.timesteps + 1
        self.out = torch.empty((B, T, S, S, Cc), dtype=torch.float64, device=dev)
        self.ws = _PinnedWorkspace(ws, plan.repaint_workspace_bytes(B), dev)
        self.graph = _capture(self._run, dev)

    def _run(self):
        self.plan.repaint_sample(self.packed, self.rd, self.hu, self.init, None, None, self.return_last, self.ws, out=self.out,
                                 rng_seed=self.seed)

    def __call__(self, hu, init_noise, seed) -> torch.Tensor:
        """seed: python int or int64 tensor.  Returns the instance's static output tensor."""
        self.hu.copy_(hu)
        self.init.copy_(init_noise)
        if torch.is_tensor(seed):
            self.seed.copy_(seed.reshape(1))
        else:
            self.seed.fill_(int(seed))
        self.graph.replay()
        return self.out


def graphed_or_eager(cache: dict, key, build, eager, max_entries: int = 2):
    """Replay the cached graph for ``key`` (building it with ``build()`` on first use, at most ``max_entries`` kept, oldest
    evicted) and return ``fn`` such that fn(*args) runs the call; if the capture fails (another thread's HIP call under a
    global capture mode, an out-of-memory while the static buffers are made) the entry is dropped, the failure is
    remembered for this key and ``eager`` is returned instead: the evaluation loop goes on without a graph."""
    hit = cache.get(key)
    if hit is not None:
        return hit if hit != "eager" else eager
    while len(cache) >= max_entries:
        cache.pop(next(iter(cache)))
    try:
        cache[key] = build()
    except (RuntimeError, torch.cuda.OutOfMemoryError) as e:
        import warnings
        warnings.warn(f"HIP-graph capture of the sampling call failed ({str(e)[:200]}); running it eagerly")
        torch.cuda.synchronize()
        cache[key] = "eager"
        return eager
    return cache[key]


# ---- flat-buffer training helpers -------------------------------------------------------------
def edm_noise_inputs(x, mask, noise, rnd_normal, P_mean=-1.2, P_std=1.2):
    B, Cc, H, W = x.shape
    x_noise = torch.empty_like(x)
    sigma = torch.empty(B, dtype=torch.float32, device=x.device)
    check(load().mcedm_edm_noise_inputs(_ptr(x), _ptr(mask), _ptr(noise), _ptr(rnd_normal), B, Cc, H, W, P_mean, P_std,
                                        _ptr(x_noise), _ptr(sigma), _stream()), "edm_noise_inputs")
    return x_noise, sigma


_RED_SCRATCH = {}


def reduce_scratch(device, stream=None) -> torch.Tensor:
    """MCEDM_REDUCE_SCRATCH_BYTES of device memory for the fixed-order sums of edm_loss / sqnorm, one area per (device, stream):
    the library owns no device state (ABI 3), and calls on different streams must not share an area."""
    if torch.cuda.is_current_stream_capturing():      # belongs to the graph being captured (its private pool), never cached
        return torch.empty(REDUCE_SCRATCH_BYTES, dtype=torch.uint8, device=device)
    key = (torch.device(device).index, int(_stream() or 0) if stream is None else int(stream))
    buf = _RED_SCRATCH.get(key)
    if buf is None:
        buf = _RED_SCRATCH[key] = torch.empty(REDUCE_SCRATCH_BYTES, dtype=torch.uint8, device=device)
    return buf


def edm_loss(D, x, mask, sigma, sigma_data=1.0, want_grad=True, scratch: Optional[torch.Tensor] = None):
    B, Cc, H, W = D.shape
    loss = torch.empty(1, dtype=torch.float32, device=D.device)
    dD = torch.empty_like(D) if want_grad else None
    scratch = reduce_scratch(D.device) if scratch is None else scratch
    check(load().mcedm_edm_loss(_ptr(D), _ptr(x), _ptr(mask), _ptr(sigma), B, Cc, H, W, float(sigma_data), _ptr(loss),
                                _ptr(dD), scratch.data_ptr(), scratch.numel() * scratch.element_size(), _stream()), "edm_loss")
    return loss, dD


# ---- PDE residuals (models/pde_loss.py) -------------------------------------------------------
def swe_fv_step(s_t: torch.Tensor, half_dt: float, dx: float) -> torch.Tensor:
    """SweFvLoss.f_t_swp1d on (b, t, x, 2) fp32 states."""
    B, T, X, _ = s_t.shape
    out = torch.empty_like(s_t)
    check(load().mcedm_swe_fv_step(_ptr(s_t), _ptr(out), B, T, X, half_dt, dx, _stream()), "swe_fv_step")
    return out


def swe_fv_residual(pred, gt, half_dt: float, dx: float, scale2_h: float, scale2_u: float, clamp: bool) -> torch.Tensor:
    B, T, X, _ = pred.shape
    out = torch.empty_like(pred)
    check(load().mcedm_swe_fv_residual(_ptr(pred), _ptr(gt), _ptr(out), B, T, X, half_dt, dx, scale2_h, scale2_u, int(clamp),
                                       _stream()), "swe_fv_residual")
    return out


def swe_fv_guidance(pred, gt, half_dt: float, dx: float, scale2_h: float, scale2_u: float) -> torch.Tensor:
    """SweFvLoss.forward(return_d=True): d mean(residual) / d pred on (b, t, x, 2) fp32 states."""
    B, T, X, _ = pred.shape
    out = torch.empty_like(pred)
    check(load().mcedm_swe_fv_guidance(_ptr(pred), _ptr(gt), _ptr(out), B, T, X, half_dt, dx, scale2_h, scale2_u, _stream()),
          "swe_fv_guidance")
    return out


def darcy_guidance(pred, two_dx: float, calc_prob: bool) -> torch.Tensor:
    B, S = pred.shape[0], pred.shape[1]
    out = torch.empty_like(pred)
    scratch = torch.empty(B * (S - 4) * (S - 4), dtype=torch.float32, device=pred.device)
    check(load().mcedm_darcy_guidance(_ptr(pred), _ptr(out), _ptr(scratch), B, S, two_dx, int(calc_prob), _stream()),
          "darcy_guidance")
    return out


def darcy_residual(pred, two_dx: float, denom: float, clamp: bool) -> torch.Tensor:
    B, S = pred.shape[0], pred.shape[1]
    out = torch.empty((B, S - 4, S - 4), dtype=torch.float32, device=pred.device)
    check(load().mcedm_darcy_residual(_ptr(pred), _ptr(out), B, S, two_dx, denom, int(clamp), _stream()), "darcy_residual")
    return out


def sqnorm(g: torch.Tensor, out: Optional[torch.Tensor] = None, scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty(1, dtype=torch.float64, device=g.device)
    scratch = reduce_scratch(g.device) if scratch is None else scratch
    check(load().mcedm_sqnorm(_ptr(g), g.numel(), _ptr(out, torch.float64), scratch.data_ptr(),
                              scratch.numel() * scratch.element_size(), _stream()), "sqnorm")
    return out


def adam_ema_step(param, grad, exp_avg, exp_avg_sq, ema, step, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8,
                  weight_decay=0.0, sqnorm_t=None, max_norm=1.0, grad_scale=1.0, ema_beta=0.999):
    check(load().mcedm_adam_ema_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(ema), param.numel(),
                                     lr, beta1, beta2, eps, weight_decay, _ptr(sqnorm_t, torch.float64), max_norm,
                                     grad_scale, ema_beta, int(step), _stream()), "adam_ema_step")


# ---- kernel-level ops (tests, per-kernel timing) ---------------------------------------------------
_OPS_BOUND = False


def _bind_ops():
    global _OPS_BOUND
    lib = load()
    if _OPS_BOUND:
        return lib
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.mcedm_op_conv_packed_floats.argtypes = [i32, i32, i32]
    lib.mcedm_op_conv_packed_floats.restype = sz
    lib.mcedm_op_pack_conv.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.mcedm_op_gn_coef.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp]
    lib.mcedm_op_conv.argtypes = [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, i32,
                                  i32, i32, vp]
    lib.mcedm_op_attention.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.mcedm_op_embedding.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    lib.mcedm_op_wgrad_scratch_floats.argtypes = [i32, i32, i32, i32, i32, i32]
    lib.mcedm_op_wgrad_scratch_floats.restype = sz
    lib.mcedm_op_conv_wgrad.argtypes = [vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                        vp, vp, vp, vp]
    lib.mcedm_op_gn_bwd.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp,
                                    i32, vp, i32, vp, vp, vp, vp, i32, vp]
    lib.mcedm_op_gn_bwd_sync.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp,
                                         i32, vp, i32, vp, vp, vp, vp, i32, vp, vp]
    lib.mcedm_op_attention_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.mcedm_op_conv_wino_packed_floats.argtypes = [i32, i32]
    lib.mcedm_op_conv_wino_packed_floats.restype = sz
    lib.mcedm_op_pack_conv_wino.argtypes = [vp, i32, i32, vp, vp]
    lib.mcedm_op_conv_wino.argtypes = [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, i32, i32, vp]
    for n in ("mcedm_op_pack_conv", "mcedm_op_gn_coef", "mcedm_op_conv", "mcedm_op_attention", "mcedm_op_embedding", "mcedm_op_conv_wgrad",
              "mcedm_op_gn_bwd", "mcedm_op_gn_bwd_sync", "mcedm_op_attention_bwd", "mcedm_op_pack_conv_wino", "mcedm_op_conv_wino"):
        getattr(lib, n).restype = C.c_int
    _OPS_BOUND = True
    return lib


OP_EXPORTS = ["mcedm_op_conv_packed_floats", "mcedm_op_pack_conv", "mcedm_op_gn_coef", "mcedm_op_conv",
              "mcedm_op_attention", "mcedm_op_set_conv_tile", "mcedm_prof_enable", "mcedm_prof_report",
              "mcedm_op_wgrad_scratch_floats", "mcedm_op_conv_wgrad", "mcedm_op_gn_bwd", "mcedm_op_attention_bwd",
              "mcedm_op_set_conv_debug", "mcedm_op_set_conv8", "mcedm_op_set_conv_resident", "mcedm_op_set_attn_fused", "mcedm_op_embedding",
              "mcedm_op_conv_wino_packed_floats", "mcedm_op_pack_conv_wino", "mcedm_op_conv_wino", "mcedm_op_set_conv_wino", "mcedm_op_set_conv_wino1",
              "mcedm_op_set_wgrad_wino", "mcedm_op_set_conv1x1_reg", "mcedm_op_gn_bwd_sync"]


def prof_enable(on: bool) -> None:
    lib = load()
    lib.mcedm_prof_enable.argtypes = [C.c_int]
    check(lib.mcedm_prof_enable(int(on)), "prof_enable")


def prof_report() -> list:
    """Kernel-level timing rows collected since prof_enable(True) (synchronises the recorded events)."""
    import json
    lib = load()
    lib.mcedm_prof_report.argtypes = [C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(1 << 16)
    check(lib.mcedm_prof_report(buf, len(buf)), "prof_report")
    return json.loads(buf.value.decode())


def set_conv_tile(mt: int = 0, ph: int = 0, pw: int = 0) -> None:
    """Test hook: force the conv tile configuration; () restores the heuristic."""
    lib = _bind_ops()
    lib.mcedm_op_set_conv_tile.argtypes = [C.c_int, C.c_int, C.c_int]
    check(lib.mcedm_op_set_conv_tile(mt, ph, pw), "set_conv_tile")
def set_conv8(enable: int = -1) -> None:
    """Select the experimental 8-wave conv kernel (1 / 0; -1 = default)."""
    lib = _bind_ops()
    lib.mcedm_op_set_conv8.argtypes = [C.c_int]
    check(lib.mcedm_op_set_conv8(int(enable)), "set_conv8")


def set_conv_wino(enable: int = -1) -> None:
    """Winograd F(2x2, 3x3) kernels in the network paths: 1 / 0; -1 = default (on).  Set before the workspace is laid out."""
    lib = _bind_ops()
    lib.mcedm_op_set_conv_wino.argtypes = [C.c_int]
    check(lib.mcedm_op_set_conv_wino(int(enable)), "set_conv_wino")


def set_conv_wino1(enable: int = -1) -> None:
    """Which Winograd kernel serves the 128-channel shapes: 1 = one wave per SIMD (conv_wino1.hip), 0 = two (conv_wino.hip);
    -1 = default (0: the one-wave kernel is 6-9 % slower).  Bit-identical results either way."""
    lib = _bind_ops()
    lib.mcedm_op_set_conv_wino1.argtypes = [C.c_int]
    check(lib.mcedm_op_set_conv_wino1(int(enable)), "set_conv_wino1")


def set_conv1x1_reg(enable: int = -1) -> None:
    """Register-direct GEMM kernel for un-transformed 1x1 convs (conv1x1_reg.hip): 1 / 0 (conv_mfma_kernel); -1 = default (on)."""
    lib = _bind_ops()
    lib.mcedm_op_set_conv1x1_reg.argtypes = [C.c_int]
    check(lib.mcedm_op_set_conv1x1_reg(int(enable)), "set_conv1x1_reg")


def set_wgrad_wino(enable: int = -1) -> None:
    """Winograd F(3x3, 2x2) weight-gradient kernel (wgrad_wino.hip): 1 / 0 (direct split-K kernel everywhere); -1 = default (on)."""
    lib = _bind_ops()
    lib.mcedm_op_set_wgrad_wino.argtypes = [C.c_int]
    check(lib.mcedm_op_set_wgrad_wino(int(enable)), "set_wgrad_wino")


def set_conv_resident(enable: int = -1) -> None:
    """Input-resident conv kernels for <= 32 x 32 images: 1 / 0; -1 = default (on)."""
    lib = _bind_ops()
    lib.mcedm_op_set_conv_resident.argtypes = [C.c_int]
    check(lib.mcedm_op_set_conv_resident(int(enable)), "set_conv_resident")


def set_attn_fused(enable: int = -1) -> None:
    """Single-launch attention block at 8 x 8 x 64 (inference): 1 / 0; -1 = default (on)."""
    lib = _bind_ops()
    lib.mcedm_op_set_attn_fused.argtypes = [C.c_int]
    check(lib.mcedm_op_set_attn_fused(int(enable)), "set_attn_fused")


RS_NONE, RS_UP, RS_DOWN, RS_S2 = 0, 1, 2, 3


def op_pack_conv(w: torch.Tensor, b: Optional[torch.Tensor], qkv_heads: int = 0, dgrad: bool = False):
    lib = _bind_ops()
    Cout, Cin, k, _ = w.shape
    rows, cols = (Cin, Cout) if dgrad else (Cout, Cin)
    n = lib.mcedm_op_conv_packed_floats(rows, cols, k)
    wpk = torch.empty(n, dtype=torch.float32, device=w.device)
    bpk = torch.zeros((Cout + 31) // 32 * 32, dtype=torch.float32, device=w.device) if b is not None else None
    check(lib.mcedm_op_pack_conv(_ptr(w), _ptr(b), Cout, Cin, k, qkv_heads, int(dgrad), _ptr(wpk), _ptr(bpk), _stream()),
          "op_pack_conv")
    return wpk, bpk


def op_gn_coef(xa, xb, gamma, beta, film=None, film_batch=0, film_stride=0, eps=1e-5, want_stats=False):
    lib = _bind_ops()
    B, Ca = xa.shape[:2]
    Cb = xb.shape[1] if xb is not None else 0
    HW = xa[0, 0].numel()
    Ct = Ca + Cb
    coef = torch.empty((B, Ct, 4), dtype=torch.float32, device=xa.device)
    stats = torch.empty((B, min(32, Ct // 4), 2), dtype=torch.float32, device=xa.device) if want_stats else None
    check(lib.mcedm_op_gn_coef(_ptr(xa), _ptr(xb), Ca, Cb, B, HW, _ptr(gamma), _ptr(beta), _ptr(film), film_batch,
                               film_stride, eps, _ptr(coef), _ptr(stats), _stream()), "op_gn_coef")
    return (coef, stats) if want_stats else coef


def op_conv(xa, xb, wpk, bias_pk, Cout, k, coef=None, coef_batch=1, act=0, resample=RS_NONE, res=None,
            res_mode=RS_NONE, out=None):
    lib = _bind_ops()
    B, Ca, Hs, Ws = xa.shape
    Cb = xb.shape[1] if xb is not None else 0
    H, W = (Hs * 2, Ws * 2) if resample == RS_UP else ((Hs // 2, Ws // 2) if resample in (RS_DOWN, RS_S2) else (Hs, Ws))
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=xa.device)
    check(lib.mcedm_op_conv(_ptr(xa), _ptr(xb), Ca, Cb, _ptr(coef), coef_batch, act, resample, Hs, Ws, H, W, _ptr(wpk),
                            _ptr(bias_pk), _ptr(res), res_mode, _ptr(out), Cout, B, k, _stream()), "op_conv")
    return out


def op_pack_conv_wino(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> the Winograd F(2x2, 3x3) weight table of mcedm_op_conv_wino."""
    lib = _bind_ops()
    Cout, Cin = w.shape[:2]
    wino = torch.empty(lib.mcedm_op_conv_wino_packed_floats(Cout, Cin), dtype=torch.float32, device=w.device)
    check(lib.mcedm_op_pack_conv_wino(_ptr(w), Cout, Cin, _ptr(wino), _stream()), "op_pack_conv_wino")
    return wino


def op_conv_wino(xa, xb, wino, bias, Cout, coef=None, coef_batch=1, act=0, resample=RS_NONE, res=None, res_mode=RS_NONE,
                 out=None):
    lib = _bind_ops()
    B, Ca, Hs, Ws = xa.shape
    Cb = xb.shape[1] if xb is not None else 0
    H, W = (Hs * 2, Ws * 2) if resample == RS_UP else (Hs, Ws)
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=xa.device)
    check(lib.mcedm_op_conv_wino(_ptr(xa), _ptr(xb), Ca, Cb, _ptr(coef), coef_batch, act, resample, H, W, _ptr(wino),
                                 _ptr(bias), _ptr(res), res_mode, _ptr(out), Cout, B, _stream()), "op_conv_wino")
    return out


def op_embedding(labels, w0, b0, w1, b1, waff, baff):
    """-> (emb [n, ch], film [n, rows]): sigma-embedding MLP + the concatenated per-block affine rows (K6)."""
    lib = _bind_ops()
    n, ch, rows = labels.numel(), w0.shape[0], waff.shape[0]
    freqs = torch.empty(ch // 2, dtype=torch.float32, device=labels.device)
    emb = torch.empty((n, ch), dtype=torch.float32, device=labels.device)
    film = torch.empty((n, rows), dtype=torch.float32, device=labels.device)
    check(lib.mcedm_op_embedding(_ptr(labels), n, ch, _ptr(w0), _ptr(b0), _ptr(w1), _ptr(b1), _ptr(waff), _ptr(baff), rows,
                                 _ptr(freqs), _ptr(emb), _ptr(film), _stream()), "op_embedding")
    return emb, film


def op_attention(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    """qkv [B, heads*3*64, H, W] in packed (head, {q,k,v}, c) channel order -> [B, heads*64, H, W]."""
    lib = _bind_ops()
    B, C3, H, W = qkv.shape
    out = torch.empty((B, C3 // 3, H, W), dtype=torch.float32, device=qkv.device)
    check(lib.mcedm_op_attention(_ptr(qkv), _ptr(out), B, heads, H * W, _stream()), "op_attention")
    return out


def op_conv_wgrad(dy, xa, xb, k, coef=None, coef_batch=1, act=0, resample=RS_NONE, qkv_heads=0):
    """-> (dw [Cout, Cin, k, k], db [Cout])"""
    lib = _bind_ops()
    B, Cout, H, W = dy.shape
    Ca, Hs, Ws = xa.shape[1], xa.shape[2], xa.shape[3]
    Cb = xb.shape[1] if xb is not None else 0
    Cin = Ca + Cb
    scratch = torch.empty(lib.mcedm_op_wgrad_scratch_floats(Cout, Cin, k, B, H, W), dtype=torch.float32, device=dy.device)
    dw = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=dy.device)
    db = torch.empty(Cout, dtype=torch.float32, device=dy.device)
    check(lib.mcedm_op_conv_wgrad(_ptr(dy), _ptr(xa), _ptr(xb), Ca, Cb, _ptr(coef), coef_batch, act, resample, Hs, Ws, H,
                                  W, Cout, B, k, qkv_heads, _ptr(scratch), _ptr(dw), _ptr(db), _stream()), "op_conv_wgrad")
    return dw, db


def op_gn_bwd(dact, xa, xb, coef, stats, gamma, beta, film=None, film_batch=0, film_stride=0, act=1,
              resample=RS_NONE, add=None, add_mode=0, dx_init=None, sync=None):
    """-> (dxa, dxb, dgamma, dbeta, dfilm or None); dx_init (tuple) makes the call accumulate into copies of it.  sync: a zeroed
    int32 tensor of GN_SYNC_WORDS * B * groups words (mcedm_op_gn_bwd_sync: slabs split over workgroups keep their pieces in LDS)."""
    lib = _bind_ops()
    B, Ca, Hs, Ws = xa.shape
    Cb = xb.shape[1] if xb is not None else 0
    Ct = Ca + Cb
    dxa = dx_init[0].clone() if dx_init else torch.empty_like(xa)
    dxb = (dx_init[1].clone() if dx_init else torch.empty_like(xb)) if xb is not None else None
    ab = torch.empty((B, Ct, 2), dtype=torch.float32, device=xa.device)
    dg, dbt = torch.empty(Ct, device=xa.device), torch.empty(Ct, device=xa.device)
    dfilm = torch.zeros((B if film_batch else 1, 2 * Ct), dtype=torch.float32, device=xa.device) if film is not None else None
    args = (_ptr(dact), resample, _ptr(xa), _ptr(xb), Ca, Cb, Hs, Ws, B, _ptr(coef), _ptr(stats),
            _ptr(gamma), _ptr(beta), _ptr(film), film_batch, film_stride, act, _ptr(dxa), _ptr(dxb),
            int(dx_init is not None), _ptr(add), add_mode, _ptr(ab), _ptr(dg), _ptr(dbt), _ptr(dfilm), 2 * Ct)
    if sync is not None:
        if sync.dtype != torch.int32 or sync.numel() < GN_SYNC_WORDS * B * min(32, Ct // 4) or not sync.is_cuda:
            raise ValueError("op_gn_bwd: sync must be a device int32 tensor of at least GN_SYNC_WORDS * B * groups zeros")
        check(lib.mcedm_op_gn_bwd_sync(*args, _ptr(sync, torch.int32), _stream()), "op_gn_bwd_sync")
    else:
        check(lib.mcedm_op_gn_bwd(*args, _stream()), "op_gn_bwd")
    return dxa, dxb, dg, dbt, dfilm


def op_attention_bwd(qkv, a, da, heads):
    lib = _bind_ops()
    B, C3, H, W = qkv.shape
    dqkv = torch.empty_like(qkv)
    lse = torch.empty(B * heads * H * W * 2, dtype=torch.float32, device=qkv.device)
    check(lib.mcedm_op_attention_bwd(_ptr(qkv), _ptr(a), _ptr(da), _ptr(dqkv), _ptr(lse), B, heads, H * W, _stream()),
          "op_attention_bwd")
    return dqkv
