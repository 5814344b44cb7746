"""CPU oracle for the m-cedm EDM hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32/fp64) restatement of the reference's
algorithm for the one hot path this repository accelerates.  It is NOT part of
the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
The product path (``m-cedm_amd/``) never imports it and fails loudly when the
HIP library is missing.

Parity pinning: every function below is checked against the reference itself
(imported from /root/reference in the build container by
``oracle/make_golden.py``) and against the golden vectors that script commits
under ``tests/golden/`` (see ``tests/test_oracle_golden.py``).

Functions cite the reference lines they restate (paths relative to the
reference checkout).  Parameters are passed as a flat ``dict[str, Tensor]``
keyed exactly like ``DhariwalUNet.state_dict()`` so that reference checkpoints
and the HIP path share one naming scheme.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# architecture description (restates DhariwalUNet.__init__, adm_blocks.py:203-317)
# --------------------------------------------------------------------------- #
@dataclass
class UNetConfig:
    """Subset of ``hparams.model`` the hot path reads (adm_edm_mcedm_res32.yaml:4-28)."""
    in_channels: int = 2
    cond_channels: int = 2
    out_ch: int = 2
    ch: int = 64
    ch_mult: Tuple[int, ...] = (1, 1, 1)
    num_res_blocks: int = 1
    attn_resolutions: Tuple[int, ...] = (32,)
    resolution: int = 128
    channels_per_head: int = 64
    eps: float = 1e-5
    # dx_cond (adm_blocks.py:233-238, 266-280): channels of the dx input (= hparams.model.in_channels) and how it enters:
    # "cat" (cat_dx=True: concatenated to conv_in's input) or "enc" (dx_enc + combine_enc); 0 / "" = dx_cond False
    dx_channels: int = 0
    dx_mode: str = ""


@dataclass
class BlockSpec:
    key: str            # e.g. "enc.64x64_down"
    cin: int
    cout: int
    up: bool = False
    down: bool = False
    attn: bool = False
    heads: int = 0
    skip_kernel: int = -1   # -1 no skip module, 0 resample-only, 1 1x1 conv


@dataclass
class UNetSpec:
    cfg: UNetConfig
    conv_in_key: str
    enc: List[BlockSpec] = field(default_factory=list)
    dec: List[BlockSpec] = field(default_factory=list)
    # channel count pushed on the skip stack by conv_in and each encoder block
    skip_channels: List[int] = field(default_factory=list)


def build_spec(cfg: UNetConfig) -> UNetSpec:
    """Block list of the network, in execution order (adm_blocks.py:282-317)."""
    ch = cfg.ch
    in_total = cfg.in_channels + cfg.cond_channels      # cat_cond=True (adm_blocks.py:236-238)
    res0 = cfg.resolution
    spec = UNetSpec(cfg=cfg, conv_in_key=f"enc.{res0}x{res0}_conv")

    def mk(key, cin, cout, up=False, down=False, attn=False):
        heads = cout // cfg.channels_per_head if attn else 0        # adm_blocks.py:135
        skip_kernel = -1
        if cout != cin or up or down:                               # adm_blocks.py:148-151
            skip_kernel = 1 if cout != cin else 0
        return BlockSpec(key, cin, cout, up, down, bool(heads), heads, skip_kernel)

    cout = in_total
    for level, mult in enumerate(cfg.ch_mult):
        res = cfg.resolution >> level
        if level == 0:
            cout = ch * mult
            spec.skip_channels.append(cout)          # conv_in output is skips[0] (adm_blocks.py:392)
        else:
            spec.enc.append(mk(f"enc.{res}x{res}_down", cout, cout, down=True))
            spec.skip_channels.append(cout)
        for idx in range(cfg.num_res_blocks):
            cin = cout
            cout = ch * mult
            spec.enc.append(mk(f"enc.{res}x{res}_block{idx}", cin, cout, attn=(res in cfg.attn_resolutions)))
            spec.skip_channels.append(cout)
    skips = list(spec.skip_channels)
    for level, mult in reversed(list(enumerate(cfg.ch_mult))):
        res = cfg.resolution >> level
        if level == len(cfg.ch_mult) - 1:
            spec.dec.append(mk(f"dec.{res}x{res}_in0", cout, cout, attn=True))
            spec.dec.append(mk(f"dec.{res}x{res}_in1", cout, cout))
        else:
            spec.dec.append(mk(f"dec.{res}x{res}_up", cout, cout, up=True))
        for idx in range(cfg.num_res_blocks + 1):
            cin = cout + skips.pop()
            cout = ch * mult
            spec.dec.append(mk(f"dec.{res}x{res}_block{idx}", cin, cout, attn=(res in cfg.attn_resolutions)))
    return spec


def block_param_shapes(b: BlockSpec, emb: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """Parameters of one UNetBlock in registration order (adm_blocks.py:140-157)."""
    k = b.key
    o = [(f"{k}.norm0.weight", (b.cin,)), (f"{k}.norm0.bias", (b.cin,)),
         (f"{k}.conv0.weight", (b.cout, b.cin, 3, 3)), (f"{k}.conv0.bias", (b.cout,)),
         (f"{k}.affine.weight", (2 * b.cout, emb)), (f"{k}.affine.bias", (2 * b.cout,)),
         (f"{k}.norm1.weight", (b.cout,)), (f"{k}.norm1.bias", (b.cout,)),
         (f"{k}.conv1.weight", (b.cout, b.cout, 3, 3)), (f"{k}.conv1.bias", (b.cout,))]
    if b.skip_kernel == 1:
        o += [(f"{k}.skip.weight", (b.cout, b.cin, 1, 1)), (f"{k}.skip.bias", (b.cout,))]
    if b.attn:
        o += [(f"{k}.norm2.weight", (b.cout,)), (f"{k}.norm2.bias", (b.cout,)),
              (f"{k}.qkv.weight", (3 * b.cout, b.cout, 1, 1)), (f"{k}.qkv.bias", (3 * b.cout,)),
              (f"{k}.proj.weight", (b.cout, b.cout, 1, 1)), (f"{k}.proj.bias", (b.cout,))]
    return o


def param_shapes(cfg: UNetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every *parameter* in ``DhariwalUNet.state_dict()`` order
    (buffers ``*.resample_filter`` are not parameters and are omitted)."""
    spec = build_spec(cfg)
    ch = cfg.ch
    emb = ch
    out: List[Tuple[str, Tuple[int, ...]]] = []
    out += [("map_layer0.weight", (emb, ch)), ("map_layer0.bias", (emb,)),
            ("map_layer1.weight", (emb, emb)), ("map_layer1.bias", (emb,))]
    in_total = cfg.in_channels + cfg.cond_channels + (cfg.dx_channels if cfg.dx_mode == "cat" else 0)   # adm_blocks.py:238
    c0 = ch * cfg.ch_mult[0]
    if cfg.dx_mode == "enc":      # registered before self.enc (adm_blocks.py:266-280)
        out += [("dx_enc.0.weight", (c0, cfg.dx_channels, 3, 3)), ("dx_enc.0.bias", (c0,)),
                ("dx_enc.2.weight", (c0, c0, 3, 3)), ("dx_enc.2.bias", (c0,)),
                ("combine_enc.weight", (c0, 2 * c0, 3, 3)), ("combine_enc.bias", (c0,))]
    # ModuleDict order: conv_in first, then encoder blocks in creation order
    out += [(f"{spec.conv_in_key}.weight", (c0, in_total, 3, 3)), (f"{spec.conv_in_key}.bias", (c0,))]
    for b in spec.enc:
        out += block_param_shapes(b, emb)
    for b in spec.dec:
        out += block_param_shapes(b, emb)
    clast = spec.dec[-1].cout
    out += [("out_norm.weight", (clast,)), ("out_norm.bias", (clast,)),
            ("out_conv.weight", (cfg.out_ch, clast, 3, 3)), ("out_conv.bias", (cfg.out_ch,))]
    return out


def fill_param(name: str, shape: Sequence[int], u: np.ndarray) -> np.ndarray:
    """Map a U(-1,1) draw to a test parameter: weights /sqrt(fan_in), GroupNorm gains 1+0.2u, rest 0.1u."""
    if name.endswith(".weight") and len(shape) >= 2:
        return u / math.sqrt(int(np.prod(shape[1:])))
    if ".norm" in name or name.startswith("out_norm"):
        return (1.0 + 0.2 * u) if name.endswith(".weight") else 0.1 * u
    return 0.1 * u


def make_params(cfg: UNetConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Deterministic non-trivial parameter fill (SURVEY.md §8c fixture recipe).

    The reference zero-initialises conv1/proj/out_conv (adm_blocks.py:145,157,317)
    so default-init outputs are identically zero; tests instead draw every
    parameter from one NumPy stream in ``param_shapes`` order:
    weights ~ U(-1,1)/sqrt(fan_in), biases ~ 0.1*U(-1,1), GroupNorm gains ~ 1+0.2*U(-1,1).
    """
    rng = np.random.default_rng(seed)
    P: Dict[str, Tensor] = {}
    for name, shape in param_shapes(cfg):
        u = rng.random(size=shape, dtype=np.float64) * 2.0 - 1.0
        P[name] = torch.from_numpy(fill_param(name, shape, u).astype(np.float32)).to(dtype)
    return P


# --------------------------------------------------------------------------- #
# primitive ops
# --------------------------------------------------------------------------- #
def positional_embedding(x: Tensor, num_channels: int, max_positions: int = 10000) -> Tensor:
    """adm_blocks.py:192-199 (endpoint=False)."""
    half = num_channels // 2
    freqs = torch.arange(0, half).to(x.dtype) / half
    freqs = (1.0 / max_positions) ** freqs
    x = torch.outer(x, freqs)
    return torch.cat([x.cos(), x.sin()], dim=1)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """adm_blocks.py:28-32."""
    y = x @ w.t()
    return y + b if b is not None else y


def group_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """adm_blocks.py:86-97: groups = min(32, C // 4)."""
    groups = min(32, x.shape[1] // 4)
    return F.group_norm(x, groups, w, b, eps)


def resample_down(x: Tensor) -> Tensor:
    """adm_blocks.py:75-77 with resample_filter [1,1]: depthwise 2x2 box, stride 2 == 2x2 mean."""
    return F.avg_pool2d(x, 2)


def resample_up(x: Tensor) -> Tensor:
    """adm_blocks.py:72-74: transposed depthwise conv with f*4 == ones(2,2), stride 2 == nearest 2x."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def conv2d(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], up=False, down=False) -> Tensor:
    """adm_blocks.py:57-82, fused_resample=False branch; w None == kernel 0."""
    if up:
        x = resample_up(x)
    if down:
        x = resample_down(x)
    if w is not None:
        x = F.conv2d(x, w, padding=w.shape[-1] // 2)
    if b is not None:
        x = x + b.reshape(1, -1, 1, 1)
    return x


def attention(qkv: Tensor, heads: int) -> Tensor:
    """adm_blocks.py:103-109,175-178.  qkv [B, 3C, H, W] -> a [B, C, H, W]."""
    B, C3, H, W = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B * heads, C // heads, 3, H * W).unbind(2)
    w = torch.einsum("ncq,nck->nqk", q, k / math.sqrt(k.shape[1])).softmax(dim=2)
    a = torch.einsum("nqk,nck->ncq", w, v)
    return a.reshape(B, C, H, W)


def unet_block(P: Dict[str, Tensor], b: BlockSpec, x: Tensor, emb: Tensor, eps: float = 1e-5) -> Tensor:
    """adm_blocks.py:159-181 (adaptive_scale=True, dropout=0, skip_scale=1)."""
    k = b.key
    orig = x
    x = conv2d(F.silu(group_norm(x, P[f"{k}.norm0.weight"], P[f"{k}.norm0.bias"], eps)),
               P[f"{k}.conv0.weight"], P[f"{k}.conv0.bias"], up=b.up, down=b.down)
    params = linear(emb, P[f"{k}.affine.weight"], P[f"{k}.affine.bias"])[:, :, None, None]
    scale, shift = params.chunk(2, dim=1)
    x = F.silu(torch.addcmul(shift, group_norm(x, P[f"{k}.norm1.weight"], P[f"{k}.norm1.bias"], eps), scale + 1))
    x = conv2d(x, P[f"{k}.conv1.weight"], P[f"{k}.conv1.bias"])
    if b.skip_kernel >= 0:
        sk = conv2d(orig, P.get(f"{k}.skip.weight"), P.get(f"{k}.skip.bias"), up=b.up, down=b.down)
    else:
        sk = orig
    x = x + sk
    if b.attn:
        qkv = conv2d(group_norm(x, P[f"{k}.norm2.weight"], P[f"{k}.norm2.bias"], eps),
                     P[f"{k}.qkv.weight"], P[f"{k}.qkv.bias"])
        a = attention(qkv, b.heads)
        x = conv2d(a, P[f"{k}.proj.weight"], P[f"{k}.proj.bias"]) + x
    return x


def noise_embedding(P: Dict[str, Tensor], cfg: UNetConfig, noise_labels: Tensor) -> Tensor:
    """adm_blocks.py:367-379 with map_augment/map_label absent (augment_dim=label_dim=0)."""
    emb = positional_embedding(noise_labels, cfg.ch)
    emb = F.silu(linear(emb, P["map_layer0.weight"], P["map_layer0.bias"]))
    emb = linear(emb, P["map_layer1.weight"], P["map_layer1.bias"])
    return F.silu(emb)


def unet_forward(P: Dict[str, Tensor], cfg: UNetConfig, x: Tensor, noise_labels: Tensor,
                 cond: Optional[Tensor] = None, spec: Optional[UNetSpec] = None, dx: Optional[Tensor] = None) -> Tensor:
    """DhariwalUNet.forward, adm_blocks.py:364-404 (cat_cond=True; cond_enc/self_cond off; dx_cond per cfg.dx_mode)."""
    spec = spec or build_spec(cfg)
    emb = noise_embedding(P, cfg, noise_labels)
    x_in = x
    if cfg.cond_channels > 0:
        if cond is None:                                 # adm_blocks.py:328-331
            cond = torch.zeros((x.shape[0], cfg.cond_channels) + tuple(x.shape[2:]), dtype=x.dtype)
        x = torch.cat((cond, x), dim=1)                  # cond FIRST (adm_blocks.py:332)
    if cfg.dx_mode == "cat":                             # adm_blocks.py:335-339: dx LAST, zeros when None
        x = torch.cat((x, torch.zeros_like(x_in[:, :cfg.dx_channels]) if dx is None else dx), dim=1)
    x = conv2d(x, P[f"{spec.conv_in_key}.weight"], P[f"{spec.conv_in_key}.bias"])
    if cfg.dx_mode == "enc":                             # adm_blocks.py:352-362
        if dx is not None:
            d = conv2d(dx, P["dx_enc.0.weight"], P["dx_enc.0.bias"])
            d = conv2d(F.gelu(d), P["dx_enc.2.weight"], P["dx_enc.2.bias"])
        else:
            d = torch.zeros_like(x)
        x = conv2d(torch.cat([x, d], dim=1), P["combine_enc.weight"], P["combine_enc.bias"])
    skips = [x]
    for b in spec.enc:
        x = unet_block(P, b, x, emb, cfg.eps)
        skips.append(x)
    for b in spec.dec:
        if x.shape[1] != b.cin:
            x = torch.cat([x, skips.pop()], dim=1)
        x = unet_block(P, b, x, emb, cfg.eps)
    x = conv2d(F.silu(group_norm(x, P["out_norm.weight"], P["out_norm.bias"])),   # out_norm uses default eps
               P["out_conv.weight"], P["out_conv.bias"])
    return x


# --------------------------------------------------------------------------- #
# EDM preconditioning, loss, sampler (models/mcedm.py)
# --------------------------------------------------------------------------- #
SIGMA_DATA = 1.0      # mcedm.py:47
P_MEAN, P_STD = -1.2, 1.2   # mcedm.py:45-46
SIGMA_MIN, SIGMA_MAX = 0.002, 80   # mcedm.py:49-50


def precond_coeffs(sigma: Tensor, sigma_data: float = SIGMA_DATA):
    """mcedm.py:203-206 / :448-451 (fp32)."""
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = sigma * sigma_data / (sigma ** 2 + sigma_data ** 2).sqrt()
    c_in = 1 / (sigma_data ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    return c_skip, c_out, c_in, c_noise


def model_precond(P, cfg, x_noise: Tensor, sigma: Tensor, cond: Optional[Tensor] = None,
                  return_F: bool = False, dx: Optional[Tensor] = None):
    """PlMcedm.model_precond, mcedm.py:199-211 (== get_denoised, :443-461, with w=0)."""
    sigma = sigma.to(torch.float32).reshape(-1, 1, 1, 1)
    c_skip, c_out, c_in, c_noise = precond_coeffs(sigma)
    F_x = unet_forward(P, cfg, c_in * x_noise, c_noise.flatten(), cond, dx=dx)
    D_x = c_skip * x_noise + c_out * F_x
    return (D_x, F_x) if return_F else D_x


def get_denoised(P, cfg, xt: Tensor, t: Tensor, cond: Optional[Tensor] = None, w: Optional[float] = None,
                 dx: Optional[Tensor] = None):
    """PlMcedm.get_denoised, mcedm.py:443-461 including the classifier-free branch (models/ddim.py:1745-1763 with dx: the
    branch is taken when cond OR dx is given, and its unconditional evaluation drops both)."""
    xt = xt.to(torch.float32)
    sigma = t.to(torch.float32).reshape(-1, 1, 1, 1)
    c_skip, c_out, c_in, c_noise = precond_coeffs(sigma)
    if w is None or abs(w) < 0.001 or (cond is None and dx is None):
        F_x = unet_forward(P, cfg, c_in * xt, c_noise.flatten(), cond, dx=dx)
    else:
        F_x = (w + 1) * unet_forward(P, cfg, c_in * xt, c_noise.flatten(), cond, dx=dx) \
            - w * unet_forward(P, cfg, c_in * xt, c_noise.flatten(), None)
    D_x = c_skip * xt + c_out * F_x
    return D_x, F_x


def loss_weight(sigma: Tensor, sigma_data: float = SIGMA_DATA) -> Tensor:
    """mcedm.py:237-239."""
    return (sigma ** 2 + sigma_data ** 2) / (sigma * sigma_data) ** 2


def training_loss(P, cfg, x: Tensor, cond_in: Tensor, mask_c: Tensor, noise: Tensor, rnd_normal: Tensor) -> Tensor:
    """training_step after data_transform/get_cond_in, mcedm.py:266-278 + forward :213-235
    + NoiseEstimationLoss losses.py:48-59.  All tensors NCHW; rnd_normal [B,1,1,1]."""
    sigma = (rnd_normal * P_STD + P_MEAN).exp()
    weight = loss_weight(sigma)
    x_noise = x + mask_c * noise * sigma
    D_x = model_precond(P, cfg, x_noise, sigma.float(), cond_in)
    loss_matrix = weight * (D_x * mask_c - x * mask_c) ** 2
    return loss_matrix.sum(dim=(1, 2, 3)).mean()


def training_loss_cond(P, cfg, u: Tensor, cond_in: Tensor, noise: Tensor, rnd_normal: Tensor, dx_input=None) -> Tensor:
    """PlCondEdm.training_step (single-task conditional EDM), models/ddim.py:1700-1727 + forward :1661-1687
    with cond_p = 1, self_cond False: unmasked noising and loss.  dx_input (dx_cond, :1673-1681): callable
    (h, x_noise) -> dx, evaluated on the NOISED target and carrying no gradient; None = the 10 % branch without dx."""
    sigma = (rnd_normal * P_STD + P_MEAN).exp()
    weight = loss_weight(sigma)
    x_noise = u + noise * sigma
    dx = None if dx_input is None else dx_input(cond_in, x_noise).detach()
    D_x = model_precond(P, cfg, x_noise, sigma.float(), cond_in, dx=dx)
    return (weight * (D_x - u) ** 2).sum(dim=(1, 2, 3)).mean()


def cond_input(x: Tensor, mask: Tensor, cond_noise: Tensor) -> Tensor:
    """get_cond_in, mcedm.py:247 (add_cond_mask False, add_xt False); any layout."""
    return x * (1 - mask) + cond_noise * mask


def edm_t_steps(num_steps: int, sigma_min: float, sigma_max: float, rho: float) -> Tensor:
    """mcedm.py:579-588, float64, with t_N = 0 appended."""
    sigma_min = max(sigma_min, SIGMA_MIN)
    sigma_max = min(sigma_max, SIGMA_MAX)
    idx = torch.arange(num_steps, dtype=torch.float64)
    t = (sigma_max ** (1 / rho) + idx / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros_like(t[:1])])


@dataclass
class SamplerParams:
    """configs/diff_sampler/edm_sampler.yaml:1-20 (fields sample_edm reads)."""
    timesteps: int = 18
    sigma_min: float = 0.002
    sigma_max: float = 80.0
    rho: float = 7.0
    S_churn: float = 0.0
    S_min: float = 0.0
    S_max: float = float("inf")
    S_noise: float = 1.0
    w: float = 0.0


def guidance_dx_cond(system: str, h: Tensor, denoised: Tensor, norm_stats) -> Tensor:
    """get_dx_log_prob(h, denoised, guide_dx=True) of the single-task models (models/ddim.py:641-650 -> get_dx_pde
    :1424-1450 with calc_prob=True): gradient of the PDE residual of x_unnorm = (h, u = denoised) w.r.t. x_unnorm, then the
    mean over the two field gradients -> [B, 1, H, W] fp32.  norm_stats = (input mean, std, target mean, std)."""
    from . import pde_oracle as po
    st = [torch.as_tensor(s, dtype=torch.float32) for s in norm_stats]
    hh = h[:, :1].to(torch.float32).permute(0, 2, 3, 1)
    uu = denoised.to(torch.float32).permute(0, 2, 3, 1)
    x_un = torch.cat([hh * st[1] + st[0], uu * st[3] + st[2]], dim=-1)
    if system == "darcy":
        d = po.darcy_guidance(x_un, calc_prob=True)
    else:
        Tn, lo, hi = (0.128, -0.5, 0.5) if system == "swe_per" else (1.28, -2.5, 2.5)
        d = po.swe_fv_guidance(x_un, x_un, st[1], st[3], Tn, lo, hi, 2)
    return torch.mean(d.permute(0, 3, 1, 2), dim=1, keepdim=True)


def sample_edm_cond(P, cfg, h: Tensor, sp: SamplerParams, init_noise: Tensor,
                    step_noise: Optional[Sequence[Tensor]] = None, return_last: bool = True, guidance=None,
                    dx_input=None) -> Tensor:
    """PlCondEdm.sample_edm, models/ddim.py:1532-1601 (guide_dx False, no self-conditioning): the unmasked Heun
    sampler; ``h`` [B, cond_ch, H, W] is pure conditioning, ``init_noise`` [B, out_ch, H, W].
    ``guidance`` (guide_dx=True): callable (h, denoised[f64]) -> dx [B, 1, H, W] fp32, see guidance_dx_cond.
    ``dx_input`` (dx_cond models, ddim.py:1571, 1584): callable (h, x[f64]) -> the network's dx input, evaluated on the
    current NOISY state before each denoiser call (get_dx_input with dx_norm='prob' == guidance_dx_cond)."""
    N = sp.timesteps
    t_steps = edm_t_steps(N, sp.sigma_min, sp.sigma_max, sp.rho)
    x_next = init_noise.to(torch.float64) * t_steps[0]
    xs = [x_next]
    for i in range(N):
        t_cur, t_next = t_steps[i], t_steps[i + 1]
        gamma = min(sp.S_churn / N, math.sqrt(2) - 1) if sp.S_min <= float(t_cur) <= float(sp.S_max) else 0
        t_hat = t_cur + gamma * t_cur
        eps_i = step_noise[i] if step_noise is not None else torch.zeros_like(x_next)
        x_hat = x_next + (t_hat ** 2 - t_cur ** 2).sqrt() * sp.S_noise * eps_i
        dxi = None if dx_input is None else dx_input(h, x_hat)
        denoised = get_denoised(P, cfg, x_hat, t_hat, cond=h, w=sp.w, dx=dxi)[0].to(torch.float64)
        d_cur = (x_hat - denoised) / t_hat
        if guidance is not None:                 # guide_dx: - weight * dx / t_hat, weight = 5 (ddim.py:1577-1579)
            d_cur = d_cur - 5. * guidance(h, denoised) / t_hat
        x_next = x_hat + (t_next - t_hat) * d_cur
        if i < N - 1:
            dxi = None if dx_input is None else dx_input(h, x_next)
            denoised = get_denoised(P, cfg, x_next, t_next, cond=h, w=sp.w, dx=dxi)[0].to(torch.float64)
            d_prime = (x_next - denoised) / t_next
            if guidance is not None:             # the reference divides by t_hat here too (ddim.py:1590-1591)
                d_prime = d_prime - 5. * guidance(h, denoised) / t_hat
            x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
        xs = [x_next] if return_last else xs + [x_next]
    return torch.stack(xs, dim=0).permute(1, 0, 3, 4, 2).contiguous()


def sample_edm(P, cfg, cond: Tensor, hu_mask: Tensor, sp: SamplerParams, init_noise: Tensor,
               step_noise: Optional[Sequence[Tensor]] = None, n_state: int = 2,
               return_last: bool = True) -> Tensor:
    """PlMcedm.sample_edm, mcedm.py:570-638 with guide_dx False / dx_cond False.

    ``init_noise`` replaces ``randn_like(hu)`` (:576) and ``step_noise[i]`` the
    per-step ``randn_like(x_cur)`` (:608); when ``step_noise`` is None the churn
    term is taken as zero noise (only valid for S_churn == 0 where its factor is 0).
    Returns [B, T, H, W, C] float64 like the reference."""
    N = sp.timesteps
    t_steps = edm_t_steps(N, sp.sigma_min, sp.sigma_max, sp.rho)
    hu_known = cond[:, 0:n_state]
    x_next = init_noise.to(torch.float64) * t_steps[0]
    x_next = hu_known * (1 - hu_mask) + x_next * hu_mask
    xs = [x_next]
    for i in range(N):
        t_cur, t_next = t_steps[i], t_steps[i + 1]
        x_cur = x_next
        gamma = min(sp.S_churn / N, math.sqrt(2) - 1) if sp.S_min <= float(t_cur) <= float(sp.S_max) else 0
        t_hat = t_cur + gamma * t_cur
        eps_i = step_noise[i] if step_noise is not None else torch.zeros_like(x_cur)
        x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * sp.S_noise * eps_i * hu_mask
        denoised, _ = get_denoised(P, cfg, x_hat, t_hat, cond=cond, w=sp.w)
        denoised = denoised.to(torch.float64)
        d_cur = (x_hat - denoised) / t_hat
        x_next = x_hat + (t_next - t_hat) * d_cur * hu_mask
        if i < N - 1:
            denoised, _ = get_denoised(P, cfg, x_next, t_next, cond=cond, w=sp.w)
            denoised = denoised.to(torch.float64)
            d_prime = (x_next - denoised) / t_next
            x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime) * hu_mask
        if return_last:
            xs = [x_next]
        else:
            xs.append(x_next)
    xs = torch.stack(xs, dim=0)
    return xs.permute(1, 0, 3, 4, 2).contiguous()      # 't b c h w -> b t h w c'


# --------------------------------------------------------------------------- #
# optimizer + EMA (mcedm.py:139-168, ddim_blocks.py:44-54, trainer_ddim.yaml:8-9)
# --------------------------------------------------------------------------- #
def clip_scale(grads: Sequence[Tensor], max_norm: float = 1.0) -> Tuple[float, float]:
    """torch.nn.utils.clip_grad_norm_ semantics (Lightning gradient_clip_val=1.0, algorithm=norm):
    coef = min(1, max_norm / (total_norm + 1e-6))."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    return min(1.0, max_norm / (total + 1e-6)), total


def adam_ema_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, ema: Tensor, step: int,
                  lr=2e-4, b1=0.9, b2=0.999, eps=1e-8, clip=1.0, ema_beta=0.999):
    """One torch.optim.Adam step (wd=0, amsgrad False) on clipped grads followed by
    EmaModel.update (ddim_blocks.py:44-54).  ``step`` counts from 1.  Returns new (p, m, v, ema)."""
    g = g * clip
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    ema = ema * ema_beta + (1 - ema_beta) * p
    return p, m, v, ema


# --------------------------------------------------------------------------- #
# evaluation loops (host-side bookkeeping of models/mcedm.py:283-441)
# --------------------------------------------------------------------------- #
def masked_l1(pred: Tensor, target: Tensor, mask: Tensor, loss_dim=None) -> Tensor:
    """MaskedLoss('l1'), models/losses.py:62-78."""
    pred, target = pred * mask, target * mask
    if loss_dim is None:
        return (pred - target).abs().sum() / mask.sum()
    return (pred[..., loss_dim] - target[..., loss_dim]).abs().sum() / mask[..., loss_dim].sum()


def pde_metric(system: str, x_nhwc: Tensor, norm_stats) -> Tensor:
    """get_pde_loss with x_gt_unnorm None, clamp_loss False, reduce True (mcedm.py:468-498) for the loss objects of
    models/loss_helper.py:16-27 ('swe_per', 'swe', 'darcy'); norm_stats = (input mean, std, target mean, std)."""
    from . import pde_oracle as po
    st = [torch.as_tensor(s, dtype=torch.float32) for s in norm_stats]
    h = x_nhwc[..., 0:1].to(torch.float32) * st[1] + st[0]                 # inverse_data_transform, mcedm.py:186-197
    u = x_nhwc[..., 1:2].to(torch.float32) * st[3] + st[2]
    x_un = torch.cat([h, u], dim=-1)
    if system == "darcy":
        return po.darcy_residual(x_un, clamp_loss=False).sum()
    Tn, lo, hi = (0.128, -0.5, 0.5) if system == "swe_per" else (1.28, -2.5, 2.5)
    return po.swe_fv_residual(x_un, x_un, st[1], st[3], Tn, lo, hi, 2, clamp_loss=False).sum()


def eval_test_step(P, cfg, h: Tensor, u: Tensor, masks: Dict[str, Tensor], noises, norm_stats, sp: SamplerParams,
                   n_samples: int, system: str, down_factor: int = 1) -> Dict[str, Tensor]:
    """PlMcedm.test_step, mcedm.py:343-441 (return_last True, guide_dx False).  ``noises[name]`` = (cond noise NHWC,
    sampler initial noise [(n b), C, T, X]) replace the randn_like draws of :247 and :576.  Returns the step dict plus
    the logged scalars under 'log::<name>'."""
    st = norm_stats
    state_gt = torch.cat([(h - st[0]) / st[1], (u - st[2]) / st[3]], dim=-1)           # data_transform, b h w c
    nb = len(h)
    out: Dict[str, Tensor] = {}
    for name, mask in masks.items():
        lo = 0 if name.startswith("h") else 1
        loss_dim = torch.arange(lo, lo + 1).long()
        cond_noise, init = noises[name]
        cond_rep = cond_input(state_gt, mask, cond_noise).permute(0, 3, 1, 2).repeat(n_samples, 1, 1, 1)
        mask_rep = mask.permute(0, 3, 1, 2).repeat(n_samples, 1, 1, 1)
        xs = sample_edm(P, cfg, cond_rep, mask_rep, sp, init)                               # [(n b), 1, T, X, 2] fp64
        xs_mean = xs.reshape(n_samples, nb, *xs.shape[1:]).mean(dim=0)
        hu_last = xs_mean[:, -1]
        mask_loss = mask
        if down_factor > 1:
            each = 2 ** (down_factor - 1)
            sel = torch.zeros_like(mask)
            sel[:, ::each, ::each] = 1.0
            mask_loss = mask * sel
        out[f"loss_{name}"] = masked_l1(hu_last, state_gt, mask_loss, loss_dim)
        un = torch.cat([hu_last[..., 0:1] * st[1] + st[0], hu_last[..., 1:2] * st[3] + st[2]], dim=-1)
        out[f"loss_{name}_un"] = masked_l1(un, torch.cat([h, u], dim=-1), mask_loss, loss_dim)
        out[f"log::test_pde_loss_{name}"] = pde_metric(system, xs[:, -1], st) / n_samples / nb
        out["log::test_pde_loss_gt"] = pde_metric(system, state_gt, st) / nb
        if n_samples < 15:
            last = xs[:, -1]
            out[f"traj_{name}"] = last.reshape(n_samples, nb, *last.shape[1:]).permute(1, 2, 3, 0, 4).unsqueeze(1)
            out[f"gt_{name}"] = state_gt
    return out


def eval_validation_step(P, cfg, h, u, masks, noises, norm_stats, sp: SamplerParams, system: str) -> Dict[str, Tensor]:
    """PlMcedm.validation_step on an evaluated epoch, mcedm.py:283-341; noises[name] = (cond noise, initial noise [b,C,T,X])."""
    st = norm_stats
    state_gt = torch.cat([(h - st[0]) / st[1], (u - st[2]) / st[3]], dim=-1)
    out: Dict[str, Tensor] = {}
    for name, mask in masks.items():
        cond_noise, init = noises[name]
        cond_in = cond_input(state_gt, mask, cond_noise).permute(0, 3, 1, 2)
        xs = sample_edm(P, cfg, cond_in, mask.permute(0, 3, 1, 2), sp, init)
        hu_last = xs[:, -1]
        out[f"loss_{name}"] = masked_l1(hu_last, state_gt, mask)
        un = torch.cat([hu_last[..., 0:1] * st[1] + st[0], hu_last[..., 1:2] * st[3] + st[2]], dim=-1)
        out[f"loss_{name}_un"] = masked_l1(un, torch.cat([h, u], dim=-1), mask)
        out[f"log::val_pde_loss_{name}"] = pde_metric(system, hu_last, st) / len(h)
        out[f"traj_{name}"] = hu_last.unsqueeze(1)
        out[f"gt_{name}"] = state_gt
    return out


# --------------------------------------------------------------------------- #
# metrics shared by the evaluation loops of models/ddim.py (PlDdim :294-533, PlCondDdim :1154-1319)
# --------------------------------------------------------------------------- #
def l1(a: Tensor, b: Tensor) -> Tensor:
    """nn.L1Loss(): mean |a - b| (nan on an empty slice, as in the reference)."""
    return (a - b).abs().mean()


def scale_each_min_max(state: Tensor) -> Tensor:
    """models/ddim.py:689-698: 'b h w c' -> per (sample, channel) (x - min) / (max - min)."""
    lo = state.amin(dim=(1, 2), keepdim=True)
    hi = state.amax(dim=(1, 2), keepdim=True)
    return (state - lo) / (hi - lo)


def correlation(pred: Tensor, target: Tensor) -> Tensor:
    """CorrelationLoss(reduction='none'), models/losses.py:93-124 -> [c]."""
    x = pred.reshape(pred.shape[0], -1, pred.shape[-1])
    y = target.reshape(target.shape[0], -1, target.shape[-1])
    xb, yb = x - x.mean(dim=1, keepdim=True), y - y.mean(dim=1, keepdim=True)
    den = torch.sqrt((xb * xb).sum(dim=1) * (yb * yb).sum(dim=1))
    den = torch.where(den == 0, den + 1e-7, den)
    return ((yb * xb).sum(dim=1) / den).mean(dim=0)


def _unnorm(x: Tensor, mean, std) -> Tensor:
    return x * std + mean


def eval_cond_test_step(P, cfg, h: Tensor, u: Tensor, norm_stats, sp: SamplerParams, n_samples: int, system: str,
                        init: Tensor, guidance: bool = False) -> Dict[str, Tensor]:
    """PlCondEdm.test_step, models/ddim.py:1219-1319 (cond_channels == h_ch, return_last True, select_by_pde False,
    plot_scaled False).  ``init`` [(n b), T, X, 1] replaces the randn_like(u_rep) of :1237.  Logged scalars under 'log::'."""
    st = norm_stats
    hn, un = (h - st[0]) / st[1], (u - st[2]) / st[3]
    state_gt = torch.cat([hn, un], dim=-1)
    nb = len(h)
    cond = hn.repeat(n_samples, 1, 1, 1).permute(0, 3, 1, 2)
    g = (lambda hh, dd: guidance_dx_cond(system, hh, dd, st)) if guidance else None
    xs = sample_edm_cond(P, cfg, cond, sp, init.permute(0, 3, 1, 2), None, return_last=True, guidance=g)
    xs_mean = xs.reshape(n_samples, nb, *xs.shape[1:]).mean(dim=0)
    u_last = xs_mean[:, -1, :, :, :1]
    out = {"loss": l1(u_last, un), "loss_u_un": l1(_unnorm(u_last, st[2], st[3]), u)}
    gt_scaled, xs_scaled = scale_each_min_max(state_gt), scale_each_min_max(xs[:, -1])
    xs_scaled_mean = xs_scaled.reshape(n_samples, nb, *xs_scaled.shape[1:]).mean(dim=0)
    out["test_mae_u_scaled"] = l1(xs_scaled_mean, gt_scaled[..., 1:2])
    out["log::test_corr_u"] = correlation(xs_mean[:, -1], un).mean()
    joint = torch.cat([hn.repeat(n_samples, 1, 1, 1).double(), xs[:, -1]], dim=-1)
    out["log::test_pde_loss"] = pde_metric(system, joint, st) / n_samples / nb
    out["log::test_pde_loss_gt"] = pde_metric(system, state_gt, st) / nb
    last = xs[:, -1]
    out["traj"] = last.reshape(n_samples, nb, *last.shape[1:]).permute(1, 2, 3, 0, 4).unsqueeze(1)
    out["gt"] = un
    return out


def eval_cond_validation_step(P, cfg, h: Tensor, u: Tensor, norm_stats, sp: SamplerParams, system: str, init: Tensor) -> Dict[str, Tensor]:
    """PlCondEdm.validation_step on an evaluated epoch, models/ddim.py:1154-1217; ``init`` [b, T, X, 1] = randn_like(u)."""
    st = norm_stats
    hn, un = (h - st[0]) / st[1], (u - st[2]) / st[3]
    state_gt = torch.cat([hn, un], dim=-1)
    xs = sample_edm_cond(P, cfg, hn.permute(0, 3, 1, 2), sp, init.permute(0, 3, 1, 2), None, return_last=True)
    last = xs[:, -1]
    out = {"loss": l1(last, un), "loss_u_un": l1(_unnorm(last, st[2], st[3]), u)}
    out["val_loss_u_scaled"] = l1(scale_each_min_max(last), scale_each_min_max(state_gt)[..., 1:2])
    out["log::val_corr_u"] = correlation(last, un).mean()
    out["log::val_pde_loss"] = pde_metric(system, torch.cat([hn.double(), last], dim=-1), st) / len(h)
    out["traj"], out["gt"] = last.unsqueeze(1), un
    return out
