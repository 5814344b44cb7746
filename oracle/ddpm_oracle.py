"""CPU oracle for the RePaint-style EDM sampler on the DDPM U-Net.  TEST INFRASTRUCTURE ONLY.

Plain-PyTorch restatement of
  * the DDPM U-Net ``Model``            models/ddim_blocks.py:222-470 (ResnetBlock :107-164, AttnBlock :167-219,
                                          Downsample :85-104, Upsample :66-82, get_timestep_embedding :12-30)
  * ``PlDdim.get_denoised``             models/ddim.py:915-947   (VP preconditioning: c_skip 1, c_out -sigma)
  * ``PlDdim.round_sigma``              models/ddim.py:949-957,  ``compute_alpha`` :700-704, ``get_edm_steps`` :131-137
  * ``PlDdim.sample_edm``               models/ddim.py:959-1051  (inner ``n_repeat`` loop, known-region re-noising)
for the configuration of configs/model/ddim_res32.yaml (type simple, self_cond True, cond_channels 0, dx_cond False,
dropout 0, resamp_with_conv True).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
Pinned by tests/golden/ddpm.npz, which oracle/make_golden_ddpm.py writes by running the reference itself.
Parameters are a flat dict keyed exactly like ``Model.state_dict()``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class DdpmConfig:
    """Subset of hparams.model / hparams.diffusion the path reads (configs/model/ddim_res32.yaml:4-40)."""
    in_channels: int = 2
    out_ch: int = 2
    ch: int = 64
    ch_mult: Tuple[int, ...] = (1, 1, 1)
    num_res_blocks: int = 1
    attn_resolutions: Tuple[int, ...] = (32,)
    resolution: int = 128
    self_cond: bool = True
    num_timesteps: int = 1000
    beta_start: float = 1e-4
    beta_end: float = 0.02


# --------------------------------------------------------------------------- #
# parameter table (registration order of Model.__init__, ddim_blocks.py:252-362)
# --------------------------------------------------------------------------- #
def _res_shapes(k, cin, cout, temb):
    o = [(f"{k}.norm1.weight", (cin,)), (f"{k}.norm1.bias", (cin,)),
         (f"{k}.conv1.weight", (cout, cin, 3, 3)), (f"{k}.conv1.bias", (cout,)),
         (f"{k}.temb_proj.weight", (cout, temb)), (f"{k}.temb_proj.bias", (cout,)),
         (f"{k}.norm2.weight", (cout,)), (f"{k}.norm2.bias", (cout,)),
         (f"{k}.conv2.weight", (cout, cout, 3, 3)), (f"{k}.conv2.bias", (cout,))]
    if cin != cout:
        o += [(f"{k}.nin_shortcut.weight", (cout, cin, 1, 1)), (f"{k}.nin_shortcut.bias", (cout,))]
    return o


def _attn_shapes(k, c):
    o = [(f"{k}.norm.weight", (c,)), (f"{k}.norm.bias", (c,))]
    for n in ("q", "k", "v", "proj_out"):
        o += [(f"{k}.{n}.weight", (c, c, 1, 1)), (f"{k}.{n}.bias", (c,))]
    return o


def param_shapes(cfg: DdpmConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    ch, temb = cfg.ch, 4 * cfg.ch
    nres = len(cfg.ch_mult)
    in_total = cfg.in_channels * (2 if cfg.self_cond else 1)
    out = [("temb.dense.0.weight", (temb, ch)), ("temb.dense.0.bias", (temb,)),
           ("temb.dense.1.weight", (temb, temb)), ("temb.dense.1.bias", (temb,)),
           ("conv_in.weight", (ch, in_total, 3, 3)), ("conv_in.bias", (ch,))]
    in_mult = (1,) + tuple(cfg.ch_mult)
    res = cfg.resolution
    block_in = ch
    for lv in range(nres):
        block_in, block_out = ch * in_mult[lv], ch * cfg.ch_mult[lv]
        blocks, attns = [], []
        for j in range(cfg.num_res_blocks):
            blocks += _res_shapes(f"down.{lv}.block.{j}", block_in, block_out, temb)
            block_in = block_out
            if res in cfg.attn_resolutions:
                attns += _attn_shapes(f"down.{lv}.attn.{j}", block_in)
        out += blocks + attns
        if lv != nres - 1:
            out += [(f"down.{lv}.downsample.conv.weight", (block_in, block_in, 3, 3)), (f"down.{lv}.downsample.conv.bias", (block_in,))]
            res //= 2
    out += _res_shapes("mid.block_1", block_in, block_in, temb) + _attn_shapes("mid.attn_1", block_in) + \
        _res_shapes("mid.block_2", block_in, block_in, temb)
    ups = {}
    for lv in reversed(range(nres)):
        block_out = ch * cfg.ch_mult[lv]
        skip_in = ch * cfg.ch_mult[lv]
        blocks, attns = [], []
        for j in range(cfg.num_res_blocks + 1):
            if j == cfg.num_res_blocks:
                skip_in = ch * in_mult[lv]
            blocks += _res_shapes(f"up.{lv}.block.{j}", block_in + skip_in, block_out, temb)
            block_in = block_out
            if res in cfg.attn_resolutions:
                attns += _attn_shapes(f"up.{lv}.attn.{j}", block_in)
        lvl = blocks + attns
        if lv != 0:
            lvl += [(f"up.{lv}.upsample.conv.weight", (block_in, block_in, 3, 3)), (f"up.{lv}.upsample.conv.bias", (block_in,))]
            res *= 2
        ups[lv] = lvl
    for lv in range(nres):                         # `self.up.insert(0, up)`: state_dict lists up.0 first
        out += ups[lv]
    out += [("norm_out.weight", (block_in,)), ("norm_out.bias", (block_in,)),
            ("conv_out.weight", (cfg.out_ch, block_in, 3, 3)), ("conv_out.bias", (cfg.out_ch,))]
    return out


def fill_param(name: str, shape: Sequence[int], u: np.ndarray) -> np.ndarray:
    """U(-1,1) draw -> test parameter: conv / linear weights / sqrt(fan_in), GroupNorm gains 1 + 0.2u, the rest 0.1u."""
    if name.endswith(".weight") and len(shape) >= 2:
        return u / math.sqrt(int(np.prod(shape[1:])))
    if ".norm" in name or name.startswith("norm_out"):
        return (1.0 + 0.2 * u) if name.endswith(".weight") else 0.1 * u
    return 0.1 * u


def make_params(cfg: DdpmConfig, seed: int = 0) -> Dict[str, Tensor]:
    rng = np.random.default_rng(seed)
    P = {}
    for name, shape in param_shapes(cfg):
        u = rng.random(size=shape, dtype=np.float64) * 2.0 - 1.0
        P[name] = torch.from_numpy(fill_param(name, shape, u).astype(np.float32))
    return P


# --------------------------------------------------------------------------- #
# network
# --------------------------------------------------------------------------- #
def timestep_freqs(ch: int) -> Tensor:
    """ddim_blocks.py:22-24: exp(arange(half) * -(ln 10000 / (half - 1))) in fp32."""
    half = ch // 2
    emb = math.log(10000) / (half - 1)
    return torch.exp(torch.arange(half, dtype=torch.float32) * -emb)


def timestep_embedding(t: Tensor, ch: int) -> Tensor:
    """get_timestep_embedding, ddim_blocks.py:12-30: [sin | cos] (sin FIRST)."""
    e = t.float()[:, None] * timestep_freqs(ch)[None, :]
    return torch.cat([torch.sin(e), torch.cos(e)], dim=1)


def _swish(x):
    """nonlinearity, ddim_blocks.py:33-35 (x * sigmoid(x); F.silu differs in the last bit)."""
    return x * torch.sigmoid(x)


def _norm(x, P, k):
    return F.group_norm(x, 32, P[f"{k}.weight"], P[f"{k}.bias"], eps=1e-6)          # Normalize, ddim_blocks.py:62-63


def resnet_block(P, k, x, temb):
    """ddim_blocks.py:144-164 (dropout 0)."""
    h = F.conv2d(_swish(_norm(x, P, f"{k}.norm1")), P[f"{k}.conv1.weight"], P[f"{k}.conv1.bias"], padding=1)
    h = h + F.linear(_swish(temb), P[f"{k}.temb_proj.weight"], P[f"{k}.temb_proj.bias"])[:, :, None, None]
    h = F.conv2d(_swish(_norm(h, P, f"{k}.norm2")), P[f"{k}.conv2.weight"], P[f"{k}.conv2.bias"], padding=1)
    if f"{k}.nin_shortcut.weight" in P:
        x = F.conv2d(x, P[f"{k}.nin_shortcut.weight"], P[f"{k}.nin_shortcut.bias"])
    return x + h


def attn_block(P, k, x):
    """ddim_blocks.py:194-219: single head over all C channels, scale C^-0.5."""
    h_ = _norm(x, P, f"{k}.norm")
    q = F.conv2d(h_, P[f"{k}.q.weight"], P[f"{k}.q.bias"])
    kk = F.conv2d(h_, P[f"{k}.k.weight"], P[f"{k}.k.bias"])
    v = F.conv2d(h_, P[f"{k}.v.weight"], P[f"{k}.v.bias"])
    b, c, hh, ww = q.shape
    w_ = torch.bmm(q.reshape(b, c, hh * ww).permute(0, 2, 1), kk.reshape(b, c, hh * ww)) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    h_ = torch.bmm(v.reshape(b, c, hh * ww), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + F.conv2d(h_, P[f"{k}.proj_out.weight"], P[f"{k}.proj_out.bias"])


def downsample(P, k, x):
    """ddim_blocks.py:97-101: pad (0,1,0,1) then 3x3 stride 2."""
    return F.conv2d(F.pad(x, (0, 1, 0, 1)), P[f"{k}.conv.weight"], P[f"{k}.conv.bias"], stride=2)


def upsample(P, k, x):
    """ddim_blocks.py:77-82: nearest 2x then 3x3."""
    return F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), P[f"{k}.conv.weight"], P[f"{k}.conv.bias"], padding=1)


def model_forward(P, cfg: DdpmConfig, x: Tensor, t: Tensor, x_self_cond: Optional[Tensor] = None) -> Tensor:
    """Model.forward, ddim_blocks.py:410-470 with cond None, dx None."""
    assert x.shape[2] == x.shape[3] == cfg.resolution
    temb = timestep_embedding(t, cfg.ch)
    temb = F.linear(_swish(F.linear(temb, P["temb.dense.0.weight"], P["temb.dense.0.bias"])),
                    P["temb.dense.1.weight"], P["temb.dense.1.bias"])
    if cfg.self_cond:
        x = torch.cat((torch.zeros_like(x) if x_self_cond is None else x_self_cond, x), dim=1)
    nres = len(cfg.ch_mult)
    hs = [F.conv2d(x, P["conv_in.weight"], P["conv_in.bias"], padding=1)]
    for lv in range(nres):
        for j in range(cfg.num_res_blocks):
            h = resnet_block(P, f"down.{lv}.block.{j}", hs[-1], temb)
            if f"down.{lv}.attn.{j}.norm.weight" in P:
                h = attn_block(P, f"down.{lv}.attn.{j}", h)
            hs.append(h)
        if lv != nres - 1:
            hs.append(downsample(P, f"down.{lv}.downsample", hs[-1]))
    h = hs[-1]
    h = resnet_block(P, "mid.block_1", h, temb)
    h = attn_block(P, "mid.attn_1", h)
    h = resnet_block(P, "mid.block_2", h, temb)
    for lv in reversed(range(nres)):
        for j in range(cfg.num_res_blocks + 1):
            h = resnet_block(P, f"up.{lv}.block.{j}", torch.cat([h, hs.pop()], dim=1), temb)
            if f"up.{lv}.attn.{j}.norm.weight" in P:
                h = attn_block(P, f"up.{lv}.attn.{j}", h)
        if lv != 0:
            h = upsample(P, f"up.{lv}.upsample", h)
    return F.conv2d(_swish(_norm(h, P, "norm_out")), P["conv_out.weight"], P["conv_out.bias"], padding=1)


# --------------------------------------------------------------------------- #
# diffusion schedule, VP preconditioning, RePaint-style Heun sampler (models/ddim.py)
# --------------------------------------------------------------------------- #
def betas_of(cfg: DdpmConfig) -> Tensor:
    """get_beta_schedule('linear'), ddim_blocks.py:487-490: float64 linspace -> float32."""
    return torch.from_numpy(np.linspace(cfg.beta_start, cfg.beta_end, cfg.num_timesteps, dtype=np.float64)).float()


def edm_steps_of(betas: Tensor) -> Tensor:
    """get_edm_steps, ddim.py:131-137: sigma_t = sqrt((1 - abar_t) / abar_t), largest first."""
    ab = (1.0 - betas).cumprod(dim=0)
    return ((1 - ab) / ab).sqrt().flip(dims=(0,))


def alphas_ext_of(betas: Tensor) -> Tensor:
    """the table compute_alpha indexes with t + 1 (ddim.py:700-704)."""
    return (1 - torch.cat([torch.zeros(1), betas], dim=0)).cumprod(dim=0)


def round_sigma(steps: Tensor, sigma: Tensor, return_index: bool = False) -> Tensor:
    """ddim.py:949-957."""
    s32 = sigma.to(torch.float32)
    index = torch.cdist(s32.reshape(1, -1, 1), steps.reshape(1, -1, 1)).argmin(2)
    result = index if return_index else steps[index.flatten()]
    return result.type_as(sigma).reshape(sigma.shape)


def get_denoised(P, cfg: DdpmConfig, steps: Tensor, xt: Tensor, t: Tensor):
    """PlDdim.get_denoised, ddim.py:915-947 with cond None, x_self_cond None, dx None."""
    xt = xt.to(torch.float32)
    sigma = t.to(torch.float32).reshape(-1, 1, 1, 1)
    c_out = -sigma
    c_in = 1 / (sigma ** 2 + 1).sqrt()
    c_noise = cfg.num_timesteps - 1 - round_sigma(steps, sigma, return_index=True).to(torch.float32)
    F_x = model_forward(P, cfg, c_in * xt, c_noise.flatten())
    return 1 * xt + c_out * F_x, F_x


@dataclass
class RepaintParams:
    """configs/diff_sampler/edm_sampler_inv.yaml (fields PlDdim.sample_edm reads)."""
    timesteps: int = 18
    sigma_min: float = 0.002
    sigma_max: float = 80.0
    rho: float = 7.0
    S_churn: float = 0.0
    S_min: float = 0.0
    S_max: float = float("inf")
    S_noise: float = 1.0
    n_repeat: int = 2
    n_time_h: int = 0
    n_time_u: int = 64
    w: float = 0.0


def sample_edm_repaint(P, cfg: DdpmConfig, hu: Tensor, sp: RepaintParams, init_noise: Tensor,
                       step_noise: Sequence[Tensor], repeat_noise: Sequence[Sequence[Tensor]], h_ch: int = 1,
                       u_ch: int = 1, return_last: bool = True) -> Tensor:
    """PlDdim.sample_edm, ddim.py:959-1051, guide_dx False.  ``hu`` [B, C, H, W] is the normalised joint state (the
    reference builds it from h, u in 'b h w c'); hu_mask = 1 marks KNOWN entries (rows < n_time_* of each field).
    ``init_noise`` replaces randn_like(hu) (:969), ``step_noise[i]`` the per-step draw (:1004) and
    ``repeat_noise[i][k]`` the re-noising draw between inner repeats (:1037).  Returns [B, T, H, W, C] float64."""
    betas = betas_of(cfg)
    steps = edm_steps_of(betas)
    aext = alphas_ext_of(betas)
    N = sp.timesteps
    hu_noise = init_noise
    mask = torch.ones_like(hu)
    mask[:, 0:h_ch, sp.n_time_h:, :] = 0.0
    mask[:, h_ch:h_ch + u_ch, sp.n_time_u:, :] = 0.0
    sigma_min = max(sp.sigma_min, float(steps[cfg.num_timesteps - 1]))
    sigma_max = min(sp.sigma_max, float(steps[0]))
    idx = torch.arange(N, dtype=torch.float64)
    t_steps = (sigma_max ** (1 / sp.rho) + idx / (N - 1) * (sigma_min ** (1 / sp.rho) - sigma_max ** (1 / sp.rho))) ** sp.rho
    t_steps = torch.cat([round_sigma(steps, t_steps), torch.zeros_like(t_steps[:1])])

    def alpha(t):
        return aext.index_select(0, t.long().reshape(1) + 1).view(-1, 1, 1, 1)

    aT = alpha(t_steps[0])
    x = (hu * aT.sqrt() + hu_noise * (1.0 - aT).sqrt()) * mask + hu_noise * (1.0 - mask)
    x_next = x.to(torch.float64) * t_steps[0]
    xs = [x_next]
    for i in range(N):
        t_cur, t_next = t_steps[i], t_steps[i + 1]
        x_cur = x_next
        gamma = min(sp.S_churn / N, math.sqrt(2) - 1) if sp.S_min <= float(t_cur) <= float(sp.S_max) else 0
        t_hat = round_sigma(steps, t_cur + gamma * t_cur)
        x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * sp.S_noise * step_noise[i]
        for k in range(sp.n_repeat):
            denoised = get_denoised(P, cfg, steps, x_hat, t_hat)[0].to(torch.float64)
            d_cur = (x_hat - denoised) / t_hat
            x_next = x_hat + (t_next - t_hat) * d_cur
            if i < N - 1:
                denoised = get_denoised(P, cfg, steps, x_next, t_next)[0].to(torch.float64)
                d_prime = (x_next - denoised) / t_next
                x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
            at = alpha(t_next)
            known = at.sqrt() * hu + (1 - at).sqrt() * hu_noise
            x_next = known * mask + x_next * (1.0 - mask)
            if k < sp.n_repeat - 1:
                t_hat = round_sigma(steps, t_next + (math.sqrt(2) - 1) * t_next)
                x_hat = x_next + (t_hat ** 2 - t_next ** 2).sqrt() * sp.S_noise * repeat_noise[i][k]
        if i == N - 1:
            x_next = hu * mask + x_next * (1.0 - mask)
        xs = [x_next] if return_last else xs + [x_next]
    return torch.stack(xs, dim=0).permute(1, 0, 3, 4, 2).contiguous()


@dataclass
class DdimParams:
    """configs/diff_sampler/ddim_sampler*.yaml (fields PlDdim.sample_with_repeat reads)."""
    timesteps: int = 50
    skip_type: str = "uniform"
    eta: float = 0.0
    n_repeat: int = 5
    n_time_h: int = 128
    n_time_u: int = 0


def ddim_sequence(num_timesteps: int, sp: DdimParams) -> List[int]:
    """models/ddim.py:823-830."""
    if sp.skip_type == "uniform":
        return list(range(0, num_timesteps, num_timesteps // sp.timesteps))
    if sp.skip_type == "quad":
        return [int(v) for v in (np.linspace(0, np.sqrt(num_timesteps * 0.8), sp.timesteps) ** 2)]
    raise NotImplementedError(sp.skip_type)


def sample_with_repeat(P, cfg: DdpmConfig, hu: Tensor, sp: DdimParams, init_noise: Tensor,
                       eta_noise: Optional[Sequence[Tensor]] = None, h_ch: int = 1, u_ch: int = 1,
                       return_last: bool = True) -> Tuple[Tensor, Tensor]:
    """PlDdim.sample_with_repeat, models/ddim.py:808-913 (guide_dx False, dx_cond False; w irrelevant without dx).  ``hu``
    [B, C, H, W] normalised joint state, ``init_noise`` replaces randn_like(hu) (:832), ``eta_noise[step]`` the rand_like of
    :893.  The previous x0 prediction is the network's x_self_cond when cfg.self_cond.  Returns (xs, x0_preds) 'b t h w c'."""
    betas = betas_of(cfg)
    aext = alphas_ext_of(betas)
    a = (1 - betas).cumprod(dim=0)
    seq = ddim_sequence(cfg.num_timesteps, sp)
    mask = torch.ones_like(hu)
    mask[:, 0:h_ch, sp.n_time_h:, :] = 0.0
    mask[:, h_ch:h_ch + u_ch, sp.n_time_u:, :] = 0.0
    hu_noise = init_noise
    x = (hu * a[-1].sqrt() + hu_noise * (1.0 - a[-1]).sqrt()) * mask + hu_noise * (1.0 - mask)
    n = hu.shape[0]
    seq_next = [-1] + list(seq[:-1])
    xs, x0_preds, x0_t = [x], [], None
    for step, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
        t = torch.ones(n) * i
        at = aext.index_select(0, t.long() + 1).view(-1, 1, 1, 1)
        at_next = aext.index_select(0, (torch.ones(n) * j).long() + 1).view(-1, 1, 1, 1)
        xt = xs[-1]
        for k in range(sp.n_repeat):
            et = model_forward(P, cfg, xt, t, x_self_cond=x0_t if cfg.self_cond else None)
            x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
            x0_t = hu * mask + x0_t * (1.0 - mask)
            if k < sp.n_repeat - 1:
                xt = at.sqrt() * x0_t + (1 - at).sqrt() * et
        if abs(sp.eta) > 1e-10:
            c1 = sp.eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
            c2 = ((1 - at_next) - c1 ** 2).sqrt()
            xt_next = at_next.sqrt() * x0_t + c1 * eta_noise[step] + c2 * et
        else:
            c2 = (1 - at_next).sqrt()
            xt_next = at_next.sqrt() * x0_t + c2 * et
        xt_next = (at_next.sqrt() * hu + c2 * hu_noise) * mask + xt_next * (1.0 - mask)
        if return_last:
            x0_preds, xs = [x0_t], [xt_next]
        else:
            x0_preds.append(x0_t)
            xs.append(xt_next)
    return (torch.stack(xs, dim=0).permute(1, 0, 3, 4, 2).contiguous(),
            torch.stack(x0_preds, dim=0).permute(1, 0, 3, 4, 2).contiguous())


# --------------------------------------------------------------------------- #
# evaluation loops of PlDdim (models/ddim.py:294-533): what `trainer.test` runs for BASELINE config 5
# --------------------------------------------------------------------------- #
def eval_test_step(P, cfg: DdpmConfig, h: Tensor, u: Tensor, norm_stats, sp, n_samples: int, system: str,
                   init: Tensor, step_noise=None, repeat_noise=None) -> Dict[str, Tensor]:
    """PlDdim.test_step, models/ddim.py:372-533 with type 'edm', return_last True, select_by_pde False, plot_scaled False.
    h, u: un-normalised 'b t x 1'; norm_stats = (input mean, std, target mean, std); the noises replace the draws of
    sample_edm (:969, :1004, :1037) on the (n b) batch.  Logged scalars are returned under 'log::<name>'."""
    from . import mcedm_oracle as mo
    st = norm_stats
    hn, un = (h - st[0]) / st[1], (u - st[2]) / st[3]
    state_gt = torch.cat([hn, un], dim=-1)
    nb, n_all = len(h), h.shape[1]
    hu = state_gt.repeat(n_samples, 1, 1, 1).permute(0, 3, 1, 2)
    if isinstance(sp, DdimParams):       # sparams.type != 'edm': the DDIM sampler with RePaint loops (ddim.py:393-394), fp32 states
        xs = sample_with_repeat(P, cfg, hu, sp, init, step_noise, return_last=True)[0]
    else:
        xs = sample_edm_repaint(P, cfg, hu, sp, init, step_noise, repeat_noise, return_last=True)
    xs_mean = xs.reshape(n_samples, nb, *xs.shape[1:]).mean(dim=0)
    h_last, u_last = xs_mean[:, -1, :, :, 0:1], xs_mean[:, -1, :, :, 1:2]
    out = {"loss_h": mo.l1(h_last, hn), "loss": mo.l1(u_last, un)}
    h_un, u_un = h_last * st[1] + st[0], u_last * st[3] + st[2]
    out["loss_h_un"], out["loss_u_un"] = mo.l1(h_un, h), mo.l1(u_un, u)
    unknown = torch.ones(nb, n_all, h.shape[2], 2, dtype=h_un.dtype)
    if sp.n_time_h > 0:
        unknown[:, :sp.n_time_h, :, 0] = 0.0
    if sp.n_time_u > 0:
        unknown[:, :sp.n_time_u, :, 1] = 0.0
    out["log::test_mae_hu_un"] = mo.masked_l1(torch.cat([h_un, u_un], dim=-1), torch.cat([h, u], dim=-1), unknown)
    gt_scaled, xs_scaled = mo.scale_each_min_max(state_gt), mo.scale_each_min_max(xs[:, -1])
    sc_mean = xs_scaled.reshape(n_samples, nb, *xs_scaled.shape[1:]).mean(dim=0)
    out["log::test_mae_h_scaled"] = mo.l1(sc_mean[..., 0:1], gt_scaled[..., 0:1])
    out["test_mae_u_scaled"] = mo.l1(sc_mean[..., 1:2], gt_scaled[..., 1:2])
    corr = mo.correlation(xs_mean[:, -1], state_gt)
    out["log::test_corr_h"], out["log::test_corr_u"] = corr[0:1].mean(), corr[1:2].mean()
    for tag, c, last, ref, k, on in (("h", 0, h_last, hn, sp.n_time_h, sp.n_time_h < n_all),
                                     ("u", 1, u_last, un, sp.n_time_u, n_all > sp.n_time_u > 0)):
        if on:
            out[f"log::test_{tag}_known"] = mo.l1(last[:, :k], ref[:, :k])
            out[f"log::test_{tag}_kn_scaled"] = mo.l1(sc_mean[:, :k, :, c:c + 1], gt_scaled[:, :k, :, c:c + 1])
            out[f"log::test_{tag}_unkn_scaled"] = mo.l1(sc_mean[:, k:, :, c:c + 1], gt_scaled[:, k:, :, c:c + 1])
    out["log::test_pde_loss"] = mo.pde_metric(system, xs[:, -1], st) / n_samples / nb
    out["log::test_pde_loss_gt"] = mo.pde_metric(system, state_gt, st) / nb
    last = xs[:, -1]
    out["traj"] = last.reshape(n_samples, nb, *last.shape[1:]).permute(1, 2, 3, 0, 4).unsqueeze(1)
    out["gt"] = state_gt
    return out


def eval_validation_step(P, cfg: DdpmConfig, h: Tensor, u: Tensor, norm_stats, sp: RepaintParams, system: str, u_noise: Tensor,
                         init: Tensor, step_noise, repeat_noise) -> Dict[str, Tensor]:
    """PlDdim.validation_step on an evaluated epoch, models/ddim.py:294-370: the u field handed to sample_edm is NOISE
    (``u_noise`` 'b t x 1', :306-309), so rows < n_time_u of it are treated as known."""
    from . import mcedm_oracle as mo
    st = norm_stats
    hn, un = (h - st[0]) / st[1], (u - st[2]) / st[3]
    state_gt = torch.cat([hn, un], dim=-1)
    hu = torch.cat([hn, u_noise], dim=-1).permute(0, 3, 1, 2)
    xs = sample_edm_repaint(P, cfg, hu, sp, init, step_noise, repeat_noise, return_last=True)
    last = xs[:, -1]
    h_last, u_last = last[..., 0:1], last[..., 1:2]
    out = {"loss_h": mo.l1(h_last, hn), "loss": mo.l1(u_last, un),
           "loss_h_un": mo.l1(h_last * st[1] + st[0], h), "loss_u_un": mo.l1(u_last * st[3] + st[2], u)}
    gt_scaled, xs_scaled = mo.scale_each_min_max(state_gt), mo.scale_each_min_max(last)
    out["val_loss_h_scaled"] = mo.l1(xs_scaled[..., 0:1], gt_scaled[..., 0:1])
    out["val_loss_u_scaled"] = mo.l1(xs_scaled[..., 1:2], gt_scaled[..., 1:2])
    corr = mo.correlation(last, state_gt)
    out["log::val_corr_h"], out["log::val_corr_u"] = corr[0:1].mean(), corr[1:2].mean()
    out["log::val_pde_loss"] = mo.pde_metric(system, last, st) / len(h)
    out["traj"], out["gt"] = last.unsqueeze(1), state_gt
    return out
