"""Drop-in ``Model`` for the reference's DDPM U-Net (``models/ddim_blocks.py:222-470``).

Same constructor (``Model(hparams)``), the same ``state_dict`` keys / shapes (``temb.dense.0.weight``,
``down.0.block.0.norm1.weight``, ``mid.attn_1.q.weight``, ``up.2.upsample.conv.weight`` ...) so reference checkpoints load
unchanged, and the same ``forward(x, t, cond=None, x_self_cond=None, dx=None)`` signature; the compute runs in
libmcedm_hip.so (csrc/ddpm.hip).  The sub-modules only OWN parameters (plain torch layers, i.e. the reference's own
default initialisation).  What ``PlDdim.sample_edm`` evaluates is built: inference, one timestep for the whole batch,
``cond`` / ``x_self_cond`` / ``dx`` None.  Everything else (cond_enc, dx conditioning, a self-conditioning tensor, per-sample
timesteps, autograd) raises instead of silently computing something else.  There is no PyTorch fallback.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
from torch import nn

from . import lib as _lib
from .adm_blocks import EmaModel  # noqa: F401  (re-exported like the reference module does)


def _get(hp, name, default):
    try:
        return getattr(hp, name)
    except (AttributeError, KeyError):
        return default


def Normalize(in_channels):
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class ResnetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)


class Downsample(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.with_conv = True
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)


class Upsample(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.with_conv = True
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)


class Model(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        m = hparams.model
        unsupported = []
        if _get(m, "cond_channels", 0) > 0:
            unsupported.append("cond_channels > 0 (cond_enc / cat_cond)")
        if _get(m, "dx_cond", False):
            unsupported.append("dx_cond")
        if m.dropout:
            unsupported.append("dropout")
        if not m.resamp_with_conv:
            unsupported.append("resamp_with_conv=False")
        if m.type == "bayesian":
            unsupported.append("type=bayesian")
        if unsupported:
            raise NotImplementedError("outside the built path (SURVEY.md section 8 f1): " + ", ".join(unsupported))
        ch, mult = m.ch, tuple(m.ch_mult)
        self.ch, self.temb_ch = ch, 4 * ch
        self.num_resolutions, self.num_res_blocks, self.resolution = len(mult), m.num_res_blocks, m.resolution
        self.self_condition = bool(_get(m, "self_cond", False))
        self.cat_condition, self.dx_cond, self.cat_dx, self.cond_channels = False, False, False, 0
        self.state_channels = m.in_channels
        self.in_channels = m.in_channels * (2 if self.self_condition else 1)
        self.cond_enc = self.dx_enc = self.combine_enc = None
        self._arch = dict(in_channels=m.in_channels, out_channels=m.out_ch, ch=ch, ch_mult=mult, num_res_blocks=m.num_res_blocks,
                          attn_resolutions=tuple(m.attn_resolutions), resolution=m.resolution, self_cond=self.self_condition)
        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(ch, self.temb_ch), nn.Linear(self.temb_ch, self.temb_ch)])
        self.conv_in = nn.Conv2d(self.in_channels, ch, 3, 1, 1)
        curr_res, in_mult = m.resolution, (1,) + mult
        self.down = nn.ModuleList()
        block_in = ch
        for lv in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_mult[lv], ch * mult[lv]
            for _ in range(self.num_res_blocks):
                block.append(ResnetBlock(block_in, block_out, self.temb_ch))
                block_in = block_out
                if curr_res in m.attn_resolutions:
                    attn.append(AttnBlock(block_in))
            down = nn.Module()
            down.block, down.attn = block, attn
            if lv != self.num_resolutions - 1:
                down.downsample = Downsample(block_in)
                curr_res //= 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in, self.temb_ch)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in, self.temb_ch)
        self.up = nn.ModuleList()
        for lv in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out, skip_in = ch * mult[lv], ch * mult[lv]
            for j in range(self.num_res_blocks + 1):
                if j == self.num_res_blocks:
                    skip_in = ch * in_mult[lv]
                block.append(ResnetBlock(block_in + skip_in, block_out, self.temb_ch))
                block_in = block_out
                if curr_res in m.attn_resolutions:
                    attn.append(AttnBlock(block_in))
            up = nn.Module()
            up.block, up.attn = block, attn
            if lv != 0:
                up.upsample = Upsample(block_in)
                curr_res *= 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, m.out_ch, 3, 1, 1)
        # runtime state (not part of the state_dict)
        self._plan: Optional[_lib.DdpmPlan] = None
        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None
        self._ws = _lib.Workspace()

    # ---- HIP plumbing -------------------------------------------------------------------------------
    @property
    def plan(self) -> _lib.DdpmPlan:
        if self._plan is None:
            self._plan = _lib.DdpmPlan(**self._arch)
            if [n for n, _ in self.named_parameters()] != self._plan.param_names:
                raise RuntimeError("parameter table of the HIP plan and of the module disagree")
        return self._plan

    def timestep_freqs(self, device) -> torch.Tensor:
        """get_timestep_embedding's frequency vector, built with the reference's own expression (ddim_blocks.py:22-24)."""
        half = self.ch // 2
        emb = math.log(10000) / (half - 1)
        return torch.exp(torch.arange(half, dtype=torch.float32) * -emb).to(device)

    def invalidate_packed(self) -> None:
        self._packed_key = None

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)

    def packed_weights(self) -> torch.Tensor:
        params: Dict[str, torch.Tensor] = dict(self.named_parameters())
        key = tuple((p.data_ptr(), p._version) for p in params.values())
        if self._packed is None or key != self._packed_key:
            dev = next(iter(params.values())).device
            keep = self._packed if (self._packed is not None and self._packed.device == dev) else None
            self._packed = self.plan.pack(params, self.timestep_freqs(dev), keep)
            self._packed_key = key
        return self._packed

    def forward(self, x, t, cond=None, x_self_cond=None, dx=None):
        if cond is not None or dx is not None:
            raise NotImplementedError("cond / dx are outside the built path (the samplers pass None for both)")
        if x_self_cond is not None and not self.self_condition:
            raise RuntimeError("x_self_cond given to a network built with self_cond: False")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("the DDPM U-Net is built for inference (RePaint sampling); call it under torch.no_grad()")
        t = torch.as_tensor(t).reshape(-1)
        if t.numel() != 1 and not bool((t == t[0]).all()):
            raise NotImplementedError("one timestep for the whole batch (what the sampler evaluates)")
        sc = None if x_self_cond is None else x_self_cond.to(torch.float32).contiguous()
        return self.plan.forward(self.packed_weights(), x.to(torch.float32).contiguous(), float(t[0]), ws=self._ws, x_self_cond=sc)
