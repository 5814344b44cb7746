"""Drop-in ``DhariwalUNet`` for the reference's ``models/adm_blocks.py`` -- same constructor
(``DhariwalUNet(hparams)``), same ``state_dict`` keys / shapes (so reference checkpoints load
unchanged, incl. the ``resample_filter`` buffers) and the same ``forward`` signature
(models/adm_blocks.py:364), but the compute runs in libmcedm_hip.so (include/mcedm_hip.h).

The sub-modules below only OWN parameters (initialised with the reference's distributions,
models/adm_blocks.py:10-15,221-222); the network is executed as one fused schedule by the C library,
not module by module.  ``dx_cond`` (the network conditioned on the PDE-residual gradient, models/adm_blocks.py:233-280,
334-362) is built in both of the reference's forms: ``cat_dx=True`` (dx concatenated to conv_in's input) and
``cat_dx=False`` (``dx_enc`` = Conv3x3 -> GELU -> Conv3x3 and ``combine_enc``).  Configurations outside the hot path
(cond_enc / self-conditioning / class or augment labels / dropout) raise NotImplementedError instead of silently computing
something else.  There is no PyTorch fallback: without the HIP library or on a CPU tensor, forward raises.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
from torch import nn

from . import lib as _lib


def _init(shape, fan_in, gain):
    """kaiming_uniform * gain: U(-1,1) * sqrt(3/fan_in) * gain (reference weight_init, adm_blocks.py:13)."""
    if gain == 0:
        return torch.zeros(*shape)
    return (torch.rand(*shape) * 2 - 1) * (math.sqrt(3.0 / fan_in) * gain)


class Linear(nn.Module):
    def __init__(self, in_features, out_features, gain_w=math.sqrt(1 / 3), gain_b=math.sqrt(1 / 3)):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(_init([out_features, in_features], in_features, gain_w))
        self.bias = nn.Parameter(_init([out_features], in_features, gain_b))


class Conv2d(nn.Module):
    """kernel in {0, 1, 3}; kernel 0 owns no weights (resample-only skip, adm_blocks.py:49-52)."""

    def __init__(self, in_channels, out_channels, kernel, up=False, down=False, gain_w=math.sqrt(1 / 3),
                 gain_b=math.sqrt(1 / 3)):
        super().__init__()
        assert not (up and down)
        self.in_channels, self.out_channels, self.up, self.down = in_channels, out_channels, up, down
        fan_in = in_channels * kernel * kernel
        self.weight = nn.Parameter(_init([out_channels, in_channels, kernel, kernel], fan_in, gain_w)) if kernel else None
        self.bias = nn.Parameter(_init([out_channels], fan_in, gain_b)) if kernel else None
        # [1,1] box filter: outer([1,1],[1,1]) / 4 (adm_blocks.py:53-55); a buffer only so checkpoints round-trip
        self.register_buffer("resample_filter", torch.full((1, 1, 2, 2), 0.25) if (up or down) else None)


class GroupNorm(nn.Module):
    def __init__(self, num_channels, num_groups=32, min_channels_per_group=4, eps=1e-5):
        super().__init__()
        self.num_groups = min(num_groups, num_channels // min_channels_per_group)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))


class UNetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, emb_channels, up=False, down=False, attention=False,
                 channels_per_head=64):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_heads = out_channels // channels_per_head if attention else 0
        self.norm0 = GroupNorm(in_channels)
        self.conv0 = Conv2d(in_channels, out_channels, 3, up=up, down=down)
        self.affine = Linear(emb_channels, out_channels * 2)
        self.norm1 = GroupNorm(out_channels)
        self.conv1 = Conv2d(out_channels, out_channels, 3, gain_w=0, gain_b=0)          # init_zero
        self.skip = None
        if out_channels != in_channels or up or down:
            self.skip = Conv2d(in_channels, out_channels, 1 if out_channels != in_channels else 0, up=up, down=down)
        if self.num_heads:
            self.norm2 = GroupNorm(out_channels)
            self.qkv = Conv2d(out_channels, out_channels * 3, 1)
            self.proj = Conv2d(out_channels, out_channels, 1, gain_w=0, gain_b=0)       # init_zero


class PositionalEmbedding(nn.Module):
    def __init__(self, num_channels, max_positions=10000, endpoint=False):
        super().__init__()
        self.num_channels, self.max_positions, self.endpoint = num_channels, max_positions, endpoint


def _get(hp, name, default):
    return getattr(hp, name) if hasattr(hp, name) else default


class DhariwalUNet(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        m = hparams.model
        unsupported = []
        if _get(m, "self_cond", False):
            unsupported.append("self_cond")
        if m.augment_dim or m.label_dim:
            unsupported.append("augment_dim/label_dim")
        if m.dropout:
            unsupported.append("dropout")
        cond_channels = _get(m, "cond_channels", 0)
        if cond_channels > 0 and not _get(m, "cat_cond", False):
            unsupported.append("cond_enc (cat_cond=False)")
        ch, mult = m.ch, tuple(m.ch_mult)
        # attention widths: the bottleneck 'in0' block always, plus every level named in attn_resolutions; the
        # kernels are built for head_dim 64, the reference's head_dim is C / (C // 64) (adm_blocks.py:135,175)
        attn_widths = {ch * mult[-1]} | {ch * mu for lv, mu in enumerate(mult) if (m.resolution >> lv) in m.attn_resolutions}
        bad = sorted(c for c in attn_widths if c >= 64 and c % 64)
        if bad:
            unsupported.append(f"attention over {bad} channels (head_dim != 64)")
        if unsupported:
            raise NotImplementedError("outside the MI355X hot path (SURVEY.md section 8a): " + ", ".join(unsupported))
        self.resolution = m.resolution
        self.dx_cond = bool(_get(m, "dx_cond", False))
        self.cat_dx = bool(_get(m, "cat_dx", False))
        dx_mode = _lib.DX_NONE if not self.dx_cond else (_lib.DX_CAT if self.cat_dx else _lib.DX_ENC)
        # adm_blocks.py:236-238: cond and (cat_dx) dx are stacked to the input
        self.in_channels = m.in_channels + cond_channels + (m.in_channels if dx_mode == _lib.DX_CAT else 0)
        self.cond_channels = cond_channels
        self.state_channels = m.in_channels
        self.out_channels = m.out_ch
        self.cat_condition = True
        self.self_condition = False
        self.label_dropout = m.label_dropout
        self._arch = dict(in_channels=m.in_channels, cond_channels=cond_channels, out_channels=m.out_ch, ch=ch,
                          ch_mult=mult, num_res_blocks=m.num_res_blocks, attn_resolutions=tuple(m.attn_resolutions),
                          resolution=m.resolution, dx_channels=m.in_channels if self.dx_cond else 0, dx_mode=dx_mode)
        self.map_noise = PositionalEmbedding(ch)
        self.map_augment = None
        self.map_layer0 = Linear(ch, ch)
        self.map_layer1 = Linear(ch, ch)
        self.map_label = None
        self.cond_enc = self.dx_enc = self.combine_enc = None
        if dx_mode == _lib.DX_ENC:           # adm_blocks.py:266-280 (registered before self.enc: state_dict order)
            c0 = ch * mult[0]
            self.dx_enc = nn.Sequential(Conv2d(m.in_channels, c0, 3), nn.GELU(), Conv2d(c0, c0, 3))
            self.combine_enc = Conv2d(2 * c0, c0, 3)
        self.enc = nn.ModuleDict()
        cout = self.in_channels
        skips = []
        for level, mu in enumerate(mult):
            res = m.resolution >> level
            if level == 0:
                self.enc[f"{res}x{res}_conv"] = Conv2d(cout, ch * mu, 3)
                cout = ch * mu
            else:
                self.enc[f"{res}x{res}_down"] = UNetBlock(cout, cout, ch, down=True)
            skips.append(cout)
            for idx in range(m.num_res_blocks):
                cin, cout = cout, ch * mu
                self.enc[f"{res}x{res}_block{idx}"] = UNetBlock(cin, cout, ch, attention=(res in m.attn_resolutions))
                skips.append(cout)
        self.dec = nn.ModuleDict()
        for level, mu in reversed(list(enumerate(mult))):
            res = m.resolution >> level
            if level == len(mult) - 1:
                self.dec[f"{res}x{res}_in0"] = UNetBlock(cout, cout, ch, attention=True)
                self.dec[f"{res}x{res}_in1"] = UNetBlock(cout, cout, ch)
            else:
                self.dec[f"{res}x{res}_up"] = UNetBlock(cout, cout, ch, up=True)
            for idx in range(m.num_res_blocks + 1):
                cin, cout = cout + skips.pop(), ch * mu
                self.dec[f"{res}x{res}_block{idx}"] = UNetBlock(cin, cout, ch, attention=(res in m.attn_resolutions))
        self.out_norm = GroupNorm(cout)
        self.out_conv = Conv2d(cout, m.out_ch, 3, gain_w=0, gain_b=0)
        # runtime state (not part of the state_dict)
        self._plan: Optional[_lib.Plan] = None
        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None
        self._ws = _lib.Workspace()

    # ---- HIP plumbing -------------------------------------------------------------------------------
    @property
    def plan(self) -> _lib.Plan:
        if self._plan is None:
            self._plan = _lib.Plan(**self._arch)
            names = [n for n, _ in self.named_parameters()]
            if names != self._plan.param_names:
                raise RuntimeError("parameter table of the HIP plan and of the module disagree")
        return self._plan

    def named_param_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self.named_parameters())

    def invalidate_packed(self) -> None:
        """Force a re-pack at the next use.  The cache key below is (data_ptr, autograd version) per parameter; a write
        through ``p.data`` (``p.data.mul_()``, ``p.data.copy_()``, some clip / init utilities) bumps neither, so any code
        that writes parameters that way must call this.  ``load_state_dict`` and the fused trainer do."""
        self._packed_key = None

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)

    def packed_weights(self) -> torch.Tensor:
        """MFMA-ordered copies of the weights; re-packed whenever a parameter was written (autograd-visible), moved or
        REPLACED: the key walks the live ``self.parameters()`` and holds (id, data_ptr, version) of each, so a new
        ``nn.Parameter`` object put in by ``setattr`` / parametrize / pruning is seen (walking ~200 parameters costs tens
        of microseconds per sampling call or training step)."""
        key = tuple((id(p), p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is not None and key == self._packed_key:
            return self._packed
        params = self.named_param_dict()
        reuse = self._packed is not None and self._packed.device == next(iter(params.values())).device
        self._packed = self.plan.pack(params, self._packed if reuse else None)
        self._packed_key = key
        return self._packed

    def _check_extra(self, x_self_cond, dx, class_labels, augment_labels):
        if x_self_cond is not None or class_labels is not None or augment_labels is not None:
            raise NotImplementedError("x_self_cond / class_labels / augment_labels are outside the hot path")
        if dx is not None and not self.dx_cond:
            raise NotImplementedError("dx given to a network built with dx_cond=False (the reference ignores it silently)")

    def forward(self, x, noise_labels, cond=None, x_self_cond=None, dx=None, class_labels=None, augment_labels=None):
        self._check_extra(x_self_cond, dx, class_labels, augment_labels)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("autograd through DhariwalUNet.forward goes through PlMcedm.training_step's fused "
                                      "loss (mcedm_amd.mcedm); call it under torch.no_grad() for inference")
        x = x.to(torch.float32).contiguous()
        labels = noise_labels.to(torch.float32).reshape(-1).contiguous()
        cond = cond.to(torch.float32).contiguous() if cond is not None else None
        dx = dx.to(torch.float32).contiguous() if dx is not None else None
        return self.plan.forward(self.packed_weights(), x, labels, cond=cond, ws=self._ws, dx=dx)


class EmaModel(nn.Module):
    """models/ddim_blocks.py:38-59: deep copy whose parameters follow ma = beta*ma + (1-beta)*w."""

    def __init__(self, model, beta):
        super().__init__()
        import copy
        self.beta = beta
        plan, packed, model._plan, model._packed = model._plan, model._packed, None, None   # not deep-copyable
        self.ma_model = copy.deepcopy(model)
        model._plan, model._packed = plan, packed
        self.ma_model._packed_key = None

    def update(self, current_model):
        if isinstance(current_model, nn.parallel.DistributedDataParallel):
            current_model = current_model.module
        with torch.no_grad():
            cur_p = [c for c in current_model.parameters()]
            ma_p = [m for m in self.ma_model.parameters()]
            live = [(c, m) for c, m in zip(cur_p, ma_p) if c.requires_grad]
            if live:           # ma = ma * beta + (1 - beta) * cur, as two multi-tensor launches instead of 3 per parameter
                mas = [m for _, m in live]
                torch._foreach_mul_(mas, self.beta)
                torch._foreach_add_(mas, [c.detach() for c, _ in live], alpha=1 - self.beta)

    def forward(self, *args, **kwargs):
        return self.ma_model(*args, **kwargs)
