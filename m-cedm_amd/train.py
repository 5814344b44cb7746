"""Data-parallel EDM training step on flat parameter buffers.

One process per GPU.  Per step (models/mcedm.py:254-281 + Lightning's DDP / clip / Adam / EmaModel.update):

    x_noise, sigma      <- mcedm_edm_noise_inputs            (HIP)
    D                   <- mcedm_edm_denoise(training)        (HIP, activations kept in the workspace)
    loss, dD            <- mcedm_edm_loss                     (HIP)
    grads (flat)        <- mcedm_edm_denoise_backward         (HIP)
    grads               <- all-reduce(sum) over ranks          (RCCL via torch.distributed; ONE message: the flat buffer)
    |g|^2               <- mcedm_sqnorm                        (HIP)
    params, m, v, ema   <- mcedm_adam_ema_step                 (HIP: 1/world scaling + clip + Adam + EMA fused)

The flat layout is the parameter order of ``DhariwalUNet.state_dict()``; each ``nn.Parameter`` of the model (and of the
EMA copy) is re-pointed to a view of the flat buffer, so ``state_dict()`` / checkpoints are unchanged.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import lib as _lib


def shard_range(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_mean_(flat: torch.Tensor, world: Optional[int] = None, average: bool = True) -> torch.Tensor:
    """In-place sum all-reduce of one flat buffer (RCCL on GPUs, gloo on CPU), optionally divided by world."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.div_(world or dist.get_world_size())
    return flat


def flatten_params_(params: Sequence[torch.nn.Parameter]) -> torch.Tensor:
    """Move the parameters into one contiguous fp32 buffer and re-point each one at its slice."""
    params = list(params)
    dev = params[0].device
    flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
    return flat


def views_like(flat: torch.Tensor, params: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    out, off = [], 0
    for p in params:
        out.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    return out


class EdmTrainer:
    """Fused training step for a ``mcedm_amd.mcedm.PlMcedm`` (or anything with ``.model`` / ``.ema_model``)."""

    def __init__(self, module, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, clip=1.0, ema_beta=0.999,
                 P_mean=-1.2, P_std=1.2, sigma_data=1.0):
        self.module = module
        self.net = module.model
        self.hp = dict(lr=lr, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, max_norm=clip,
                       ema_beta=ema_beta)
        self.P_mean, self.P_std, self.sigma_data = P_mean, P_std, sigma_data
        params = list(self.net.parameters())
        self.flat_p = flatten_params_(params)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.flat_ema = None
        if getattr(module, "ema_model", None) is not None:
            self.flat_ema = flatten_params_(list(module.ema_model.ma_model.parameters()))
        self.grad_views = views_like(self.flat_g, params)
        self.sq = torch.zeros(1, dtype=torch.float64, device=self.flat_p.device)
        self.step_count = 0
        self.ws = _lib.Workspace()
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def step(self, x, cond_in, mask, noise, rnd_normal):
        """One optimisation step on this rank's shard (all tensors NCHW fp32 on the device). Returns the local loss."""
        net = self.net
        plan, packed = net.plan, net.packed_weights()
        x_noise, sigma = _lib.edm_noise_inputs(x, mask, noise, rnd_normal.reshape(-1).contiguous(), self.P_mean, self.P_std)
        D = plan.denoise(packed, x_noise, sigma, cond=cond_in, ws=self.ws, training=True, sigma_data=self.sigma_data)
        loss, dD = _lib.edm_loss(D, x, mask, sigma, sigma_data=self.sigma_data, want_grad=True)
        plan.denoise_backward(packed, net.named_param_dict(), x_noise, sigma, cond_in, dD, self.grad_views, self.ws,
                              sigma_data=self.sigma_data)
        allreduce_mean_(self.flat_g, average=False)                       # sum; the 1/world goes into the fused step
        _lib.sqnorm(self.flat_g, self.sq)
        self.step_count += 1
        _lib.adam_ema_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.flat_ema, self.step_count,
                           sqnorm_t=self.sq, grad_scale=1.0 / self.world, **self.hp)
        # the kernel wrote parameters (and the EMA copy) in place behind autograd's back: drop the packed copies
        net._packed_key = None
        if self.flat_ema is not None:
            self.module.ema_model.ma_model._packed_key = None
        return loss
