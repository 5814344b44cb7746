"""``FusedAdamEma``: the fused clip + Adam + EMA kernels (K11: mcedm_sqnorm, mcedm_adam_ema_step) behind the reference's own seam.

The reference's training step is driven by Lightning (models/mcedm.py:139-168, configs/trainer/trainer_ddim.yaml:8-9):

    optimizer.zero_grad(); loss = training_step(...); loss.backward()      [DDP averages the gradients]
    configure_gradient_clipping -> clip_grad_norm_(parameters, 1.0)
    optimizer_step -> optimizer.step()  (torch.optim.Adam)  ->  ema_model.update(model)   (3 launches per parameter)

``PlMcedm.configure_optimizers`` returns this optimiser for ``optimizer: Adam`` (``MCEDM_FUSED_OPT=0`` keeps torch.optim.Adam).
It IS a ``torch.optim.Optimizer`` -- Lightning, LR schedulers, ``state_dict()`` / ``load_state_dict()`` (in ``torch.optim.Adam`` form,
so checkpoints move freely between this optimiser, the reference's and ``train.EdmTrainer``) see nothing unusual -- but the
model's parameters (and the EMA copy's) are views of ONE flat fp32 buffer each, and ``step()`` is two kernels over the flat
buffers: the squared gradient norm (fixed-order reduction) and clip + Adam + EMA fused.  ``PlMcedm.configure_gradient_clipping``
hands the clip value over instead of running ``clip_grad_norm_`` and ``PlMcedm.optimizer_step`` skips the separate EMA pass.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import lib as _lib
from .train import flatten_params_, views_like


class FusedAdamEma(torch.optim.Optimizer):
    def __init__(self, model: torch.nn.Module, ema_model: Optional[torch.nn.Module] = None, lr=2e-4, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, amsgrad=False, ema_beta=0.999, max_norm: Optional[float] = None):
        if amsgrad:
            raise NotImplementedError("amsgrad=True (the reference configures amsgrad: False, configs/model/*.yaml)")
        params = [p for p in model.parameters()]
        if not params or any(not p.is_cuda or p.dtype != torch.float32 for p in params):
            raise RuntimeError("FusedAdamEma needs fp32 parameters on the device (move the module first)")
        if any(not p.requires_grad for p in params):
            raise NotImplementedError("frozen parameters: use torch.optim.Adam (MCEDM_FUSED_OPT=0)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None))
        self._params = params
        self._nets = [model]
        # the parameters become views of one flat buffer (values unchanged, state_dict keys unchanged)
        self.flat_p = flatten_params_(params)
        self.flat_ema = None
        self.ema_beta = float(ema_beta)
        if ema_model is not None:
            eparams = [p for p in ema_model.parameters()]
            if [tuple(p.shape) for p in eparams] != [tuple(p.shape) for p in params]:
                raise RuntimeError("the EMA copy's parameters do not mirror the model's")
            self.flat_ema = flatten_params_(eparams)
            self._nets.append(ema_model)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self._flat_g = torch.zeros_like(self.flat_p)
        self._g_views = views_like(self._flat_g, params)
        self._m_views = views_like(self.flat_m, params)
        self._v_views = views_like(self.flat_v, params)
        self.max_norm = max_norm                  # set per step by PlMcedm.configure_gradient_clipping; None: no clipping
        self.grad_scale = 1.0                     # gradients arrive averaged (DDP); train.EdmTrainer folds 1 / world in here instead
        self.step_count = 0
        self._sq = torch.zeros(1, dtype=torch.float64, device=self.flat_p.device)
        self._scratch = torch.empty(_lib.REDUCE_SCRATCH_BYTES, dtype=torch.uint8, device=self.flat_p.device)
        self.last_grad_norm = None                # device fp64 scalar: sqrt of it is the total norm before clipping

    # ---- gradients: zero-copy when autograd handed over views of ONE flat tensor in parameter order (what _EdmTrainLoss.backward
    # returns, and what DDP leaves in place), else one multi-tensor copy into the optimiser's own flat buffer
    def _flat_grad(self) -> torch.Tensor:
        ps = self._params
        g0 = ps[0].grad
        if g0 is None:
            raise RuntimeError("FusedAdamEma.step: a parameter has no gradient (every parameter of the network gets one per step)")
        base = g0._base if g0._base is not None else None
        if base is not None and base.dtype == torch.float32 and base.is_contiguous() and base.numel() == self.flat_p.numel():
            off, ok = 0, True
            for p in ps:
                g = p.grad
                if g is None or g._base is not base or g.storage_offset() != base.storage_offset() + off or not g.is_contiguous():
                    ok = False
                    break
                off += p.numel()
            if ok:
                return base
        grads = []
        for p in ps:
            if p.grad is None:
                raise RuntimeError("FusedAdamEma.step: a parameter has no gradient")
            grads.append(p.grad)
        torch._foreach_copy_(self._g_views, grads)
        return self._flat_g

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:                   # Lightning's automatic optimisation: zero_grad + training_step + backward
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        flat_g = self._flat_grad()
        self.step_count += 1
        clip = self.max_norm is not None and self.max_norm > 0
        if clip:
            _lib.sqnorm(flat_g, self._sq, scratch=self._scratch)
            self.last_grad_norm = self._sq
        _lib.adam_ema_step(self.flat_p, flat_g, self.flat_m, self.flat_v, self.flat_ema, self.step_count, lr=float(g["lr"]),
                           beta1=float(g["betas"][0]), beta2=float(g["betas"][1]), eps=float(g["eps"]),
                           weight_decay=float(g["weight_decay"]), sqnorm_t=self._sq if clip else None,
                           max_norm=float(self.max_norm) if clip else 1.0, grad_scale=float(self.grad_scale),
                           ema_beta=self.ema_beta)
        # the kernel wrote parameters (and the EMA copy) behind autograd's back: version counters did not move
        for net in self._nets:
            if hasattr(net, "invalidate_packed"):
                net.invalidate_packed()
        if not self.state:                        # torch.optim.Adam's per-parameter state, as views of the flat moments
            for p, m, v in zip(self._params, self._m_views, self._v_views):
                self.state[p] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m, "exp_avg_sq": v}
        else:
            for p in self._params:
                self.state[p]["step"].fill_(float(self.step_count))
        return loss

    def load_state_dict(self, state_dict):
        """Accepts a ``torch.optim.Adam`` state dict (what Lightning stores as ``optimizer_states[0]``): the moments are copied
        into the flat buffers and the state entries re-pointed at their views."""
        super().load_state_dict(state_dict)
        if any(g.get("amsgrad", False) for g in self.param_groups):
            raise NotImplementedError("amsgrad state")
        steps = set()
        with torch.no_grad():
            for p, m, v in zip(self._params, self._m_views, self._v_views):
                st = self.state.get(p)
                if not st:
                    continue
                if st["exp_avg"].data_ptr() != m.data_ptr():
                    m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                    st["exp_avg"], st["exp_avg_sq"] = m, v
                st["step"] = torch.as_tensor(float(st["step"])).cpu()
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise RuntimeError(f"optimizer state: parameters disagree on the step count ({sorted(steps)})")
        if steps:
            self.step_count = steps.pop()
        else:
            self.flat_m.zero_(); self.flat_v.zero_(); self.step_count = 0
