"""Device-side PDE residual metrics: drop-in for the reference's ``models/pde_loss.py`` (forward residual only).

``SweFvLoss`` / ``DarcyLoss`` keep the reference's constructor and ``forward(pred, gt, normalizer_h, normalizer_u,
return_d=False, calc_prob=False, clamp_loss=False)`` signature and return bit-identical tensors; the stencil work runs
in ``libmcedm_hip.so`` (csrc/pde.hip).  ``return_d=True`` (the guidance gradient) runs the stencils' analytic adjoints
(csrc/pde.hip) and agrees with the reference's autograd result to rounding.  ``get_pde_loss_function`` mirrors ``models/loss_helper.py:16-38``.
"""
import numpy as np
import torch
from torch import nn

from . import lib


def flip_state(pred, gt, normalizer_h, normalizer_u):
    """models/pde_loss.py:6-17: undo the (h, u) -> (u, h) flip of the data module."""
    h_ch = len(normalizer_h.subtract) if len(normalizer_h.subtract.size()) > 0 else 1
    u_ch = len(normalizer_u.subtract) if len(normalizer_u.subtract.size()) > 0 else 1
    pred = torch.cat([pred[..., h_ch:u_ch + h_ch], pred[..., :h_ch]], dim=-1)
    gt = torch.cat([gt[..., h_ch:u_ch + h_ch], gt[..., :h_ch]], dim=-1)
    return pred, gt


def _f32(t):
    return t.to(torch.float32).contiguous()


class SweFvLoss(nn.Module):
    """models/pde_loss.py:91-249 (FORCE finite-volume residual of the 1-D shallow-water equations)."""

    def __init__(self, Tn=0.128, x_min=-2.5, x_max=2.5, n_ghosts=2, reduction="none", flip_xy=False):
        super().__init__()
        self.flip_xy = flip_xy
        self.g = 1.0
        self.Tn = Tn
        self.x_min = x_min
        self.x_max = x_max
        self.n_ghosts = n_ghosts
        self.eps = 1e-8
        if n_ghosts < 1:
            raise ValueError("n_ghosts must be >= 1")

    def gen_x(self, nx, s_t):
        step = (self.x_max - self.x_min) / nx
        n = nx + 2 * self.n_ghosts
        if n % 2 == 0:
            x = torch.linspace(self.x_min + step / 2 - step * self.n_ghosts, self.x_max - step / 2 + step * self.n_ghosts, n)
        else:
            x = torch.linspace(self.x_min - step * self.n_ghosts, self.x_max + step * self.n_ghosts, n)
        return x.type_as(s_t)

    def _dx(self, nx):
        x = self.gen_x(nx, torch.zeros((), dtype=torch.float32))          # host, fp32: the reference's own expression
        return float(x[1] - x[0])

    def f_t_swp1d(self, s_t, dt):
        s = _f32(s_t)
        return lib.swe_fv_step(s, float(np.float32(0.5 * dt)), self._dx(s.shape[2]))

    def get_scaling(self, normalizer_h, normalizer_u):
        scale_h, scale_u = normalizer_h.divide, normalizer_u.divide
        scale = torch.stack((scale_u, scale_h), dim=-1) if self.flip_xy else torch.stack((scale_h, scale_u), dim=-1)
        return scale ** 2

    def calculate_loss(self, pred, gt, normalizer_h, normalizer_u, clamp_loss=False):
        pred, gt = _f32(pred), _f32(gt)
        scale = self.get_scaling(normalizer_h, normalizer_u).to(torch.float32).reshape(-1).cpu()
        if scale.numel() != 2:
            raise NotImplementedError("per-channel normaliser statistics with more than one channel per field")
        dt = self.Tn / pred.shape[1]
        return lib.swe_fv_residual(pred, gt, float(np.float32(0.5 * dt)), self._dx(pred.shape[2]), float(scale[0]),
                                   float(scale[1]), clamp_loss)

    def _scales(self, normalizer_h, normalizer_u):
        scale = self.get_scaling(normalizer_h, normalizer_u).to(torch.float32).reshape(-1).cpu()
        if scale.numel() != 2:
            raise NotImplementedError("per-channel normaliser statistics with more than one channel per field")
        return float(scale[0]), float(scale[1])

    def guidance_desc(self, normalizer_h, normalizer_u, n_times: int, nx: int, weight: float = 5.0):
        """C description of this residual for the guided sampler (mcedm_heun_sample_guided)."""
        if self.flip_xy:
            raise NotImplementedError("flip_xy with device-side guidance")
        sub = [float(normalizer_h.subtract), float(normalizer_u.subtract)]
        div = [float(normalizer_h.divide), float(normalizer_u.divide)]
        return lib.GuidanceDesc(1, float(np.float32(0.5 * (self.Tn / n_times))), self._dx(nx), 0.0, sub[0], div[0], sub[1], div[1],
                                float(weight))

    def forward(self, pred, gt, normalizer_h, normalizer_u, return_d=False, calc_prob=False, clamp_loss=False):
        if self.flip_xy:
            pred, gt = flip_state(pred, gt, normalizer_h, normalizer_u)
        if return_d:           # the guidance gradient (models/pde_loss.py:231-242); calc_prob is ignored by the reference here
            pred, gt = _f32(pred), _f32(gt)
            s2h, s2u = self._scales(normalizer_h, normalizer_u)
            dt = self.Tn / pred.shape[1]
            return lib.swe_fv_guidance(pred, gt, float(np.float32(0.5 * dt)), self._dx(pred.shape[2]), s2h, s2u)
        return self.calculate_loss(pred, gt, normalizer_h, normalizer_u, clamp_loss)


class SweSimulatorLoss(SweFvLoss):
    """models/loss_helper.py:3-11: the finite-volume loss stands in when the simulator loss is unavailable."""


class DarcyLoss(nn.Module):
    """models/pde_loss.py:20-88 (steady Darcy flow, -div(a grad u) = 1)."""

    def __init__(self, reduction="none", flip_xy=False):
        super().__init__()
        self.flip_xy = flip_xy
        self.D = 1.0
        self.eps = 1e-8

    def guidance_desc(self, normalizer_h, normalizer_u, n_times: int, nx: int, weight: float = 5.0):
        if self.flip_xy:
            raise NotImplementedError("flip_xy with device-side guidance")
        if n_times != nx:
            raise ValueError("DarcyLoss expects a square grid")
        sub = [float(normalizer_h.subtract), float(normalizer_u.subtract)]
        div = [float(normalizer_h.divide), float(normalizer_u.divide)]
        return lib.GuidanceDesc(2, 0.0, 0.0, float(np.float32(2 * (self.D / nx))), sub[0], div[0], sub[1], div[1], float(weight))

    def forward(self, pred, gt, normalizer_h, normalizer_u, return_d=False, calc_prob=False, clamp_loss=False):
        if self.flip_xy:
            pred, gt = flip_state(pred, gt, normalizer_h, normalizer_u)
        pred = _f32(pred)
        size = pred.shape[1]
        if pred.shape[2] != size or pred.shape[-1] != 2:
            raise ValueError("DarcyLoss expects (b, s, s, 2) = (a, u)")
        if return_d:           # the guidance gradient (models/pde_loss.py:60-75)
            return lib.darcy_guidance(pred, float(np.float32(2 * (self.D / size))), bool(calc_prob))
        n = size - 4
        return lib.darcy_residual(pred, float(np.float32(2 * (self.D / size))), float(n * n), clamp_loss)


def get_pde_loss_function(system, flip_xy, Tn_mult=1.0):
    """models/loss_helper.py:14-38."""
    if system == "swe_per":
        Tn, x_min, x_max = 0.128 * Tn_mult, -0.5, 0.5
        return (SweFvLoss(Tn=Tn, x_min=x_min, x_max=x_max, flip_xy=flip_xy),
                SweSimulatorLoss(Tn=Tn, x_min=x_min, x_max=x_max, flip_xy=flip_xy))
    if system == "darcy":
        return DarcyLoss(flip_xy=flip_xy), DarcyLoss(flip_xy=flip_xy)
    if system == "reactor":
        raise NotImplementedError("ReactorLoss is not defined in the reference either (models/loss_helper.py:29-32)")
    Tn = 1.28 * Tn_mult          # "swe" and the default
    return SweFvLoss(Tn=Tn, flip_xy=flip_xy), SweSimulatorLoss(Tn=Tn, flip_xy=flip_xy)
