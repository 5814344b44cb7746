"""Drop-ins for the reference's ``models/ddim.py``:

``PlDdim`` (bottom of this file): the joint DDPM baseline sampled with the EDM Heun sampler and
RePaint-style resampling -- ``sample_edm`` (models/ddim.py:959-1051), ``get_denoised`` (:915-947), ``round_sigma``
(:949-957), ``compute_alpha`` (:700-704) on the DDPM U-Net of ``m-cedm_amd/ddim_blocks.py``; the loop runs in
``mcedm_repaint_sample`` (csrc/ddpm.hip).

``PlCondEdm`` for ``models/ddim.py:1608-1773`` (single-task conditional EDM: the
conditioning field h is given, the state u is generated) -- SURVEY.md section 8(f2).

It runs on the same HIP path as ``PlMcedm``: the same ``DhariwalUNet`` (``in_channels`` 1 + ``cond_channels`` 1 ->
``out_ch`` 1 in ``configs/model/adm_edm_cond_h_res32.yaml``), the same EDM preconditioning, and the UNMASKED variants of
the loss and of the Heun sampler (``mask = NULL`` in the C ABI).  Constructor, attributes, state_dict keys (incl. the
DDPM-schedule buffers ``betas`` / ``logvar`` that ``PlDdim.__init__`` registers, models/ddim.py:22-30) and method
signatures follow the reference; DDIM sampling, PDE guidance, self-conditioning and the ``node_type`` channel are outside
the hot path and raise.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import lib as _lib
from .adm_blocks import DhariwalUNet, EmaModel
from .mcedm import DotDict, Normalizer, _Base, _EdmTrainLoss, _nchw


def _opt(cfg, name, default):
    """cfg.<name> if present (DictConfig, attribute dicts whose __getattr__ raises KeyError, plain objects)."""
    try:
        return getattr(cfg, name)
    except (AttributeError, KeyError):
        return default


def _beta_schedule(kind, beta_start, beta_end, n):
    """models/ddim_blocks.py:473-505."""
    if kind == "quad":
        b = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2
    elif kind == "linear":
        b = np.linspace(beta_start, beta_end, n, dtype=np.float64)
    elif kind == "const":
        b = beta_end * np.ones(n, dtype=np.float64)
    elif kind == "jsd":
        b = 1.0 / np.linspace(n, 1, n, dtype=np.float64)
    elif kind == "sigmoid":
        b = 1 / (np.exp(-np.linspace(-6, 6, n)) + 1) * (beta_end - beta_start) + beta_start
    else:
        raise NotImplementedError(kind)
    return torch.from_numpy(b).float()


class PlCondEdm(_Base):
    def __init__(self, hparams):
        super().__init__()
        self.save_hyperparameters()
        m, o, d, df = hparams.model, hparams.optimization, hparams.data, hparams.diffusion
        for flag in ("dx_cond", "node_type", "self_cond"):
            if hasattr(m, flag) and getattr(m, flag):
                raise NotImplementedError(f"hparams.model.{flag}=True is outside the MI355X hot path")
        if not str(hparams.name).startswith("adm"):
            raise NotImplementedError("only the ADM/EDM U-Net (hparams.name = 'adm*') is on the hot path")
        # DDPM schedule buffers of PlDdim (kept for checkpoint compatibility; the EDM path never reads them)
        betas = _beta_schedule(df.beta_schedule, df.beta_start, df.beta_end, df.num_diffusion_timesteps)
        acp = (1.0 - betas).cumprod(dim=0)
        post_var = betas * (1.0 - torch.cat([torch.ones(1), acp[:-1]])) / (1.0 - acp)
        self.model_var_type = m.var_type
        self.register_buffer("betas", betas)
        self.num_timesteps = betas.shape[0]
        if m.var_type == "fixedlarge":
            self.register_buffer("logvar", betas.log())
        elif m.var_type == "fixedsmall":
            self.register_buffer("logvar", post_var.clamp(min=1e-20).log())
        self.cond_p = m.cond_p if hasattr(m, "cond_p") else 0.8
        self.dx_cond = self.node_type = False
        self.model = DhariwalUNet(hparams)
        self.ema_model = EmaModel(self.model, beta=m.ema_rate) if m.ema else None
        self.normalization, self.rescaled = d.normalization, d.rescaled
        self.uniform_dequantization, self.gaussian_dequantization = d.uniform_dequantization, d.gaussian_dequantization
        self.normalizer_input = Normalizer((m.in_channels,) if m.in_channels > 1 else ())
        self.normalizer_target = Normalizer((m.out_ch,) if m.out_ch > 1 else ())
        self.optimizer, self.lr, self.weight_decay = o.optimizer, o.lr, o.weight_decay
        self.beta1, self.amsgrad, self.eps = o.beta1, o.amsgrad, o.eps
        if hasattr(o, "pde_loss_lambda") and o.pde_loss_lambda:
            raise NotImplementedError("pde_loss_lambda != 0 is outside the hot path")
        from .pde_loss import get_pde_loss_function
        self.pde_loss, self.pde_loss_simulator = get_pde_loss_function(system="swe", flip_xy=False)    # models/ddim.py:76-78
        self.P_mean, self.P_std, self.sigma_data = -1.2, 1.2, 1.0
        self.sigma_min, self.sigma_max = 0.002, 80
        self.sparams = hparams.sampler if hparams.get("sampler", None) is not None else self.get_edm_sampler_params()
        self.test_sparams = self.sparams
        self.h_ch, self.u_ch = m.cond_channels, m.out_ch
        self._train_ws, self._sample_ws, self._grad_buf = _lib.Workspace(), _lib.Workspace(), None
        self._train_generation = 0
        self._graphs = {}

    # ---- configuration ----------------------------------------------------------------------------------
    @staticmethod
    def get_edm_sampler_params():
        return DotDict(name="edm", type="edm", timesteps=50, sigma_min=0.002, sigma_max=80, rho=7, S_churn=15.0, S_min=0,
                       S_max="inf", S_noise=1, n_samples=5, n_repeat=2, n_time_h=128, n_time_u=0, return_last=True,
                       select_by_pde=False, use_gt_pde_select=True, guide_dx=False, w=0.0, plot_scaled=False)

    def set_test_sampler_params(self, params):
        if params.type != "edm":
            print("Model with EDM preconditioning supports only EDM sampler ")
            params = self.get_edm_sampler_params()
        self.test_sparams = params

    def set_pde_loss_function(self, system, flip_xy):
        """models/ddim.py:97-101; the residuals and their guidance gradients run on the device (m-cedm_amd/pde_loss.py)."""
        from .pde_loss import get_pde_loss_function
        self.pde_loss, self.pde_loss_simulator = get_pde_loss_function(system, flip_xy)

    def setup(self, stage: str = None) -> None:
        if stage == "fit":
            st = self.trainer.datamodule.get_norm_stats()
            key = ("min", "min_max") if self.normalization == "min_max" else ("mean", "std")
            self.normalizer_input.set_stats(st[f"input_{key[0]}"], st[f"input_{key[1]}"])
            self.normalizer_target.set_stats(st[f"target_{key[0]}"], st[f"target_{key[1]}"])

    def configure_optimizers(self):
        if self.optimizer != "Adam":
            raise NotImplementedError(f"Optimizer {self.optimizer} not understood.")
        return {"optimizer": torch.optim.Adam(self.model.parameters(), lr=self.lr, weight_decay=self.weight_decay,
                                              betas=(self.beta1, 0.999), amsgrad=self.amsgrad, eps=self.eps)}

    def optimizer_step(self, *args, **kwargs):
        super().optimizer_step(*args, **kwargs)
        if self.ema_model is not None:
            self.ema_model.update(self.model)

    # ---- data ------------------------------------------------------------------------------------------------
    def data_transform(self, h, u):
        x = torch.cat([self.normalizer_input(h), self.normalizer_target(u)], dim=-1)
        if self.uniform_dequantization:
            x = x / 256.0 * 255.0 + torch.rand_like(x) / 256.0
        if self.gaussian_dequantization:
            x = x + torch.randn_like(x) * 0.01
        return 2 * x - 1.0 if self.rescaled else x

    def inverse_data_transform_u(self, u):
        if self.rescaled:
            u = (u + 1.0) / 2.0
        if self.normalization == "min_max":
            u = torch.clamp(u, 0.0, 1.0)
        return self.normalizer_target(u, inverse=True)

    def get_cond_in(self, h, u, dx, dt):
        """models/ddim.py:1081-1116 (node_type False)."""
        cc = self.model.cond_channels
        if cc == self.h_ch:
            return h
        u_ic = u[:, 0:1].repeat(1, u.shape[1], 1, 1)
        if cc == self.h_ch + self.u_ch:
            return torch.cat([h, u_ic], dim=-1)
        if cc == self.h_ch + 2:
            return torch.cat([h, dt, dx], dim=-1)
        if cc == self.h_ch + self.u_ch + 2:
            return torch.cat([h, u_ic, dt, dx], dim=-1)
        raise RuntimeError(f"Number of conditional channels {cc} does not match the known state channels {self.h_ch}")

    def get_loss_weight(self, sigma):
        return (sigma ** 2 + self.sigma_data ** 2) / (sigma * self.sigma_data) ** 2

    # ---- HIP path -------------------------------------------------------------------------------------------
    def _net(self, model):
        return model.ma_model if isinstance(model, EmaModel) else model

    def _grad_views(self, params):
        n = sum(p.numel() for p in params)
        if self._grad_buf is None or self._grad_buf.numel() != n or self._grad_buf.device != params[0].device:
            self._grad_buf = torch.empty(n, dtype=torch.float32, device=params[0].device)
        views, off = [], 0
        for p in params:
            views.append(self._grad_buf[off:off + p.numel()].view_as(p))
            off += p.numel()
        return views

    def model_precond(self, x_noise, sigma, cond=None, x_self_cond=None, dx=None):
        if x_self_cond is not None or dx is not None:
            raise NotImplementedError("x_self_cond / dx are outside the hot path")
        net = self.model
        with torch.no_grad():
            return net.plan.denoise(net.packed_weights(), x_noise.float().contiguous(),
                                    sigma.to(torch.float32).reshape(-1).contiguous(),
                                    cond=None if cond is None else cond.float().contiguous(), ws=net._ws,
                                    sigma_data=self.sigma_data)

    def get_denoised(self, model, xt, t, cond=None, x_self_cond=None, dx=None, w=None):
        if x_self_cond is not None or dx is not None:
            raise NotImplementedError("x_self_cond / dx are outside the hot path")
        net = self._net(model)
        xt = xt.to(torch.float32).contiguous()
        sigma = torch.as_tensor(t).to(torch.float32).reshape(-1).contiguous().to(xt.device)
        cond = None if cond is None else cond.float().contiguous()
        pk = net.packed_weights()
        with torch.no_grad():
            D, F = net.plan.denoise(pk, xt, sigma, cond=cond, ws=net._ws, sigma_data=self.sigma_data, want_F=True)
            if not (w is None or abs(w) < 0.001 or cond is None):
                _, Fu = net.plan.denoise(pk, xt, sigma, cond=None, ws=net._ws, sigma_data=self.sigma_data, want_F=True)
                F = (w + 1) * F - w * Fu
                s = sigma.reshape(-1, 1, 1, 1)
                D = self.sigma_data ** 2 / (s ** 2 + self.sigma_data ** 2) * xt + \
                    s * self.sigma_data / (s ** 2 + self.sigma_data ** 2).sqrt() * F
        return D, F

    def training_step(self, train_batch, batch_idx):
        h_unnorm, dx, dt, u_unnorm = train_batch
        self.h_ch, self.u_ch = h_ch, u_ch = h_unnorm.shape[-1], u_unnorm.shape[-1]
        x = self.data_transform(h_unnorm, u_unnorm)
        h, u = x[..., 0:h_ch], x[..., h_ch:h_ch + u_ch]
        cond_in = _nchw(self.get_cond_in(h, u, dx, dt)).float()
        u = _nchw(u).float()
        noise = torch.randn_like(u)
        rnd_normal = torch.randn([u.shape[0], 1, 1, 1]).type_as(u)
        u_noise, sigma = _lib.edm_noise_inputs(u, None, noise, rnd_normal.reshape(-1).contiguous(), self.P_mean, self.P_std)
        torch.rand(1)                                            # the cond_p draw of models/ddim.py:1675 (cond_p = 1)
        loss = _EdmTrainLoss.apply(self, u, u_noise, sigma, cond_in, None, *self.model.parameters())
        self.log("train_loss", loss, prog_bar=True, on_epoch=True, on_step=False, sync_dist=True)
        return loss

    def get_dx_pde(self, cond, x_denoised, calc_prob=False):
        """models/ddim.py:1424-1450: gradient of the PDE residual of (h from cond, u = x_denoised), un-normalised, w.r.t. that
        state; mean (calc_prob) or sum over the two field gradients."""
        h = cond[:, :self.h_ch].to(torch.float32).permute(0, 2, 3, 1)
        u = x_denoised.to(torch.float32).permute(0, 2, 3, 1)
        h_un = self.normalizer_input((h + 1.0) / 2.0 if self.rescaled else h, inverse=True)
        u_un = self.inverse_data_transform_u(u)
        x_un = torch.cat([h_un, u_un], dim=-1).contiguous()
        d = self.pde_loss(x_un, x_un, self.normalizer_input, self.normalizer_target, True, calc_prob).permute(0, 3, 1, 2)
        return torch.mean(d, dim=1, keepdim=True) if calc_prob else torch.sum(d, dim=1)

    def get_dx_log_prob(self, cond, x_denoised, guide_dx):
        """models/ddim.py:641-650 (the residual classes already zero the NaNs of the gradient)."""
        if not guide_dx:
            return torch.zeros_like(x_denoised)
        return self.get_dx_pde(cond, x_denoised, calc_prob=True)

    def sample_edm(self, h, u_noise, sparams, return_last=True, guide_dx=False):
        """h, u_noise in the reference's 'b h w c' layout; returns [b, t, h, w, c] float64 (models/ddim.py:1532-1601).
        guide_dx=True: after every denoiser call d -= 5 * dx / t_hat with dx the PDE-residual gradient, evaluated on the
        device by the stencils' analytic adjoints (csrc/pde.hip) instead of torch.autograd."""
        guidance = None
        if guide_dx:
            if self.pde_loss is None or not hasattr(self.pde_loss, "guidance_desc"):
                raise NotImplementedError("guide_dx=True needs set_pde_loss_function('swe' | 'swe_per' | 'darcy')")
            if self.rescaled or self.normalization == "min_max" or self.h_ch != 1 or self.u_ch != 1:
                raise NotImplementedError("guide_dx=True is built for scalar gauss-normalised fields h, u")
            guidance = self.pde_loss.guidance_desc(self.normalizer_input, self.normalizer_target, h.shape[1], h.shape[2])
        net = self._net(self.ema_model if self.ema_model is not None else self.model)
        h, init = _nchw(h).float(), _nchw(u_noise).float()
        sd = _lib.sampler_desc(sparams, self.sigma_data, self.sigma_min, self.sigma_max)
        N, t = sd.timesteps, _lib.edm_t_steps(sd)
        churn = any((min(sd.S_churn / N, math.sqrt(2) - 1) if sd.S_min <= t[i] <= sd.S_max else 0) > 0 for i in range(N))
        step_noise = torch.randn((N,) + tuple(init.shape), dtype=torch.float64, device=init.device) if churn else None
        with torch.no_grad():
            return net.plan.sample(net.packed_weights(), sd, h, None, init, step_noise, return_last=return_last,
                                   ws=self._sample_ws, guidance=guidance)

    # ---- evaluation bookkeeping (models/ddim.py:1154-1330; scaled-MAE / correlation / PDE extras are host metrics)
    def _eval(self, batch, sp, n):
        h_unnorm, dx, dt, u_unnorm = batch
        self.h_ch, self.u_ch = h_ch, u_ch = h_unnorm.shape[-1], u_unnorm.shape[-1]
        state = self.data_transform(h_unnorm, u_unnorm)
        h, u = state[..., :h_ch], state[..., h_ch:h_ch + u_ch]
        cond = self.get_cond_in(h, u, dx, dt).repeat(n, 1, 1, 1)
        xs = self.sample_edm(cond, torch.randn_like(u.repeat(n, 1, 1, 1)), sp, return_last=sp.return_last, guide_dx=sp.guide_dx)
        nb = len(h_unnorm)
        xs_mean = xs.reshape(n, nb, *xs.shape[1:]).mean(dim=0)
        u_last = xs_mean[:, -1, :, :, :u_ch]
        loss_u = (u_last - u).abs().mean()
        loss_u_un = (self.inverse_data_transform_u(u_last) - u_unnorm).abs().mean()
        return xs, u, loss_u, loss_u_un, nb

    def validation_step(self, val_batch, batch_idx):
        if (self.current_epoch + 1) % 100 != 0 and self.current_epoch != 0:
            return {"epoch": self.current_epoch}
        xs, u, loss_u, loss_u_un, _ = self._eval(val_batch, self.sparams, 1)
        self.log("val_mae_u", loss_u, prog_bar=True, on_epoch=True, on_step=False, sync_dist=True)
        self.log("val_mae_u_un", loss_u_un, prog_bar=True, on_epoch=True, on_step=False, sync_dist=True)
        return {"epoch": self.current_epoch, "loss": loss_u, "loss_u_un": loss_u_un, "traj": xs[:, -1].unsqueeze(1), "gt": u}

    def test_step(self, test_batch, test_idx):
        n = self.test_sparams.n_samples
        xs, u, loss_u, loss_u_un, nb = self._eval(test_batch, self.test_sparams, n)
        print(f"\nLoss u {loss_u}, loss u un {loss_u_un}")
        self.log("test_mae_u", loss_u, prog_bar=True, on_epoch=True, on_step=False, sync_dist=True)
        self.log("test_mae_u_un", loss_u_un, prog_bar=True, on_epoch=True, on_step=False, sync_dist=True)
        last = xs[:, -1]
        traj = last.reshape(n, nb, *last.shape[1:]).permute(1, 2, 3, 0, 4).unsqueeze(1)       # '(n b) h w c -> b h w n c'
        return {"loss": loss_u, "loss_u_un": loss_u_un, "traj": traj, "gt": u}


class PlDdim(_Base):
    """models/ddim.py:16-1051, the part BASELINE config 5 exercises: EDM / RePaint sampling of the joint (h, u) DDPM.
    Constructor, buffers (``betas``, ``logvar``), attributes and the signatures of ``set_test_sampler_params``,
    ``get_edm_steps``, ``compute_alpha``, ``round_sigma``, ``get_denoised`` and ``sample_edm`` follow the reference.
    DDPM training, the DDIM ``sample`` / ``sample_with_repeat`` loops and PDE guidance are not built and raise."""

    def __init__(self, hparams):
        super().__init__()
        self.save_hyperparameters()
        m, o, d, df = hparams.model, hparams.optimization, hparams.data, hparams.diffusion
        for flag in ("dx_cond", "node_type"):
            if _opt(m, flag, False):
                raise NotImplementedError(f"hparams.model.{flag}=True is outside the built path")
        if str(hparams.name).startswith("adm"):
            raise NotImplementedError("PlDdim with the ADM U-Net is not built; use PlMcedm / PlCondEdm for ADM networks")
        from .ddim_blocks import Model
        betas = _beta_schedule(df.beta_schedule, df.beta_start, df.beta_end, df.num_diffusion_timesteps)
        acp = (1.0 - betas).cumprod(dim=0)
        post_var = betas * (1.0 - torch.cat([torch.ones(1), acp[:-1]])) / (1.0 - acp)
        self.model_var_type = m.var_type
        self.register_buffer("betas", betas)
        self.num_timesteps = betas.shape[0]
        if m.var_type == "fixedlarge":
            self.register_buffer("logvar", betas.log())
        elif m.var_type == "fixedsmall":
            self.register_buffer("logvar", post_var.clamp(min=1e-20).log())
        self.cond_p = 0.0
        self.dx_cond = self.node_type = False
        self.model = Model(hparams)
        self.ema_model = EmaModel(self.model, beta=m.ema_rate) if m.ema else None
        self.normalization, self.rescaled = d.normalization, d.rescaled
        self.uniform_dequantization, self.gaussian_dequantization = d.uniform_dequantization, d.gaussian_dequantization
        n_state = m.out_ch // 2
        self.normalizer_input = Normalizer((n_state,) if n_state > 1 else ())
        self.normalizer_target = Normalizer((n_state,) if n_state > 1 else ())
        self.optimizer, self.lr, self.weight_decay = o.optimizer, o.lr, o.weight_decay
        self.beta1, self.amsgrad, self.eps = o.beta1, o.amsgrad, o.eps
        self.sparams = hparams.sampler if hparams.get("sampler", None) is not None else \
            DotDict(type="ddim", timesteps=50, skip_type="uniform", eta=0.0, n_samples=1, n_repeat=5, n_time_h=128, n_time_u=0)
        self.test_sparams = self.sparams
        self.h_ch = self.u_ch = n_state
        self.edm_steps = None
        self.sigma_min = self.sigma_max = None
        self._sample_ws = _lib.Workspace()
        self._graphs = {}

    # ---- schedule (host side, the reference's own expressions on CPU tensors) ----------------------------------
    def set_test_sampler_params(self, params):
        self.test_sparams = params
        if params.type == "edm":                                   # models/ddim.py:125-129
            self.edm_steps = self.get_edm_steps()
            self.sigma_min = float(self.edm_steps[self.num_timesteps - 1])
            self.sigma_max = float(self.edm_steps[0])

    def get_edm_steps(self):
        """models/ddim.py:131-137, evaluated on the CPU like the schedule buffers themselves."""
        b = self.betas.detach().cpu()
        alphas_bar = (1.0 - b).cumprod(dim=0)
        return ((1 - alphas_bar) / alphas_bar).sqrt().flip(dims=(0,))

    def _alphas_ext(self):
        b = self.betas.detach().cpu()
        return (1 - torch.cat([torch.zeros(1), b], dim=0)).cumprod(dim=0)

    def compute_alpha(self, t):
        """models/ddim.py:700-704."""
        return self._alphas_ext().index_select(0, torch.as_tensor(t).cpu().reshape(-1) + 1).view(-1, 1, 1, 1)

    def round_sigma(self, sigma, return_index=False):
        """models/ddim.py:949-957 (host tensors: the schedule is scalar work)."""
        if self.edm_steps is None:
            raise RuntimeError("call set_test_sampler_params(params) with params.type == 'edm' first (models/ddim.py:122-129)")
        sigma = torch.as_tensor(sigma)
        s32 = sigma.detach().cpu().to(torch.float32)
        index = torch.cdist(s32.reshape(1, -1, 1), self.edm_steps.reshape(1, -1, 1)).argmin(2)
        result = index if return_index else self.edm_steps[index.flatten()]
        return result.to(device=sigma.device).type_as(sigma).reshape(sigma.shape)

    def _net(self, model):
        return model.ma_model if isinstance(model, EmaModel) else model

    def get_denoised(self, model, xt, t, cond=None, x_self_cond=None, dx=None, w=None):
        """models/ddim.py:915-947 at one noise level: VP preconditioning around the DDPM network."""
        if cond is not None or x_self_cond is not None or dx is not None:
            raise NotImplementedError("cond / x_self_cond / dx are outside the built path")
        net = self._net(model)
        t = torch.as_tensor(t).reshape(-1)
        if t.numel() != 1:
            raise NotImplementedError("one noise level for the whole batch (what sample_edm evaluates)")
        sigma = t.to(torch.float32)
        c_noise = self.num_timesteps - 1 - self.round_sigma(sigma.reshape(1, 1, 1, 1), return_index=True).to(torch.float32)
        with torch.no_grad():
            return net.plan.denoise(net.packed_weights(), xt.to(torch.float32).contiguous(), float(sigma), float(c_noise),
                                    ws=net._ws, want_F=True)

    # ---- sampling -----------------------------------------------------------------------------------------------
    def sample_edm(self, h, u, sparams, return_last=True, guide_dx=False):
        """models/ddim.py:959-1051.  h, u: 'b h w c' normalised fields; returns [b, t, h, w, c] float64.
        The known region is rows < n_time_h of h and rows < n_time_u of u; every step runs n_repeat Heun updates with the
        known region re-noised to the current level in between (RePaint)."""
        if guide_dx:
            raise NotImplementedError("guide_dx=True (PDE guidance) is outside the built path")
        if self.edm_steps is None:
            self.set_test_sampler_params(sparams)
        net = self._net(self.ema_model if self.ema_model is not None else self.model)
        hu = _nchw(torch.cat([h, u], dim=-1)).float()
        rd, keep = _lib.repaint_desc(sparams, self.edm_steps, self._alphas_ext(), self.h_ch, self.u_ch)
        hu_noise = torch.randn_like(hu)
        N, R = rd.timesteps, rd.n_repeat
        t = _lib.repaint_schedule(rd)
        # the reference draws a per-step tensor every step; it only matters where round_sigma(t_cur + gamma t_cur) > t_cur
        churn = float(sparams.S_churn) > 0
        step_noise = torch.randn((N,) + tuple(hu.shape), dtype=torch.float64, device=hu.device) if churn else None
        repeat_noise = torch.randn((N, R - 1) + tuple(hu.shape), dtype=torch.float64, device=hu.device) if R > 1 else None
        with torch.no_grad():
            return net.plan.repaint_sample(net.packed_weights(), rd, hu, hu_noise, step_noise, repeat_noise,
                                           return_last=return_last, ws=self._sample_ws)

    def sample(self, *a, **k):
        raise NotImplementedError("the DDIM sampler (models/ddim.py:706-806) is not built; use sample_edm")

    def sample_with_repeat(self, *a, **k):
        raise NotImplementedError("the DDIM RePaint sampler (models/ddim.py:808-913) is not built; use sample_edm")

    def training_step(self, *a, **k):
        raise NotImplementedError("DDPM (epsilon-prediction) training is not built: SURVEY.md section 8 f1 covers EDM sampling "
                                  "of a trained DDPM checkpoint")
