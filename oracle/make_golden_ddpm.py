"""Golden vectors for the RePaint-style EDM sampler on the DDPM U-Net, made by RUNNING THE
REFERENCE (build container only): models/ddim.py PlDdim (get_denoised :915-947, round_sigma :949-957, compute_alpha
:700-704, sample_edm :959-1051) with models/ddim_blocks.py Model (:222-470), configs/model/ddim_res32.yaml at
resolution 32.  oracle/ddpm_oracle.py is cross-checked on every case before anything is written.

    python oracle/make_golden_ddpm.py        # rewrites tests/golden/ddpm.npz
"""
import make_golden as mg            # reference import path, Lightning stand-in, helpers

import torch

from models.ddim import PlDdim      # reference
from oracle import ddpm_oracle as dorc
from oracle import fixtures as fx


def hparams(cfg: dorc.DdpmConfig, sampler: dict):
    """configs/model/ddim_res32.yaml as an attribute dict."""
    return mg._wrap(dict(
        name="ddim",
        model=dict(type="simple", in_channels=cfg.in_channels, cond_channels=0, cat_cond=False, out_ch=cfg.out_ch, ch=cfg.ch,
                   ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks, attn_resolutions=list(cfg.attn_resolutions),
                   dropout=0.0, var_type="fixedsmall", ema_rate=0.999, ema=True, resamp_with_conv=True,
                   resolution=cfg.resolution, self_cond=cfg.self_cond, dx_cond=False, cat_dx=False, dx_norm="l2", dx_detach=False, node_type=False),
        data=dict(normalization="gauss", uniform_dequantization=False, gaussian_dequantization=False, rescaled=False),
        diffusion=dict(beta_schedule="linear", beta_start=cfg.beta_start, beta_end=cfg.beta_end,
                       num_diffusion_timesteps=cfg.num_timesteps),
        optimization=dict(optimizer="Adam", lr=0.0002, weight_decay=0.0, beta1=0.9, amsgrad=False, eps=1e-8, grad_clip=1.0,
                          loss="l2", pde_loss_lambda=0.0, pde_loss_prop_t=False, use_gt_pde=False, factor=0.3, step_size=50),
        sampler=sampler))


def sampler_dict(**over):
    d = mg.sampler_dict(timesteps=18, S_churn=0.0, n_repeat=2, n_time_h=0, n_time_u=64)
    d.update(over)
    return d


def build(cfg, seed, sampler):
    m = PlDdim(hparams(cfg, sampler))
    named = [(n, tuple(p.shape)) for n, p in m.model.named_parameters()]
    assert named == [(n, tuple(s)) for n, s in dorc.param_shapes(cfg)], "param_shapes drifted from the reference Model"
    P = dorc.make_params(cfg, seed)
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    m.set_test_sampler_params(mg._wrap(sampler))       # builds edm_steps / sigma_min / sigma_max (ddim.py:122-129)
    m.h_ch, m.u_ch = 1, 1
    return m, P


def main():
    cfg = fx.CFG_D
    out = {}
    m, P = build(cfg, 21, sampler_dict())
    steps = dorc.edm_steps_of(dorc.betas_of(cfg))
    assert torch.equal(m.betas, dorc.betas_of(cfg)) and torch.equal(m.edm_steps, steps)
    assert torch.equal(m.compute_alpha(torch.tensor([0, 80, 999])).flatten(), dorc.alphas_ext_of(m.betas)[[1, 81, 1000]])
    out["edm_steps"], out["alphas_ext"] = steps, dorc.alphas_ext_of(m.betas)
    B, S = 3, cfg.resolution
    x = fx.randn("ddpm/x", B, 2, S, S)
    with torch.no_grad():
        for name, t in (("t937", fx.DDPM_T), ("t0", torch.tensor([0.0])), ("tB", torch.tensor([3.0, 500.0, 999.0]))):
            y = m.model(x, t)
            mg.check(f"Model.forward {name}", dorc.model_forward(P, cfg, x, t), y, rtol=1e-4, atol=1e-5)
            out[f"F_{name}"] = y
        for i, s in enumerate(fx.DDPM_SIGMAS):
            sig = torch.tensor(s, dtype=torch.float64)
            D, Fx = m.get_denoised(m.ema_model, (x * (1 + s)).double(), sig, w=0.0)
            Do, Fo = dorc.get_denoised(P, cfg, steps, (x * (1 + s)).double(), sig)
            mg.check(f"get_denoised sigma={s}", Do, D, rtol=1e-4, atol=1e-5)
            out[f"D_sigma{i}"] = D
            out[f"cnoise_sigma{i}"] = cfg.num_timesteps - 1 - m.round_sigma(sig.float().reshape(1, 1, 1, 1), return_index=True).float().flatten()
    for tag, (N, R, churn, nth, ntu) in fx.REPAINT_CASES.items():
        sp = sampler_dict(timesteps=N, n_repeat=R, S_churn=churn, n_time_h=nth, n_time_u=ntu)
        m, P = build(cfg, 21, sp)
        h, u, init, stp, reps = fx.repaint_inputs(tag)
        queue = [init]
        for i in range(N):
            queue += [stp[i]] + reps[i]
        with torch.no_grad(), mg._Inject(queue) as inj:
            xs = m.sample_edm(h, u, mg._wrap(sp), return_last=False)
        assert not inj.like_queue and xs.dtype == torch.float64 and tuple(xs.shape) == (fx.REPAINT_B, N + 1, S, S, 2)
        hu = torch.cat([h, u], dim=-1).permute(0, 3, 1, 2)
        spo = dorc.RepaintParams(timesteps=N, n_repeat=R, S_churn=churn, n_time_h=nth, n_time_u=ntu)
        with torch.no_grad():
            xo = dorc.sample_edm_repaint(P, cfg, hu, spo, init, stp, reps, return_last=False)
        mg.check(f"PlDdim.sample_edm {tag}", xo, xs, rtol=1e-4, atol=1e-5)
        # the known region of the final state is the clean data (ddim.py:1041-1043)
        known = torch.ones(fx.REPAINT_B, S, S, 2, dtype=torch.bool)
        known[:, nth:, :, 0] = False
        known[:, ntu:, :, 1] = False
        assert torch.equal(xs[:, -1][known], torch.cat([h, u], dim=-1).double()[known])
        out[f"{tag}_xs"] = xs
    # Model.forward with a self-conditioning tensor (ddim_blocks.py:417-420)
    m, P = build(cfg, 21, sampler_dict())
    xsc = fx.randn("ddpm/x_self_cond", B, 2, S, S)
    with torch.no_grad():
        y = m.model(x, fx.DDPM_T, x_self_cond=xsc)
        mg.check("Model.forward x_self_cond", dorc.model_forward(P, cfg, x, fx.DDPM_T, x_self_cond=xsc), y, rtol=1e-4, atol=1e-5)
    out["F_selfcond"] = y
    # the DDIM sampler with RePaint loops (models/ddim.py:808-913): uniform / quad skipping, eta = 0 and eta != 0 (the
    # reference's torch.rand_like draw injected), the x0 prediction fed back as x_self_cond
    for tag, (N, skip, eta, R, nth, ntu) in fx.DDIM_CASES.items():
        sp = sampler_dict(type="ddim", skip_type=skip, eta=eta, timesteps=N, n_repeat=R, n_time_h=nth, n_time_u=ntu)
        m, P = build(cfg, 21, sp)
        h, u, init, etas = fx.ddim_inputs(tag)
        spo = dorc.DdimParams(timesteps=N, skip_type=skip, eta=eta, n_repeat=R, n_time_h=nth, n_time_u=ntu)
        nseq = len(dorc.ddim_sequence(cfg.num_timesteps, spo))
        queue = list(etas)
        real_rand_like = torch.rand_like
        torch.rand_like = lambda t_, **k: queue.pop(0).to(t_.dtype)
        try:
            with torch.no_grad(), mg._Inject([init]) as inj:
                xs, x0s = m.sample_with_repeat(h, u, mg._wrap(sp), return_last=False)
        finally:
            torch.rand_like = real_rand_like
        assert not inj.like_queue and xs.dtype == torch.float32 and tuple(xs.shape) == (fx.REPAINT_B, nseq + 1, S, S, 2)
        assert tuple(x0s.shape) == (fx.REPAINT_B, nseq, S, S, 2) and len(queue) == len(etas) - (nseq if abs(eta) > 1e-10 else 0)
        hu = torch.cat([h, u], dim=-1).permute(0, 3, 1, 2)
        with torch.no_grad():
            xo, x0o = dorc.sample_with_repeat(P, cfg, hu, spo, init, etas, return_last=False)
        scale = float(xs.abs().max())
        mg.check(f"PlDdim.sample_with_repeat {tag} xs", xo, xs, rtol=1e-4, atol=1e-5 * scale)
        mg.check(f"PlDdim.sample_with_repeat {tag} x0", x0o, x0s, rtol=1e-4, atol=1e-5 * float(x0s.abs().max()))
        out[f"ddim_{tag}_xs"], out[f"ddim_{tag}_x0"] = xs, x0s
    mg.save("ddpm.npz", seed=21, **out)


if __name__ == "__main__":
    main()
