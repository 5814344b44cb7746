"""GPU: SURVEY.md section 8 f1 -- the DDPM U-Net (models/ddim_blocks.py Model) and PlDdim's RePaint-style EDM sampler
(models/ddim.py:915-1051) through the C ABI and through the drop-in classes, against the reference's own outputs
(tests/golden/ddpm.npz, written by oracle/make_golden_ddpm.py) and against the oracle."""
import numpy as np
import pytest
import torch

from oracle import ddpm_oracle as dorc
from oracle import fixtures as fx
from tests.test_hip_module import wrap

from tests._tol import close_per_entry

pytestmark = pytest.mark.gpu
CFG = fx.CFG_D


@pytest.fixture(scope="module")
def net():
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    assert torch.cuda.is_available()
    plan = L.DdpmPlan(CFG.in_channels, CFG.out_ch, CFG.ch, CFG.ch_mult, CFG.num_res_blocks, CFG.attn_resolutions, CFG.resolution)
    P = dorc.make_params(CFG, 21)
    assert plan.param_names == [n for n, _ in dorc.param_shapes(CFG)]
    packed = plan.pack({k: v.cuda() for k, v in P.items()}, dorc.timestep_freqs(CFG.ch).cuda())
    return L, plan, packed, P


def close(got, ref, rtol=1e-4, atol=1e-5, what=""):
    got, ref = torch.as_tensor(got).detach().cpu(), torch.as_tensor(ref)
    assert got.shape == ref.shape and got.dtype == ref.dtype, (what, got.shape, ref.shape, got.dtype, ref.dtype)
    err = float((got.double() - ref.double()).abs().max())
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol, msg=lambda m: f"{what} (max|d| {err:.3e}): {m}")


def test_model_forward_golden(net, golden):
    L, plan, packed, P = net
    g = golden("ddpm.npz")
    x = fx.randn("ddpm/x", 3, 2, CFG.resolution, CFG.resolution).cuda()
    close(plan.forward(packed, x, float(fx.DDPM_T)), g["F_t937"], what="Model.forward t=937")
    close(plan.forward(packed, x, 0.0), g["F_t0"], what="Model.forward t=0")
    for i, t in enumerate((3.0, 500.0, 999.0)):          # the reference's per-sample timesteps, one call per level
        close(plan.forward(packed, x, t)[i], g["F_tB"][i], what=f"Model.forward t={t}")
    with pytest.raises(RuntimeError, match="resolution"):
        plan.forward(packed, torch.zeros(1, 2, 16, 16, device="cuda"), 1.0)


def test_get_denoised_golden(net, golden):
    L, plan, packed, P = net
    g = golden("ddpm.npz")
    x = fx.randn("ddpm/x", 3, 2, CFG.resolution, CFG.resolution)
    for i, s in enumerate(fx.DDPM_SIGMAS):
        ref = torch.as_tensor(g[f"D_sigma{i}"])
        D = plan.denoise(packed, (x * (1 + s)).cuda(), s, float(g[f"cnoise_sigma{i}"][0]))
        close(D, ref, atol=1e-5 * float(ref.abs().max()), what=f"get_denoised sigma={s}")


def repaint_desc(L, tag):
    N, R, churn, nth, ntu = fx.REPAINT_CASES[tag]
    sp = dorc.RepaintParams(timesteps=N, n_repeat=R, S_churn=churn, n_time_h=nth, n_time_u=ntu)
    betas = dorc.betas_of(CFG)
    return L.repaint_desc(sp, dorc.edm_steps_of(betas), dorc.alphas_ext_of(betas), 1, 1)


@pytest.mark.parametrize("tag", list(fx.REPAINT_CASES))
def test_repaint_sample_golden(net, golden, tag):
    """mcedm_repaint_sample: n_repeat inner Heun updates per step, known region re-noised in between."""
    L, plan, packed, P = net
    g = golden("ddpm.npz")
    N, R, churn, nth, ntu = fx.REPAINT_CASES[tag]
    h, u, init, steps, reps = fx.repaint_inputs(tag)
    hu = torch.cat([h, u], dim=-1).permute(0, 3, 1, 2).contiguous()
    rd, keep = repaint_desc(L, tag)
    step_noise = torch.stack(steps).cuda() if churn > 0 else None
    repeat_noise = torch.stack([torch.stack(r) for r in reps]).cuda()
    xs = plan.repaint_sample(packed, rd, hu.cuda(), init.cuda(), step_noise, repeat_noise, return_last=False)
    ref = torch.as_tensor(g[f"{tag}_xs"])
    scale = float(ref.abs().max())
    print(f"repaint {tag}: max|d| = {float((xs.cpu() - ref).abs().max()):.3e} on max|x| = {scale:.1f}")
    close_per_entry(xs, ref, what=f"repaint {tag}")        # each step's state against its own magnitude
    last = plan.repaint_sample(packed, rd, hu.cuda(), init.cuda(), step_noise, repeat_noise, return_last=True)
    assert tuple(last.shape) == (fx.REPAINT_B, 1, CFG.resolution, CFG.resolution, 2) and torch.equal(last[:, 0], xs[:, -1])
    known = torch.ones(fx.REPAINT_B, CFG.resolution, CFG.resolution, 2, dtype=torch.bool)
    known[:, nth:, :, 0] = False
    known[:, ntu:, :, 1] = False
    assert torch.equal(last[:, 0].cpu()[known], torch.cat([h, u], dim=-1).double()[known]), "known entries must be the clean data"
    if R > 1:
        with pytest.raises(RuntimeError, match="repeat_noise"):
            plan.repaint_sample(packed, rd, hu.cuda(), init.cuda(), step_noise, None)


def hparams(sampler):
    return wrap(dict(
        name="ddim",
        model=dict(type="simple", in_channels=CFG.in_channels, cond_channels=0, cat_cond=False, out_ch=CFG.out_ch, ch=CFG.ch,
                   ch_mult=list(CFG.ch_mult), num_res_blocks=CFG.num_res_blocks, attn_resolutions=list(CFG.attn_resolutions),
                   dropout=0.0, var_type="fixedsmall", ema_rate=0.999, ema=True, resamp_with_conv=True,
                   resolution=CFG.resolution, self_cond=True, dx_cond=False, cat_dx=False, dx_norm="l2", dx_detach=False),
        data=dict(normalization="gauss", uniform_dequantization=False, gaussian_dequantization=False, rescaled=False),
        diffusion=dict(beta_schedule="linear", beta_start=CFG.beta_start, beta_end=CFG.beta_end,
                       num_diffusion_timesteps=CFG.num_timesteps),
        optimization=dict(optimizer="Adam", lr=0.0002, weight_decay=0.0, beta1=0.9, amsgrad=False, eps=1e-8),
        sampler=sampler))


def test_plddim_module_golden(golden, monkeypatch):
    """The drop-in PlDdim: state_dict layout, schedule tables, get_denoised and sample_edm by their reference names."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd.ddim import PlDdim
    g = golden("ddpm.npz")
    tag = "churn_r3"
    N, R, churn, nth, ntu = fx.REPAINT_CASES[tag]
    sp = wrap(dict(name="edm", type="edm", timesteps=N, sigma_min=0.002, sigma_max=80, rho=7, S_churn=churn, S_min=0, S_max="inf",
                   S_noise=1, n_samples=1, n_repeat=R, n_time_h=nth, n_time_u=ntu, return_last=True, guide_dx=False, w=0.0))
    m = PlDdim(hparams(sp)).cuda()
    P = dorc.make_params(CFG, 21)
    sd = m.state_dict()
    for n, s in dorc.param_shapes(CFG):
        assert tuple(sd[f"model.{n}"].shape) == tuple(s) and f"ema_model.ma_model.{n}" in sd
    assert "betas" in sd and "logvar" in sd
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    m.set_test_sampler_params(sp)
    m.noise_source = "torch"            # per-step / per-loop noise as torch tensors, so the reference's draws can be injected
    # the schedule tables are CPU tensor arithmetic of the host (like the reference's): same expression, last-bit
    # differences between hosts are possible, so they are compared to 2 ulp, not bitwise
    es, ae = torch.as_tensor(g["edm_steps"]), torch.as_tensor(g["alphas_ext"])
    print(f"edm_steps max rel diff vs golden: {float(((m.edm_steps - es).abs() / es).max()):.2e}")
    torch.testing.assert_close(m.edm_steps, es, rtol=3e-7, atol=0)
    torch.testing.assert_close(m.compute_alpha(torch.tensor([0, 80, 999])).flatten(), ae[[1, 81, 1000]], rtol=3e-7, atol=0)
    x = fx.randn("ddpm/x", 3, 2, CFG.resolution, CFG.resolution)
    s = fx.DDPM_SIGMAS[1]
    D, F = m.get_denoised(m.ema_model, (x * (1 + s)).double().cuda(), torch.tensor(s, dtype=torch.float64), w=0.0)
    ref = torch.as_tensor(g["D_sigma1"])
    close(D, ref, atol=1e-5 * float(ref.abs().max()), what="PlDdim.get_denoised")
    h, u, init, steps, reps = fx.repaint_inputs(tag)
    real_randn = torch.randn
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: init.to(t.device))

    def fake_randn(*a, **k):
        shape = a[0] if len(a) == 1 and isinstance(a[0], (tuple, list)) else a
        if k.get("dtype") == torch.float64 and len(shape) == 5:
            return torch.stack(steps).cuda()
        if k.get("dtype") == torch.float64 and len(shape) == 6:
            return torch.stack([torch.stack(r) for r in reps]).cuda()
        return real_randn(*a, **k)
    monkeypatch.setattr(torch, "randn", fake_randn)
    xs = m.sample_edm(h.cuda(), u.cuda(), sp, return_last=False)
    monkeypatch.undo()
    ref = torch.as_tensor(g[f"{tag}_xs"])
    close_per_entry(xs, ref, what="PlDdim.sample_edm")
    with pytest.raises(NotImplementedError):
        m.training_step(None, 0)


def test_device_normal_generator_statistics(net):
    """mcedm_normal_fill (Philox4x32-10 + Box-Muller, csrc/edm.hip): moments of N(0, 1), reproducible per (seed, draw),
    different across draws and seeds, no correlation between the two values of a counter or between neighbouring draws."""
    L = net[0]
    n = 1 << 22
    seed = torch.tensor([1234567], dtype=torch.int64, device="cuda")
    a = L.normal_fill(torch.empty(n, dtype=torch.float64, device="cuda"), seed, 0)
    b = L.normal_fill(torch.empty(n, dtype=torch.float64, device="cuda"), seed, 1)
    assert torch.equal(a, L.normal_fill(torch.empty(n, dtype=torch.float64, device="cuda"), seed, 0))
    c = L.normal_fill(torch.empty(n, dtype=torch.float64, device="cuda"), seed + 1, 0)
    assert not torch.equal(a, b) and not torch.equal(a, c) and bool(torch.isfinite(a).all())
    se = 1.0 / n ** 0.5
    for x in (a, b, c):
        m, v = float(x.mean()), float(x.var())
        skew, kurt = float((x ** 3).mean()), float((x ** 4).mean())
        assert abs(m) < 5 * se and abs(v - 1) < 5 * se * 2 ** 0.5 and abs(skew) < 5 * se * 15 ** 0.5 and abs(kurt - 3) < 5 * se * 96 ** 0.5, (m, v, skew, kurt)
        assert float(x.abs().max()) > 4.5                      # the tails are there (P(|z| > 4.5) * n ~ 28)
    corr = lambda x, y: float((x * y).mean())
    assert abs(corr(a[0::2], a[1::2])) < 5 * (2.0 / n) ** 0.5 and abs(corr(a, b)) < 5 * se and abs(corr(a, c)) < 5 * se
    # an odd length ends on the first value of a pair
    odd = L.normal_fill(torch.empty(7, dtype=torch.float64, device="cuda"), seed, 0)
    assert torch.equal(odd, a[:7])


def test_repaint_device_noise_matches_materialised_draws(net):
    """mcedm_repaint_sample_rng == mcedm_repaint_sample fed with the tensors mcedm_normal_fill produces for the same seed
    (draw index i * n_repeat for the step noise, i * n_repeat + 1 + k between inner loops): the in-kernel generator is the
    documented stream, and the noise-free plumbing is the golden-tested one."""
    L, plan, packed, P = net
    tag = "churn_r3"
    N, R, churn, nth, ntu = fx.REPAINT_CASES[tag]
    h, u, init, _, _ = fx.repaint_inputs(tag)
    hu = torch.cat([h, u], dim=-1).permute(0, 3, 1, 2).contiguous().cuda()
    rd, keep = repaint_desc(L, tag)
    seed = torch.tensor([99], dtype=torch.int64, device="cuda")
    shape = (fx.REPAINT_B, 2, CFG.resolution, CFG.resolution)
    steps = torch.stack([L.normal_fill(torch.empty(shape, dtype=torch.float64, device="cuda"), seed, i * R) for i in range(N)])
    reps = torch.stack([torch.stack([L.normal_fill(torch.empty(shape, dtype=torch.float64, device="cuda"), seed, i * R + 1 + k)
                                     for k in range(R - 1)]) for i in range(N)])
    want = plan.repaint_sample(packed, rd, hu, init.cuda(), steps, reps, return_last=False)
    got = plan.repaint_sample(packed, rd, hu, init.cuda(), return_last=False, rng_seed=seed)
    assert torch.equal(got, want)
    other = plan.repaint_sample(packed, rd, hu, init.cuda(), return_last=False, rng_seed=seed + 1)
    assert not torch.equal(other, want)


def test_plddim_sample_edm_device_noise_graph_and_seed(monkeypatch):
    """PlDdim.sample_edm in its default mode: noise generated on the device from a seed drawn from torch's generator, the
    whole call replayed from one HIP graph.  Same torch seed -> same states; graph == eager launches; known rows exact."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    from mcedm_amd.ddim import PlDdim
    N, R, nth, ntu = 4, 3, 0, 16
    sp = wrap(dict(name="edm", type="edm", timesteps=N, sigma_min=0.002, sigma_max=80, rho=7, S_churn=15.0, S_min=0, S_max="inf",
                   S_noise=1, n_samples=1, n_repeat=R, n_time_h=nth, n_time_u=ntu, return_last=True, guide_dx=False, w=0.0))
    m = PlDdim(hparams(sp)).cuda()
    P = dorc.make_params(CFG, 21)
    with torch.no_grad():
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    m.set_test_sampler_params(sp)
    h, u = fx.randn("ddpm/dev/h", 3, 32, 32, 1).cuda(), fx.randn("ddpm/dev/u", 3, 32, 32, 1).cuda()

    def run(seed):
        torch.manual_seed(seed)
        return m.sample_edm(h, u, sp, return_last=True)
    a, b, c = run(5), run(5), run(6)
    assert isinstance(next(iter(m._graphs.values())), L.GraphedRepaint) and len(m._graphs) == 1
    assert torch.equal(a, b) and not torch.equal(a, c) and bool(torch.isfinite(a).all())
    assert torch.equal(a[:, 0, :ntu, :, 1], u[:, :ntu, :, 0].double()), "known rows of u are the clean data"
    monkeypatch.setenv("MCEDM_HIP_GRAPH", "0")
    assert torch.equal(run(5), a), "graph replay and eager launches must agree bit for bit"


def test_model_forward_with_self_conditioning_golden(net, golden):
    """Model.forward(x, t, x_self_cond) (ddim_blocks.py:417-420): the self-conditioning half of conv_in's input."""
    L, plan, packed, P = net
    g = golden("ddpm.npz")
    x = fx.randn("ddpm/x", 3, 2, CFG.resolution, CFG.resolution).cuda()
    xsc = fx.randn("ddpm/x_self_cond", 3, 2, CFG.resolution, CFG.resolution).cuda()
    close(plan.forward(packed, x, float(fx.DDPM_T), x_self_cond=xsc), g["F_selfcond"], what="Model.forward x_self_cond")
    assert not torch.equal(plan.forward(packed, x, float(fx.DDPM_T), x_self_cond=xsc), plan.forward(packed, x, float(fx.DDPM_T)))


@pytest.mark.parametrize("tag", list(fx.DDIM_CASES))
def test_ddim_repaint_sample_golden(net, golden, tag):
    """mcedm_ddim_repaint_sample = PlDdim.sample_with_repeat (models/ddim.py:808-913): uniform / quad skipping, eta = 0 and
    eta != 0 (the reference's uniform rand_like draw), x0 prediction fed back as x_self_cond; against the reference's outputs."""
    L, plan, packed, P = net
    g = golden("ddpm.npz")
    N, skip, eta, R, nth, ntu = fx.DDIM_CASES[tag]
    h, u, init, etas = fx.ddim_inputs(tag)
    hu = torch.cat([h, u], dim=-1).permute(0, 3, 1, 2).contiguous()
    sp = dorc.DdimParams(timesteps=N, skip_type=skip, eta=eta, n_repeat=R, n_time_h=nth, n_time_u=ntu)
    dd, keep = L.ddim_desc(sp, dorc.alphas_ext_of(dorc.betas_of(CFG)), 1, 1, True)
    nseq = len(dorc.ddim_sequence(CFG.num_timesteps, sp))
    eta_noise = torch.stack(etas[:nseq]).cuda() if abs(eta) > 1e-10 else None
    xs, x0 = plan.ddim_repaint_sample(packed, dd, hu.cuda(), init.cuda(), eta_noise, return_last=False)
    for got, key in ((xs, "xs"), (x0, "x0")):
        ref = torch.as_tensor(g[f"ddim_{tag}_{key}"])
        print(f"ddim {tag} {key}: max|d| = {float((got.cpu() - ref).abs().max()):.3e} on max|x| = {float(ref.abs().max()):.1f}")
        close_per_entry(got, ref, what=f"ddim {tag} {key}")
    last_xs, last_x0 = plan.ddim_repaint_sample(packed, dd, hu.cuda(), init.cuda(), eta_noise, return_last=True)
    assert torch.equal(last_xs[:, 0], xs[:, -1]) and torch.equal(last_x0[:, 0], x0[:, -1])
    # known rows of every x0 prediction are the clean data (ddim.py:878-879)
    known = torch.ones(fx.REPAINT_B, CFG.resolution, CFG.resolution, 2, dtype=torch.bool)
    known[:, nth:, :, 0] = False
    known[:, ntu:, :, 1] = False
    assert torch.equal(x0[:, -1].cpu()[known], torch.cat([h, u], dim=-1)[known])
    if abs(eta) > 1e-10:
        with pytest.raises(RuntimeError, match="eta_noise"):
            plan.ddim_repaint_sample(packed, dd, hu.cuda(), init.cuda(), None)
