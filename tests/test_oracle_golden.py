"""CPU: the oracle (oracle/mcedm_oracle.py) reproduces every committed golden vector.

The golden vectors are outputs of the reference itself (oracle/make_golden.py); inputs and
parameters are rebuilt from the tags in oracle/fixtures.py.  Tolerances: the oracle restates the
reference with the same torch CPU primitives, so agreement is expected at rounding level
(rtol 1e-5); the HIP-vs-oracle bar (rtol 1e-4 / atol 1e-5, north_star) lives in the -m gpu tests.
"""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc

torch.set_num_threads(4)


def close(got, ref, rtol=1e-5, atol=1e-6):
    ref = torch.as_tensor(ref)
    assert got.shape == ref.shape and got.dtype == ref.dtype
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize("C", [64, 128, 256])
def test_group_norm(golden, C):
    g = golden("ops.npz")
    x = fx.randn(f"ops/gn{C}/x", 2, C, 8, 8) * 1.5 + 0.3
    y = orc.group_norm(x, fx.param(f"ops/gn{C}", "norm.weight", (C,)), fx.param(f"ops/gn{C}", "norm.bias", (C,)))
    close(y, g[f"gn{C}_y"])


@pytest.mark.parametrize("tag", list(fx.CONV_CASES))
def test_conv2d(golden, tag):
    g = golden("ops.npz")
    kw = fx.CONV_CASES[tag]
    cin, cout = fx.conv_channels(tag)
    w = b = None
    if kw["kernel"]:
        w = fx.param(f"ops/conv_{tag}", "conv.weight", (cout, cin, kw["kernel"], kw["kernel"]))
        b = fx.param(f"ops/conv_{tag}", "conv.bias", (cout,))
    x = fx.randn(f"ops/conv_{tag}/x", 2, cin, 8, 12)
    close(orc.conv2d(x, w, b, up=kw.get("up", False), down=kw.get("down", False)), g[f"conv_{tag}_y"])


@pytest.mark.parametrize("T", list(fx.ATTN_CASES))
def test_attention_fwd_bwd(golden, T):
    g = golden("ops.npz")
    B, hw = fx.ATTN_CASES[T]
    qkv = fx.randn(f"ops/attn{T}/qkv", B, 384, *hw).requires_grad_(True)
    a = orc.attention(qkv, 2)
    close(a.detach(), g[f"attn{T}_a"])
    (dqkv,) = torch.autograd.grad(a, qkv, fx.randn(f"ops/attn{T}/da", *a.shape))
    close(dqkv, g[f"attn{T}_dqkv"], rtol=1e-4, atol=2e-6)


def test_positional_embedding(golden):
    close(orc.positional_embedding(fx.PE_LABELS, 64), golden("ops.npz")["pe_y"])


@pytest.mark.parametrize("tag", list(fx.BLOCK_CASES))
@pytest.mark.parametrize("n_emb", [1, 2])
def test_unet_block(golden, tag, n_emb):
    x, emb = fx.block_inputs(tag, n_emb)
    y = orc.unet_block(fx.block_params(tag), fx.block_spec(tag), x, emb)
    close(y, golden("blocks.npz")[f"{tag}_n{n_emb}_y"], rtol=1e-4, atol=1e-5)


def test_param_layout_matches_reference_counts():
    # 1,587,010 parameters (ch=64); 6,120,834 (ch=128, 3 levels, attn_resolutions=[16] so
    # only dec.32x32_in0 attends -- SURVEY.md §8d)
    assert sum(int(np.prod(s)) for _, s in orc.param_shapes(fx.CFG_P)) == 1_587_010
    cfg128 = orc.UNetConfig(ch=128, attn_resolutions=(16,))
    assert sum(int(np.prod(s)) for _, s in orc.param_shapes(cfg128)) == 6_120_834
    spec = orc.build_spec(fx.CFG_P)
    assert len(spec.enc) + len(spec.dec) == 15
    assert sum(b.attn for b in spec.enc + spec.dec) == 4


def test_unet_forward_and_precond(golden):
    g = golden("unet_P.npz")
    P = orc.make_params(fx.CFG_P, int(g["seed"]))
    x = fx.randn("unet_P/x", 4, 2, 32, 32)
    cond = fx.randn("unet_P/cond", 4, 2, 32, 32)
    with torch.no_grad():
        for tag, labels in fx.UNET_LABELS.items():
            close(orc.unet_forward(P, fx.CFG_P, x, labels, cond), g[f"F_{tag}"], rtol=1e-4, atol=1e-5)
        close(orc.unet_forward(P, fx.CFG_P, x, torch.tensor([0.3]), None), g["F_nocond"], rtol=1e-4, atol=1e-5)
        for i, s in enumerate(fx.PRECOND_SIGMAS):
            close(orc.model_precond(P, fx.CFG_P, x * (1 + s), torch.tensor(s), cond), g[f"D_sigma{i}"],
                  rtol=1e-4, atol=1e-5)
        close(orc.model_precond(P, fx.CFG_P, x, fx.PRECOND_SIGMA_B, cond), g["D_sigmaB"], rtol=1e-4, atol=1e-5)
        D, _ = orc.get_denoised(P, fx.CFG_P, x.double(), torch.tensor(0.7).double(), cond, w=0.5)
        close(D, g["D_cfg_w05"], rtol=1e-4, atol=1e-5)


def test_unet_forward_wide(golden):
    g = golden("unet_W.npz")
    P = orc.make_params(fx.CFG_W, int(g["seed"]))
    with torch.no_grad():
        y = orc.unet_forward(P, fx.CFG_W, fx.randn("unet_W/x", 2, 2, 16, 16), fx.UNET_W_LABELS,
                             fx.randn("unet_W/cond", 2, 2, 16, 16))
    close(y, g["F"], rtol=1e-4, atol=1e-5)


def test_t_steps(golden):
    t = orc.edm_t_steps(18, 0.002, 80, 7)
    assert t.dtype == torch.float64 and t.shape == (19,) and t[0] == 80.0 and t[-1] == 0.0
    np.testing.assert_array_equal(t.numpy(), golden("sampler_P.npz")["t_steps"])


@pytest.mark.parametrize("tag", ["det_u", "churn_u"])
def test_sample_edm(golden, tag):
    g = golden("sampler_P.npz")
    P = orc.make_params(fx.CFG_P, int(g["seed"]))
    cond, m, init, steps = fx.sampler_inputs(tag)
    sp = orc.SamplerParams(S_churn=fx.SAMPLER_CASES[tag][0])
    with torch.no_grad():
        xs = orc.sample_edm(P, fx.CFG_P, cond, m, sp, init, steps, return_last=False)
    assert xs.dtype == torch.float64 and xs.shape == (4, 19, 32, 32, 2)
    close(xs[:, -1:], g[f"{tag}_xs_last"], rtol=1e-3, atol=1e-4)
    close(xs[:, ::6], g[f"{tag}_xs_traj"], rtol=1e-3, atol=1e-4)
    # observed entries stay exactly the conditioning values (mcedm.py:597,618,628)
    obs = (m == 0).permute(0, 2, 3, 1)
    assert torch.equal(xs[:, -1][obs], cond.permute(0, 2, 3, 1).double()[obs])


def test_training_loss_grads_adam(golden):
    g = golden("training_P.npz")
    P = orc.make_params(fx.CFG_P, int(g["seed"]))
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    xc, cond_in, mc = fx.training_nchw(h, u, mask, cond_noise)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss = orc.training_loss(Pg, fx.CFG_P, xc, cond_in, mc, noise, rnd_normal)
    loss.backward()
    close(loss.detach(), g["loss"], rtol=1e-5)
    grads = [Pg[n].grad for n, _ in orc.param_shapes(fx.CFG_P)]
    coef, total = orc.clip_scale(grads, 1.0)
    assert abs(total - float(g["clip_total_norm"])) <= 1e-5 * total
    for n in fx.TRAIN_GRAD_NAMES:
        ref = torch.as_tensor(g[f"grad::{n}"])
        close(Pg[n].grad, ref, rtol=1e-3, atol=2e-6 * float(ref.abs().max()))
        p1, _, _, e1 = orc.adam_ema_step(P[n], ref, torch.zeros_like(ref), torch.zeros_like(ref), P[n], 1, clip=coef)
        close(p1, g[f"adam::{n}"], rtol=1e-5, atol=1e-7)
        close(e1, g[f"ema::{n}"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("tag", list(fx.COND_SAMPLER_CASES))
def test_cond_edm_sampler(golden, tag):
    """Section 8(f2): PlCondEdm.sample_edm (models/ddim.py:1532-1601), unmasked Heun sampler."""
    g = golden("cond_edm.npz")
    P = orc.make_params(fx.CFG_C, int(g["seed"]))
    h, u_noise, steps = fx.cond_sampler_inputs(tag)
    with torch.no_grad():
        xs = orc.sample_edm_cond(P, fx.CFG_C, h.permute(0, 3, 1, 2), orc.SamplerParams(S_churn=fx.COND_SAMPLER_CASES[tag]),
                                 u_noise.permute(0, 3, 1, 2), steps, return_last=False)
    close(xs[:, -1:], g[f"{tag}_xs_last"], rtol=1e-3, atol=1e-4)
    close(xs[:, ::6], g[f"{tag}_xs_traj"], rtol=1e-3, atol=1e-4)


def test_cond_edm_training(golden):
    g = golden("cond_edm.npz")
    P = orc.make_params(fx.CFG_C, int(g["seed"]))
    h, u, noise, rnd_normal = fx.cond_training_inputs()
    st = fx.TRAIN_NORM_STATS
    hn, un = ((h - st[0]) / st[1]).permute(0, 3, 1, 2), ((u - st[2]) / st[3]).permute(0, 3, 1, 2)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss = orc.training_loss_cond(Pg, fx.CFG_C, un, hn, noise, rnd_normal)
    loss.backward()
    close(loss.detach(), g["loss"], rtol=1e-5)
    for n in fx.COND_GRAD_NAMES:
        ref = torch.as_tensor(g[f"grad::{n}"])
        close(Pg[n].grad, ref, rtol=1e-3, atol=2e-6 * float(ref.abs().max()))


# ---- evaluation loops + classifier-free sampler (tests/golden/steps.npz, oracle/make_golden_steps.py) -----------------
@pytest.mark.parametrize("tag", ["swe_n2", "darcy_n16"])
def test_eval_test_step(golden, tag):
    """PlMcedm.test_step (models/mcedm.py:343-441); 'darcy_n16' is BASELINE config 4's path."""
    g = golden("steps.npz")
    c = fx.STEP_CASES[tag]
    P = orc.make_params(fx.CFG_P, int(g["seed"]))
    h, u, masks, noises = fx.step_inputs(tag)
    with torch.no_grad():
        o = orc.eval_test_step(P, fx.CFG_P, h, u, masks, noises, fx.STEP_NORM_STATS, orc.SamplerParams(), c["n_samples"],
                               c["system"], c["down_factor"] if c["down_interp"] else 1)
    ref_keys = sorted(k.split("::", 1)[1] for k in g if k.startswith(tag + "::") and "::log::" not in k)
    assert ref_keys == sorted(k for k in o if not k.startswith("log::"))
    assert ("traj_u" in o) == (c["n_samples"] < 15)
    for k in ref_keys:
        close(o[k], g[f"{tag}::{k}"], rtol=1e-4, atol=1e-5)
    for name in masks:
        close(o[f"log::test_pde_loss_{name}"], g[f"{tag}::log::test_pde_loss_{name}"], rtol=2e-3, atol=1e-6)
    close(o["log::test_pde_loss_gt"], g[f"{tag}::log::test_pde_loss_gt"], rtol=1e-5, atol=1e-6)


def test_eval_validation_step(golden):
    g = golden("steps.npz")
    P = orc.make_params(fx.CFG_P, int(g["seed"]))
    h, u, masks, noises = fx.step_inputs("swe_n2")
    vnoise = {k: (noises[k][0], noises[k][1][:fx.STEP_B]) for k in masks}
    with torch.no_grad():
        o = orc.eval_validation_step(P, fx.CFG_P, h, u, masks, vnoise, fx.STEP_NORM_STATS, orc.SamplerParams(), "swe_per")
    for k in sorted(k[5:] for k in g if k.startswith("val::") and "::log::" not in k):
        close(o[k], g[f"val::{k}"], rtol=1e-4, atol=1e-5)
    for name in masks:
        close(o[f"log::val_pde_loss_{name}"], g[f"val::log::val_pde_loss_{name}"], rtol=2e-3, atol=1e-6)


def test_sample_edm_classifier_free(golden):
    """w = 0.5 through all 35 evaluations (get_denoised's blend, models/mcedm.py:453-458)."""
    g = golden("steps.npz")
    P = orc.make_params(fx.CFG_P, int(g["seed"]))
    cond, m, init, steps = fx.sampler_inputs("det_u")
    with torch.no_grad():
        xs = orc.sample_edm(P, fx.CFG_P, cond, m, orc.SamplerParams(w=fx.CFG_SAMPLER_W), init, steps, return_last=False)
    scale = float(np.abs(g["cfg_u_xs_last"]).max())
    close(xs[:, -1:], g["cfg_u_xs_last"], rtol=1e-4, atol=1e-5 * scale)
    close(xs[:, ::6], g["cfg_u_xs_traj"], rtol=1e-4, atol=1e-5 * scale)


# ---- RePaint-style EDM sampler on the DDPM U-Net (tests/golden/ddpm.npz, oracle/make_golden_ddpm.py) --------------------
def test_ddpm_forward_and_denoised(golden):
    from oracle import ddpm_oracle as dorc
    g = golden("ddpm.npz")
    cfg = fx.CFG_D
    P = dorc.make_params(cfg, int(g["seed"]))
    steps = dorc.edm_steps_of(dorc.betas_of(cfg))
    assert torch.equal(steps, torch.as_tensor(g["edm_steps"]))
    assert torch.equal(dorc.alphas_ext_of(dorc.betas_of(cfg)), torch.as_tensor(g["alphas_ext"]))
    x = fx.randn("ddpm/x", 3, 2, cfg.resolution, cfg.resolution)
    with torch.no_grad():
        close(dorc.model_forward(P, cfg, x, fx.DDPM_T), g["F_t937"], rtol=1e-5, atol=1e-6)
        close(dorc.model_forward(P, cfg, x, torch.tensor([3.0, 500.0, 999.0])), g["F_tB"], rtol=1e-5, atol=1e-6)
        for i, s in enumerate(fx.DDPM_SIGMAS):
            D, _ = dorc.get_denoised(P, cfg, steps, (x * (1 + s)).double(), torch.tensor(s, dtype=torch.float64))
            close(D, g[f"D_sigma{i}"], rtol=1e-5, atol=1e-6 * float(np.abs(g[f"D_sigma{i}"]).max()))


@pytest.mark.parametrize("tag", list(fx.REPAINT_CASES))
def test_ddpm_repaint_sampler(golden, tag):
    from oracle import ddpm_oracle as dorc
    g = golden("ddpm.npz")
    cfg = fx.CFG_D
    P = dorc.make_params(cfg, int(g["seed"]))
    N, R, churn, nth, ntu = fx.REPAINT_CASES[tag]
    h, u, init, stp, reps = fx.repaint_inputs(tag)
    hu = torch.cat([h, u], dim=-1).permute(0, 3, 1, 2)
    sp = dorc.RepaintParams(timesteps=N, n_repeat=R, S_churn=churn, n_time_h=nth, n_time_u=ntu)
    with torch.no_grad():
        xs = dorc.sample_edm_repaint(P, cfg, hu, sp, init, stp, reps, return_last=False)
    close(xs, g[f"{tag}_xs"], rtol=1e-5, atol=1e-6 * float(np.abs(g[f"{tag}_xs"]).max()))


# ---- evaluation loops of models/ddim.py (tests/golden/eval_steps.npz, oracle/make_golden_eval.py) ------------------------
def _check_eval(g, prefix, o):
    ref_keys = sorted(k.split("::", 1)[1] for k in g if k.startswith(prefix + "::") and "::log::" not in k)
    assert ref_keys == sorted(k for k in o if not k.startswith("log::")), (ref_keys, sorted(o))
    for k in ref_keys:
        ref = torch.as_tensor(g[f"{prefix}::{k}"])
        close(o[k], ref, rtol=1e-4, atol=1e-5 * max(1.0, float(ref.abs().max())))
    n_logs = 0
    for k in (k.split("::log::")[1] for k in g if k.startswith(prefix + "::log::")):
        if f"log::{k}" not in o:
            continue
        ref = torch.as_tensor(g[f"{prefix}::log::{k}"])
        torch.testing.assert_close(torch.as_tensor(o[f"log::{k}"]).to(ref.dtype), ref, rtol=2e-3 if "pde" in k else 1e-4, atol=1e-6,
                                   equal_nan=True)
        n_logs += 1
    return n_logs


@pytest.mark.parametrize("tag", list(fx.EVAL_DDPM_CASES) + ["val"])
def test_eval_plddim_steps(golden, tag):
    """PlDdim.test_step / validation_step (models/ddim.py:294-533): how BASELINE config 5 is driven."""
    from oracle import ddpm_oracle as dorc
    g = golden("eval_steps.npz")
    cfg = fx.CFG_D
    P = dorc.make_params(cfg, 21)
    system, n, N, R, churn, nth, ntu = fx.EVAL_DDPM_VAL if tag == "val" else fx.EVAL_DDPM_CASES[tag]
    sp = dorc.RepaintParams(timesteps=N, n_repeat=R, S_churn=churn, n_time_h=nth, n_time_u=ntu)
    h, u, init, steps, reps, u_noise = fx.eval_ddpm_inputs(tag)
    with torch.no_grad():
        if tag == "val":
            o = dorc.eval_validation_step(P, cfg, h, u, fx.EVAL_DDPM_STATS, sp, system, u_noise, init, steps, reps)
        else:
            o = dorc.eval_test_step(P, cfg, h, u, fx.EVAL_DDPM_STATS, sp, n, system, init, steps, reps)
    assert _check_eval(g, f"ddpm_{tag}", o) >= 3
    if tag == "n1_r2":          # n_time_h = 0: the reference's L1 over an empty slice is nan, and it is logged as such
        assert np.isnan(g["ddpm_n1_r2::log::test_h_known"])


@pytest.mark.parametrize("tag", list(fx.EVAL_COND_CASES) + ["val"])
def test_eval_plcondedm_steps(golden, tag):
    """PlCondEdm.test_step / validation_step (models/ddim.py:1154-1319)."""
    g = golden("eval_steps.npz")
    cfg = fx.CFG_C
    P = orc.make_params(cfg, 13)
    h, u, init = fx.eval_cond_inputs(tag)
    with torch.no_grad():
        if tag == "val":
            o = orc.eval_cond_validation_step(P, cfg, h, u, fx.STEP_NORM_STATS, orc.SamplerParams(), "swe_per", init)
        else:
            system, n, guided, st = fx.EVAL_COND_CASES[tag]
            o = orc.eval_cond_test_step(P, cfg, h, u, st, orc.SamplerParams(), n, system, init, guidance=guided)
    assert _check_eval(g, f"cond_{tag}", o) >= 2


# ---- DDIM sampler with RePaint loops (PlDdim.sample_with_repeat, models/ddim.py:808-913) ----------------------------------
@pytest.mark.parametrize("tag", list(fx.DDIM_CASES))
def test_ddim_sample_with_repeat(golden, tag):
    from oracle import ddpm_oracle as dorc
    g = golden("ddpm.npz")
    cfg = fx.CFG_D
    P = dorc.make_params(cfg, int(g["seed"]))
    N, skip, eta, R, nth, ntu = fx.DDIM_CASES[tag]
    h, u, init, etas = fx.ddim_inputs(tag)
    hu = torch.cat([h, u], dim=-1).permute(0, 3, 1, 2)
    sp = dorc.DdimParams(timesteps=N, skip_type=skip, eta=eta, n_repeat=R, n_time_h=nth, n_time_u=ntu)
    with torch.no_grad():
        xs, x0 = dorc.sample_with_repeat(P, cfg, hu, sp, init, etas, return_last=False)
    close(xs, g[f"ddim_{tag}_xs"], rtol=1e-5, atol=1e-6 * float(np.abs(g[f"ddim_{tag}_xs"]).max()))
    close(x0, g[f"ddim_{tag}_x0"], rtol=1e-5, atol=1e-6 * float(np.abs(g[f"ddim_{tag}_x0"]).max()))
    xsc = fx.randn("ddpm/x_self_cond", 3, 2, cfg.resolution, cfg.resolution)
    with torch.no_grad():
        y = dorc.model_forward(P, cfg, fx.randn("ddpm/x", 3, 2, cfg.resolution, cfg.resolution), fx.DDPM_T, x_self_cond=xsc)
    close(y, g["F_selfcond"], rtol=1e-5, atol=1e-6)


def test_eval_plddim_test_step_with_the_ddim_sampler(golden):
    """PlDdim.test_step with sparams.type == 'ddim' (the default diff_sampler: models/ddim.py:393-394)."""
    from oracle import ddpm_oracle as dorc
    g = golden("eval_steps.npz")
    cfg = fx.CFG_D
    P = dorc.make_params(cfg, 21)
    system, n, N, skip, eta, R, nth, ntu = fx.EVAL_DDIM
    st = fx.EVAL_DDPM_STATS
    h, u, init, _ = fx.ddim_inputs("eval", B=n * fx.EVAL_B)
    h, u = h[:fx.EVAL_B] * st[1] + st[0], u[:fx.EVAL_B] * st[3] + st[2]
    with torch.no_grad():
        o = dorc.eval_test_step(P, cfg, h, u, st, dorc.DdimParams(timesteps=N, skip_type=skip, eta=eta, n_repeat=R, n_time_h=nth,
                                                                   n_time_u=ntu), n, system, init)
    assert _check_eval(g, "ddpm_ddim", o) >= 3


@pytest.mark.parametrize("mode", ["cat", "enc"])
def test_dx_cond_network_and_sampler(golden, mode):
    """dx_cond (second clause): the oracle's dx-conditioned network, classifier-free branch and the
    single-task sampler with dx_in = get_dx_input(h, x) against the reference's outputs (oracle/make_golden_dxcond.py)."""
    import dataclasses
    g = golden("dxcond.npz")
    cfg = dataclasses.replace(fx.CFG_C, dx_channels=1, dx_mode=mode)
    P = orc.make_params(cfg, int(g["seed"]))
    x, cond, dx, sig = fx.dxcond_net_inputs()
    with torch.no_grad():
        close(orc.unet_forward(P, cfg, x, sig.log() / 4, cond, dx=dx), g[f"{mode}_F_dx"])
        close(orc.unet_forward(P, cfg, x, sig.log() / 4, cond, dx=None), g[f"{mode}_F_none"])
        D, F = orc.get_denoised(P, cfg, x.double(), sig, cond=cond, w=0.5, dx=dx)
        close(D, g[f"{mode}_D_w"])
        close(F, g[f"{mode}_F_w"])
        if mode == "enc":
            h, u_noise, steps = fx.cond_sampler_inputs("det")
            st = fx.STEP_NORM_STATS
            xs = orc.sample_edm_cond(P, cfg, h.permute(0, 3, 1, 2), orc.SamplerParams(), u_noise.permute(0, 3, 1, 2), steps,
                                     dx_input=lambda hh, d: orc.guidance_dx_cond("swe_per", hh, d, st))
            ref = torch.as_tensor(g["enc_swe_per_xs_last"])
            torch.testing.assert_close(xs, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max()))


@pytest.mark.parametrize("tag", list(fx.COND_IN_CASES))
def test_cond_in_variants(golden, tag):
    """models/mcedm.py:25-34, 241-252 (round 4): add_cond_mask / add_xt widen the conditioning input; get_cond_in, the training
    loss with its gradients and the sampler of the reference, reproduced by the oracle on the widened network."""
    g = golden("cond_in.npz")
    cfg = fx.cond_in_cfg(tag)
    P = orc.make_params(cfg, int(g["seed"]))
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    dx, dt = fx.cond_in_xt()
    xc, cond_in, mc = fx.cond_in_nchw(tag, h, u, mask, cond_noise, dx, dt)
    close(cond_in.permute(0, 2, 3, 1).contiguous(), g[f"{tag}::cond_in"], rtol=0, atol=0)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss = orc.training_loss(Pg, cfg, xc, cond_in, mc, noise, rnd_normal)
    loss.backward()
    close(loss.detach(), g[f"{tag}::loss"], rtol=1e-5, atol=1e-6)
    for n in fx.TRAIN_GRAD_NAMES:
        ref = torch.as_tensor(g[f"{tag}::grad::{n}"])
        close(Pg[n].grad, ref, rtol=1e-3, atol=1e-5 * float(ref.abs().max()))
    init = fx.randn(f"condin/{tag}/init", 4, 2, 32, 32)
    with torch.no_grad():
        xs = orc.sample_edm(P, cfg, cond_in, mc, orc.SamplerParams(timesteps=18), init)
    close(xs, g[f"{tag}::xs_last"], rtol=1e-3, atol=1e-4)


def test_cond_edm_training_with_the_conditioning_dropped(golden):
    """models/ddim.py:1683-1684 with cond_p = 0: the batch trains with cond = None, i.e. zeros (adm_blocks.py:328-331)."""
    g = golden("cond_in.npz")
    cfg = fx.CFG_C
    P = orc.make_params(cfg, 13)
    h, u, noise, rnd_normal = fx.cond_training_inputs()
    st = fx.TRAIN_NORM_STATS
    hn, un = ((h - st[0]) / st[1]).permute(0, 3, 1, 2), ((u - st[2]) / st[3]).permute(0, 3, 1, 2)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss = orc.training_loss_cond(Pg, cfg, un, torch.zeros_like(hn), noise, rnd_normal)
    loss.backward()
    close(loss.detach(), g["cond_drop::loss"], rtol=1e-5, atol=1e-6)
    for n in fx.COND_GRAD_NAMES:
        ref = torch.as_tensor(g[f"cond_drop::grad::{n}"])
        close(Pg[n].grad, ref, rtol=1e-3, atol=1e-5 * float(ref.abs().max()))


def test_cond_edm_node_type_channel(golden):
    """models/ddim.py:36-38, 1105-1114 (round 4): node_type adds a boundary-flag conditioning channel to the single-task model."""
    g = golden("cond_in.npz")
    cfg = fx.CFG_NODE
    P = orc.make_params(cfg, 31)
    h, u, noise, rnd_normal = fx.cond_training_inputs()
    st = fx.TRAIN_NORM_STATS
    hn4 = (h - st[0]) / st[1]
    cond = fx.node_cond(hn4)
    close(cond, g["node::cond_in"], rtol=0, atol=0)
    un = ((u - st[2]) / st[3]).permute(0, 3, 1, 2)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss = orc.training_loss_cond(Pg, cfg, un, cond.permute(0, 3, 1, 2).contiguous(), noise, rnd_normal)
    loss.backward()
    close(loss.detach(), g["node::loss"], rtol=1e-5, atol=1e-6)
    for n in fx.COND_GRAD_NAMES:
        ref = torch.as_tensor(g[f"node::grad::{n}"])
        close(Pg[n].grad, ref, rtol=1e-3, atol=1e-5 * float(ref.abs().max()))
    hs, u_noise, steps = fx.cond_sampler_inputs("det")
    with torch.no_grad():
        xs = orc.sample_edm_cond(P, cfg, fx.node_cond(hs).permute(0, 3, 1, 2), orc.SamplerParams(), u_noise.permute(0, 3, 1, 2), steps)
    close(xs, g["node::xs_last"], rtol=1e-3, atol=1e-4)
