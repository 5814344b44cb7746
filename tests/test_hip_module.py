"""GPU: the drop-in host modules (mcedm_amd.adm_blocks.DhariwalUNet / mcedm_amd.mcedm.PlMcedm) reproduce the reference's
golden vectors through their reference-named methods, and the fused trainer reproduces one Lightning optimisation step
(clip -> Adam -> EMA)."""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc

pytestmark = pytest.mark.gpu


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def wrap(d):
    return AttrDict({k: wrap(v) for k, v in d.items()}) if isinstance(d, dict) else d


def hparams(cfg, **sampler):
    sp = dict(name="edm", type="edm", timesteps=18, sigma_min=0.002, sigma_max=80, rho=7, S_churn=0.0, S_min=0, S_max="inf",
              S_noise=1, n_samples=1, n_repeat=2, n_time_h=128, n_time_u=0, return_last=True, select_by_pde=False,
              use_gt_pde_select=True, guide_dx=False, w=0.0, plot_scaled=False)
    sp.update(sampler)
    return wrap(dict(
        name="adm_edm_mcedm",
        model=dict(in_channels=cfg.in_channels, cond_channels=cfg.cond_channels, cat_cond=True, out_ch=cfg.out_ch, ch=cfg.ch,
                   ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks, attn_resolutions=list(cfg.attn_resolutions),
                   dropout=0.0, label_dim=0, augment_dim=0, label_dropout=0, ema_rate=0.999, ema=True, resamp_with_conv=True,
                   resolution=cfg.resolution, self_cond=False, cond_p=1.0, dx_cond=False, cat_dx=False, dx_norm="l2",
                   dx_detach=False, add_cond_mask=False, add_xt=False),
        data=dict(normalization="gauss", uniform_dequantization=False, gaussian_dequantization=False, rescaled=False),
        optimization=dict(optimizer="Adam", lr=0.0002, weight_decay=0.0, beta1=0.9, amsgrad=False, eps=1e-8, grad_clip=1.0,
                          loss="l2", pde_loss_lambda=0.0, pde_loss_prop_t=False, use_gt_pde=False, factor=0.3, step_size=50),
        sampler=sp))


@pytest.fixture()
def module():
    import mcedm_amd  # noqa: F401
    from mcedm_amd.mcedm import PlMcedm
    assert torch.cuda.is_available()
    m = PlMcedm(hparams(fx.CFG_P)).cuda()
    P = orc.make_params(fx.CFG_P, 7)
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    return m, P


def close(got, ref, rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(got.detach().cpu(), torch.as_tensor(ref), rtol=rtol, atol=atol)


def test_state_dict_keys_match_reference_layout(module):
    m, P = module
    sd = m.state_dict()
    names = [n for n, _ in orc.param_shapes(fx.CFG_P)]
    for n in names:
        assert tuple(sd[f"model.{n}"].shape) == tuple(P[n].shape)
        assert f"ema_model.ma_model.{n}" in sd
    # 204 entries per network (196 parameters + 8 resample_filter buffers) + 4 normaliser buffers
    assert sum(k.startswith("model.") for k in sd) == 204 and len(sd) == 412
    assert tuple(sd["model.enc.64x64_down.conv0.resample_filter"].shape) == (1, 1, 2, 2)


def test_forward_precond_denoised_golden(module, golden):
    m, _ = module
    g = golden("unet_P.npz")
    x, cond = fx.randn("unet_P/x", 4, 2, 32, 32).cuda(), fx.randn("unet_P/cond", 4, 2, 32, 32).cuda()
    with torch.no_grad():
        close(m.model(x, fx.UNET_LABELS["nB"].cuda(), cond), g["F_nB"])
        close(m.model(x, torch.tensor([0.3]).cuda(), None), g["F_nocond"])
    close(m.model_precond(x, fx.PRECOND_SIGMA_B.cuda(), cond), g["D_sigmaB"])
    D, F = m.get_denoised(m.ema_model, x.double(), torch.tensor(0.7).double(), cond=cond, w=0.5)
    close(D, g["D_cfg_w05"])
    with pytest.raises(NotImplementedError):
        m.model(x, torch.tensor([0.3]).cuda(), cond, dx=x)


def test_sample_edm_method_golden(module, golden, monkeypatch):
    m, _ = module
    g = golden("sampler_P.npz")
    for tag in ("det_u", "churn_u"):
        cond, mk, init, steps = fx.sampler_inputs(tag)
        sp = hparams(fx.CFG_P, S_churn=fx.SAMPLER_CASES[tag][0]).sampler
        queue = [init.cuda()]
        real_randn = torch.randn
        monkeypatch.setattr(torch, "randn_like", lambda t, **k: queue.pop(0).to(k.get("dtype", t.dtype)))
        monkeypatch.setattr(torch, "randn", lambda *a, **k: torch.stack(steps).cuda() if k.get("dtype") == torch.float64 else real_randn(*a, **k))
        xs = m.sample_edm(torch.zeros(4, 2, 32, 32).cuda(), cond.cuda(), mk.cuda(), sp, return_last=True)
        monkeypatch.undo()
        assert xs.dtype == torch.float64 and tuple(xs.shape) == (4, 1, 32, 32, 2)
        close(xs, g[f"{tag}_xs_last"], rtol=1e-4, atol=1e-5)


def test_training_step_autograd_and_fused_trainer(module, golden, monkeypatch):
    m, P = module
    g = golden("training_P.npz")
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    st = fx.TRAIN_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    like = [cond_noise.cuda(), noise.cuda()]
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: like.pop(0))
    monkeypatch.setattr(torch, "randn", lambda *a, **k: rnd_normal)
    loss = m.training_step((h.cuda(), None, None, u.cuda(), mask.cuda()), 0)
    monkeypatch.undo()
    close(loss, torch.as_tensor(g["loss"]), rtol=1e-4, atol=1e-4)
    loss.backward()
    grads = dict(m.model.named_parameters())
    for n in fx.TRAIN_GRAD_NAMES:
        ref = torch.as_tensor(g[f"grad::{n}"])
        close(grads[n].grad, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
    # the fused trainer: same batch -> one clip/Adam/EMA step equal to Lightning's
    from mcedm_amd.train import EdmTrainer
    m.zero_grad()
    tr = EdmTrainer(m)
    xc, cond_in, mc = fx.training_nchw(h, u, mask, cond_noise)
    l2 = tr.step(xc.cuda(), cond_in.cuda(), mc.cuda(), noise.cuda(), rnd_normal.cuda())
    close(l2.reshape(()), torch.as_tensor(g["loss"]), rtol=1e-4, atol=1e-4)
    new_p, new_e = dict(m.model.named_parameters()), dict(m.ema_model.ma_model.named_parameters())
    for n in fx.TRAIN_GRAD_NAMES:
        close(new_p[n], g[f"adam::{n}"], rtol=1e-4, atol=2e-6)
        close(new_e[n], g[f"ema::{n}"], rtol=1e-5, atol=1e-6)
    # parameters changed in place -> the next forward must see re-packed weights
    x, cond = fx.randn("unet_P/x", 4, 2, 32, 32).cuda(), fx.randn("unet_P/cond", 4, 2, 32, 32).cuda()
    with torch.no_grad():
        F1 = m.model(x, torch.tensor([0.3]).cuda(), cond)
        Pn = {k: v.detach().cpu() for k, v in new_p.items()}
        close(F1, orc.unet_forward(Pn, fx.CFG_P, x.cpu(), torch.tensor([0.3]), cond.cpu()))


def test_fused_trainer_resumes_bit_for_bit_from_a_checkpoint(module):
    """Checkpoint / resume (configs/callbacks/callbacks_ddim.yaml:1-10, run.py:68-72) for the fused trainer: module
    state_dict (weights + EMA copy) and the Adam state in torch.optim.Adam form after step 1, loaded into a FRESH module and
    trainer; step 2 from there equals step 2 of the uninterrupted run bit for bit (the step is bitwise reproducible)."""
    import copy
    import mcedm_amd  # noqa: F401
    from mcedm_amd.mcedm import PlMcedm
    from mcedm_amd.train import EdmTrainer
    m, _ = module
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    xc, cond_in, mc = (t.cuda() for t in fx.training_nchw(h, u, mask, cond_noise))
    nz, rn = noise.cuda(), rnd_normal.cuda()
    tr = EdmTrainer(m)
    tr.step(xc, cond_in, mc, nz, rn)
    ckpt = {"state_dict": copy.deepcopy(m.state_dict()), "optimizer_states": [copy.deepcopy(tr.optimizer_state_dict())]}
    tr.step(xc, cond_in, mc, nz * 0.5, rn)
    want = {k: v.clone() for k, v in m.state_dict().items()}
    # resume
    m2 = PlMcedm(hparams(fx.CFG_P)).cuda()
    m2.load_state_dict(ckpt["state_dict"], strict=True)
    tr2 = EdmTrainer(m2)
    tr2.load_optimizer_state_dict(ckpt["optimizer_states"][0])
    assert tr2.step_count == 1
    tr2.step(xc, cond_in, mc, nz * 0.5, rn)
    got = m2.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # the exported state is what torch's own Adam accepts for the same parameter list
    opt = torch.optim.Adam(m2.model.parameters(), lr=1.0)
    opt.load_state_dict(tr2.optimizer_state_dict())
    assert float(opt.state[next(iter(m2.model.parameters()))]["step"]) == 2


def _fresh_module():
    import mcedm_amd  # noqa: F401
    from mcedm_amd.mcedm import PlMcedm
    m = PlMcedm(hparams(fx.CFG_P)).cuda()
    P = orc.make_params(fx.CFG_P, 7)
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    st = fx.TRAIN_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    return m


def test_fused_optimizer_behind_the_lightning_seam(golden, monkeypatch):
    """VERDICT r4 item 5 (models/mcedm.py:139-168): the way Lightning drives a step -- configure_optimizers, then
    optimizer_step(closure = zero_grad + training_step + backward + configure_gradient_clipping) -- runs the fused kernels:
    the optimiser is a torch.optim.Optimizer over flat buffers, the clip is folded into its kernel, the separate EmaModel.update
    pass is skipped.  Result: the reference's golden Adam / EMA values, and train.EdmTrainer's bit for bit."""
    from mcedm_amd.optim import FusedAdamEma
    from mcedm_amd.train import EdmTrainer
    g = golden("training_P.npz")
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    m = _fresh_module()
    opt = m.configure_optimizers()["optimizer"]
    assert isinstance(opt, FusedAdamEma) and isinstance(opt, torch.optim.Optimizer)
    ema_calls = []
    monkeypatch.setattr(m.ema_model, "update", lambda *a, **k: ema_calls.append(1))
    batch = (h.cuda(), None, None, u.cuda(), mask.cuda())

    def closure():
        opt.zero_grad()
        like = [cond_noise.cuda(), noise.cuda()]
        real_like, real_randn = torch.randn_like, torch.randn
        torch.randn_like, torch.randn = (lambda t, **k: like.pop(0)), (lambda *a, **k: rnd_normal)
        try:
            loss = m.training_step(batch, 0)
        finally:
            torch.randn_like, torch.randn = real_like, real_randn
        loss.backward()
        m.configure_gradient_clipping(opt, 0, 1.0, "norm")          # pytorch_lightning 1.8's positional form
        return loss
    m.optimizer_step(0, 0, opt, 0, closure)
    assert not ema_calls, "the fused kernel already updated the EMA copy: EmaModel.update must be skipped"
    assert opt.max_norm == 1.0 and opt.step_count == 1
    new_p, new_e = dict(m.model.named_parameters()), dict(m.ema_model.ma_model.named_parameters())
    for n in fx.TRAIN_GRAD_NAMES:
        close(new_p[n], g[f"adam::{n}"], rtol=1e-4, atol=2e-6)
        close(new_e[n], g[f"ema::{n}"], rtol=1e-5, atol=1e-6)
    # bit for bit the custom loop's step
    m2 = _fresh_module()
    tr = EdmTrainer(m2)
    xc, cond_in, mc = fx.training_nchw(h, u, mask, cond_noise)
    tr.step(xc.cuda(), cond_in.cuda(), mc.cuda(), noise.cuda(), rnd_normal.cuda())
    assert torch.equal(opt.flat_p, tr.flat_p) and torch.equal(opt.flat_ema, tr.flat_ema)
    assert torch.equal(opt.flat_m, tr.flat_m) and torch.equal(opt.flat_v, tr.flat_v)
    # the next forward sees the new weights (packed copies were invalidated)
    x, cond = fx.randn("unet_P/x", 4, 2, 32, 32).cuda(), fx.randn("unet_P/cond", 4, 2, 32, 32).cuda()
    with torch.no_grad():
        F1 = m.model(x, torch.tensor([0.3]).cuda(), cond)
        close(F1, orc.unet_forward({k: v.detach().cpu() for k, v in new_p.items()}, fx.CFG_P, x.cpu(), torch.tensor([0.3]), cond.cpu()))
    # state_dict is torch.optim.Adam's: a plain Adam over the same parameters loads it, takes a step, and its state loads back
    sd = opt.state_dict()
    assert sd["param_groups"][0]["lr"] == 0.0002 and len(sd["state"]) == len(list(m.model.parameters()))
    plain = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in m.model.parameters()], lr=1.0)
    plain.load_state_dict(sd)
    assert float(plain.state[plain.param_groups[0]["params"][0]]["step"]) == 1 and plain.param_groups[0]["lr"] == 0.0002
    m3 = _fresh_module()
    opt3 = m3.configure_optimizers()["optimizer"]
    opt3.load_state_dict(plain.state_dict())
    assert opt3.step_count == 1 and torch.equal(opt3.flat_m, opt.flat_m) and torch.equal(opt3.flat_v, opt.flat_v)
    # MCEDM_FUSED_OPT=0: the plain path of the reference, with the separate EMA pass
    monkeypatch.setenv("MCEDM_FUSED_OPT", "0")
    m4 = _fresh_module()
    o4 = m4.configure_optimizers()["optimizer"]
    assert type(o4) is torch.optim.Adam
