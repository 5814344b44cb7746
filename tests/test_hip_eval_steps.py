"""GPU: the evaluation loops of the drop-ins for models/ddim.py against the reference's own outputs
(tests/golden/eval_steps.npz, written by oracle/make_golden_eval.py):

  * PlDdim.test_step / validation_step (models/ddim.py:294-533) -- BASELINE config 5 is run as
    ``eval_model.py ... diff_sampler.n_time_h=0 n_time_u=64`` -> trainer.test -> test_step -> sample_edm (RePaint);
  * PlCondEdm.test_step / validation_step (models/ddim.py:1154-1319), incl. Darcy with n_samples = 16 and a Darcy-guided
    run whose residuals keep the log-probability form saturated (the whole guided trajectory is comparable).
Every returned entry and every logged metric is compared."""
import numpy as np
import pytest
import torch

from oracle import ddpm_oracle as dorc
from oracle import fixtures as fx
from oracle import mcedm_oracle as orc
from tests.test_hip_cond_edm import cond_hparams
from tests.test_hip_ddpm import hparams as ddpm_hparams
from tests.test_hip_module import wrap

pytestmark = pytest.mark.gpu


def _fill(m, P, st, system):
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    m.normalizer_input.set_stats(torch.tensor(st[0]).cuda(), torch.tensor(st[1]).cuda())
    m.normalizer_target.set_stats(torch.tensor(st[2]).cuda(), torch.tensor(st[3]).cuda())
    m.set_pde_loss_function(system, False)
    logs = {}
    m.log = lambda name, value, **k: logs.__setitem__(name, torch.as_tensor(value).detach().cpu())
    m.current_epoch = 0
    return logs


def _compare(g, prefix, res, logs):
    ref_keys = sorted(k.split("::", 1)[1] for k in g if k.startswith(prefix + "::") and "::log::" not in k)
    assert sorted(res) == ref_keys
    for k in ref_keys:
        ref = torch.as_tensor(g[f"{prefix}::{k}"])
        got = torch.as_tensor(res[k]).detach().cpu()
        assert got.shape == ref.shape and got.dtype == ref.dtype, (prefix, k, got.shape, ref.shape, got.dtype, ref.dtype)
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5 * max(1.0, float(ref.abs().max())), msg=lambda s: f"{prefix} {k}: {s}")
    log_keys = sorted(k.split("::log::")[1] for k in g if k.startswith(prefix + "::log::"))
    assert sorted(logs) == log_keys, (sorted(logs), log_keys)
    for k in log_keys:
        ref = torch.as_tensor(g[f"{prefix}::log::{k}"])
        # the PDE metrics sum thousands of squared residuals that divide by the water height: rtol 2e-3 (as for PlMcedm)
        tol = dict(rtol=2e-3, atol=1e-6) if "pde" in k else dict(rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(logs[k].to(ref.dtype), ref, equal_nan=True, msg=lambda s: f"{prefix} log {k}: {s}", **tol)


def _repaint_sampler(N, R, churn, nth, ntu, n):
    return wrap(dict(name="edm", type="edm", timesteps=N, sigma_min=0.002, sigma_max=80, rho=7, S_churn=churn, S_min=0, S_max="inf",
                     S_noise=1, n_samples=n, n_repeat=R, n_time_h=nth, n_time_u=ntu, return_last=True, select_by_pde=False,
                     use_gt_pde_select=True, guide_dx=False, w=0.0, plot_scaled=False))


def _inject_repaint(monkeypatch, like_queue, steps, reps):
    """randn_like draws come from a queue; the drop-in draws the per-step / per-repeat noise in two fp64 torch.randn calls."""
    def randn_like(t, **k):
        v = like_queue.pop(0)
        assert tuple(v.shape) == tuple(t.shape), (v.shape, t.shape)
        return v.to(device=t.device, dtype=k.get("dtype", t.dtype))
    real_randn = torch.randn

    def randn(*a, **k):
        shape = a[0] if len(a) == 1 and isinstance(a[0], (tuple, list)) else a
        if k.get("dtype") == torch.float64 and len(shape) == 5:
            return torch.stack(steps).cuda()
        if k.get("dtype") == torch.float64 and len(shape) == 6:
            return torch.stack([torch.stack(r) for r in reps]).cuda()
        return real_randn(*a, **k)
    monkeypatch.setattr(torch, "randn_like", randn_like)
    monkeypatch.setattr(torch, "randn", randn)


@pytest.mark.parametrize("tag", list(fx.EVAL_DDPM_CASES))
def test_plddim_test_step_golden(golden, monkeypatch, tag):
    import mcedm_amd  # noqa: F401
    from mcedm_amd.ddim import PlDdim
    g = golden("eval_steps.npz")
    system, n, N, R, churn, nth, ntu = fx.EVAL_DDPM_CASES[tag]
    sp = _repaint_sampler(N, R, churn, nth, ntu, n)
    m = PlDdim(ddpm_hparams(sp)).cuda()
    logs = _fill(m, dorc.make_params(fx.CFG_D, 21), fx.EVAL_DDPM_STATS, system)
    m.set_test_sampler_params(sp)
    m.noise_source = "torch"            # the reference's draws are injected as torch tensors
    h, u, init, steps, reps, _ = fx.eval_ddpm_inputs(tag)
    queue = [init]
    _inject_repaint(monkeypatch, queue, steps, reps)
    res = m.test_step((h.cuda(), None, None, u.cuda()), 0)
    monkeypatch.undo()
    assert not queue
    _compare(g, f"ddpm_{tag}", res, logs)
    # known rows of every sample are the clean data, bit for bit (models/ddim.py:1041-1043)
    traj = res["traj"][:, 0].cpu()                                   # b h w n c
    state = torch.as_tensor(g[f"ddpm_{tag}::gt"]).double()
    for c, k in ((0, nth), (1, ntu)):
        for s in range(n):
            assert torch.equal(traj[:, :k, :, s, c], state[:, :k, :, c])


def test_plddim_validation_step_golden(golden, monkeypatch):
    import mcedm_amd  # noqa: F401
    from mcedm_amd.ddim import PlDdim
    g = golden("eval_steps.npz")
    system, n, N, R, churn, nth, ntu = fx.EVAL_DDPM_VAL
    sp = _repaint_sampler(N, R, churn, nth, ntu, n)
    m = PlDdim(ddpm_hparams(sp)).cuda()
    logs = _fill(m, dorc.make_params(fx.CFG_D, 21), fx.EVAL_DDPM_STATS, system)
    m.set_test_sampler_params(sp)          # builds edm_steps (the reference's run.py does this before fit / test too)
    m.noise_source = "torch"
    h, u, init, steps, reps, u_noise = fx.eval_ddpm_inputs("val")
    queue = [u_noise, init]
    _inject_repaint(monkeypatch, queue, steps, reps)
    batch = (h.cuda(), None, None, u.cuda())
    res = m.validation_step(batch, 0)
    monkeypatch.undo()
    assert not queue and res.pop("epoch") == 0
    _compare(g, "ddpm_val", res, logs)
    m.current_epoch = 7
    assert m.validation_step(batch, 0) == {"epoch": 7}


def _cond_module(sp, system, st):
    import mcedm_amd  # noqa: F401
    from mcedm_amd.ddim import PlCondEdm
    m = PlCondEdm(cond_hparams(**sp)).cuda()
    logs = _fill(m, orc.make_params(fx.CFG_C, 13), st, system)
    return m, logs


@pytest.mark.parametrize("tag", list(fx.EVAL_COND_CASES))
def test_plcondedm_test_step_golden(golden, monkeypatch, tag):
    g = golden("eval_steps.npz")
    system, n, guided, st = fx.EVAL_COND_CASES[tag]
    sp = dict(n_samples=n, guide_dx=guided)
    m, logs = _cond_module(sp, system, st)
    m.set_test_sampler_params(cond_hparams(**sp).sampler)
    h, u, init = fx.eval_cond_inputs(tag)
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: init.to(t.device))
    res = m.test_step((h.cuda(), None, None, u.cuda()), 0)
    monkeypatch.undo()
    _compare(g, f"cond_{tag}", res, logs)
    if guided:
        # the same call without guidance gives the same states: the device-side Darcy gradient of saturated cells is 0
        m.test_sparams = cond_hparams(n_samples=n, guide_dx=False).sampler
        monkeypatch.setattr(torch, "randn_like", lambda t, **k: init.to(t.device))
        res0 = m.test_step((h.cuda(), None, None, u.cuda()), 0)
        monkeypatch.undo()
        assert torch.equal(res0["traj"], res["traj"])


def test_plcondedm_validation_step_golden(golden, monkeypatch):
    g = golden("eval_steps.npz")
    m, logs = _cond_module({}, "swe_per", fx.STEP_NORM_STATS)
    h, u, init = fx.eval_cond_inputs("val")
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: init.to(t.device))
    batch = (h.cuda(), None, None, u.cuda())
    res = m.validation_step(batch, 0)
    monkeypatch.undo()
    assert res.pop("epoch") == 0
    _compare(g, "cond_val", res, logs)
    m.current_epoch = 7
    assert m.validation_step(batch, 0) == {"epoch": 7}


def test_plddim_test_step_with_the_ddim_sampler_golden(golden, monkeypatch):
    """sparams.type == 'ddim' (configs/diff_sampler/ddim_sampler.yaml, the reference's default): test_step runs
    sample_with_repeat (models/ddim.py:393-394) = mcedm_ddim_repaint_sample; states are fp32 there."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd.ddim import PlDdim
    g = golden("eval_steps.npz")
    system, n, N, skip, eta, R, nth, ntu = fx.EVAL_DDIM
    sp = wrap(dict(name="ddim", type="ddim", timesteps=N, skip_type=skip, eta=eta, n_samples=n, n_repeat=R, n_time_h=nth, n_time_u=ntu,
                   return_last=True, select_by_pde=False, use_gt_pde_select=True, guide_dx=False, w=0.0, plot_scaled=False))
    m = PlDdim(ddpm_hparams(sp)).cuda()
    st = fx.EVAL_DDPM_STATS
    logs = _fill(m, dorc.make_params(fx.CFG_D, 21), st, system)
    m.set_test_sampler_params(sp)
    h, u, init, _ = fx.ddim_inputs("eval", B=n * fx.EVAL_B)
    h, u = h[:fx.EVAL_B] * st[1] + st[0], u[:fx.EVAL_B] * st[3] + st[2]
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: init.to(t.device))
    res = m.test_step((h.cuda(), None, None, u.cuda()), 0)
    monkeypatch.undo()
    _compare(g, "ddpm_ddim", res, logs)
    with pytest.raises(NotImplementedError):
        m.sample(None, None, sp)
