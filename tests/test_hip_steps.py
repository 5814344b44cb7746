"""GPU: the evaluation loops of the drop-in PlMcedm (test_step / validation_step) and the
classifier-free branch of the C sampler, against the reference's own outputs (tests/golden/steps.npz, written by
oracle/make_golden_steps.py).  'darcy_n16' is BASELINE config 4's path (conditional sampling, n_samples = 16, Darcy
residual metric) at plumbing size."""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc
from tests.test_hip_module import hparams

pytestmark = pytest.mark.gpu


class _Datamodule:
    def __init__(self, down_factor, down_interp):
        self.down_factor, self.down_interp = down_factor, down_interp


class _Trainer:
    def __init__(self, dm):
        self.datamodule = dm


def build(sampler, system):
    import mcedm_amd  # noqa: F401
    from mcedm_amd.mcedm import PlMcedm
    m = PlMcedm(hparams(fx.CFG_P, **sampler)).cuda()
    P = orc.make_params(fx.CFG_P, 7)
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    st = fx.STEP_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]).cuda(), torch.tensor(st[1]).cuda())
    m.normalizer_target.set_stats(torch.tensor(st[2]).cuda(), torch.tensor(st[3]).cuda())
    m.set_pde_loss_function(system, False)
    logs = {}
    m.log = lambda name, value, **k: logs.__setitem__(name, torch.as_tensor(value).detach().cpu())
    return m, logs


def close(got, ref, rtol=1e-4, atol=1e-5, what=""):
    got, ref = torch.as_tensor(got).detach().cpu(), torch.as_tensor(ref)
    assert got.shape == ref.shape and got.dtype == ref.dtype, (what, got.shape, ref.shape, got.dtype, ref.dtype)
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")


def inject(monkeypatch, queue):
    def randn_like(t, **k):
        v = queue.pop(0)
        assert tuple(v.shape) == tuple(t.shape), (v.shape, t.shape)
        return v.to(device=t.device, dtype=k.get("dtype", t.dtype))
    monkeypatch.setattr(torch, "randn_like", randn_like)


@pytest.mark.parametrize("tag", list(fx.STEP_CASES))
def test_test_step_golden(golden, monkeypatch, tag):
    g = golden("steps.npz")
    c = fx.STEP_CASES[tag]
    n = c["n_samples"]
    m, logs = build(dict(n_samples=n), c["system"])
    m.set_test_sampler_params(hparams(fx.CFG_P, n_samples=n).sampler)
    m.trainer = _Trainer(_Datamodule(c["down_factor"], c["down_interp"]))
    h, u, masks, noises = fx.step_inputs(tag)
    queue = []
    for name in masks:      # get_cond_in, the unused `noise`, the sampler's initial noise (S_churn = 0: no per-step draws)
        queue += [noises[name][0], torch.zeros(n * fx.STEP_B, 2, fx.STEP_T, fx.STEP_X), noises[name][1]]
    inject(monkeypatch, queue)
    res = m.test_step((h.cuda(), None, None, u.cuda(), {k: v.cuda() for k, v in masks.items()}), 0)
    monkeypatch.undo()
    assert not queue
    ref_keys = sorted(k.split("::", 1)[1] for k in g if k.startswith(tag + "::") and "::log::" not in k)
    assert sorted(res) == ref_keys                      # n_samples >= 15 drops traj_* / gt_* (mcedm.py:438)
    for k in ref_keys:
        close(res[k], g[f"{tag}::{k}"], what=f"{tag} {k}")
    for k in (k.split("::log::")[1] for k in g if k.startswith(tag + "::log::")):
        # the PDE residuals divide by h + 1e-8 and sum 4k-65k squared terms of O(1e3): rtol 2e-3 on the fp32 sum
        tol = dict(rtol=2e-3, atol=1e-6) if "pde" in k else dict(rtol=1e-4, atol=1e-6)
        close(logs[k].to(torch.as_tensor(g[f"{tag}::log::{k}"]).dtype), g[f"{tag}::log::{k}"], what=f"{tag} log {k}", **tol)


def test_validation_step_golden(golden, monkeypatch):
    g = golden("steps.npz")
    m, logs = build({}, "swe_per")
    m.current_epoch = 0
    h, u, masks, noises = fx.step_inputs("swe_n2")
    queue = [torch.zeros(fx.STEP_B, 2, fx.STEP_T, fx.STEP_X)]
    for name in masks:
        queue += [noises[name][0], noises[name][1][:fx.STEP_B]]
    inject(monkeypatch, queue)
    batch = (h.cuda(), None, None, u.cuda(), {k: v.cuda() for k, v in masks.items()})
    res = m.validation_step(batch, 0)
    monkeypatch.undo()
    assert not queue and res.pop("epoch") == 0
    ref_keys = sorted(k[5:] for k in g if k.startswith("val::") and "::log::" not in k)
    assert sorted(res) == ref_keys
    for k in ref_keys:
        close(res[k], g[f"val::{k}"], what=f"val {k}")
    for k in (k.split("::log::")[1] for k in g if k.startswith("val::log::")):
        tol = dict(rtol=2e-3, atol=1e-6) if "pde" in k else dict(rtol=1e-4, atol=1e-6)
        close(logs[k].to(torch.as_tensor(g[f"val::log::{k}"]).dtype), g[f"val::log::{k}"], what=f"val log {k}", **tol)
    m.current_epoch = 7                                  # not an evaluated epoch: early return (mcedm.py:284-285)
    assert m.validation_step(batch, 0) == {"epoch": 7}


def test_heun_sampler_classifier_free_golden(golden):
    """mcedm_heun_sample with w = 0.5: two U-Net evaluations per denoiser call blended on the device
    (csrc/plan.hip denoise_impl -> launch_precond_finish), models/mcedm.py:453-458."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    g = golden("steps.npz")
    cfg = fx.CFG_P
    plan = L.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks,
                  cfg.attn_resolutions, cfg.resolution)
    packed = plan.pack({k: v.cuda() for k, v in orc.make_params(cfg, 7).items()})
    cond, mk, init, steps = fx.sampler_inputs("det_u")
    sd = L.sampler_desc(orc.SamplerParams(w=fx.CFG_SAMPLER_W))
    xs = plan.sample(packed, sd, cond.cuda(), mk.cuda(), init.cuda(), None, return_last=False)
    scale = float(np.abs(g["cfg_u_xs_last"]).max())       # the guided trajectory of a random-weight net reaches O(300)
    err = float((xs[:, -1:].cpu() - torch.as_tensor(g["cfg_u_xs_last"])).abs().max())
    print(f"cfg sampler: max|d| = {err:.3e} on max|x| = {scale:.1f}")
    close(xs[:, -1:], g["cfg_u_xs_last"], rtol=1e-4, atol=1e-5 * scale, what="cfg last")
    close(xs[:, ::6], g["cfg_u_xs_traj"], rtol=1e-4, atol=1e-5 * scale, what="cfg trajectory")
    # and it differs from the unguided sampler (the branch really ran)
    x0 = plan.sample(packed, L.sampler_desc(orc.SamplerParams()), cond.cuda(), mk.cuda(), init.cuda(), None)
    assert float((x0 - xs[:, -1:]).abs().max()) > 1e-2
