"""GPU, SURVEY.md section 8(f2): the single-task conditional EDM (reference models/ddim.py PlCondEdm) on the HIP path --
unmasked Heun sampler and training step through the drop-in ``mcedm_amd.ddim.PlCondEdm`` against the reference's golden
vectors (tests/golden/cond_edm.npz)."""
import pytest
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc
from tests.test_hip_module import hparams, wrap

pytestmark = pytest.mark.gpu


def cond_hparams(**sampler):
    hp = hparams(fx.CFG_C, **sampler)
    hp["name"] = "adm_edm_cond_h"
    hp.model.update(type="simple", var_type="fixedsmall", node_type=False)
    hp["diffusion"] = wrap(dict(beta_schedule="linear", beta_start=0.0001, beta_end=0.02, num_diffusion_timesteps=1000))
    return hp


@pytest.fixture()
def module(golden):
    import mcedm_amd  # noqa: F401
    from mcedm_amd.ddim import PlCondEdm
    assert torch.cuda.is_available()
    m = PlCondEdm(cond_hparams()).cuda()
    P = orc.make_params(fx.CFG_C, int(golden("cond_edm.npz")["seed"]))
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    return m


def close(got, ref, rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(got.detach().cpu(), torch.as_tensor(ref), rtol=rtol, atol=atol)


def test_state_dict_has_reference_buffers(module, golden):
    g = golden("cond_edm.npz")
    sd = module.state_dict()
    close(sd["betas"], g["betas"], rtol=0, atol=0)
    close(sd["logvar"], g["logvar"], rtol=1e-6, atol=1e-6)
    assert tuple(sd["model.enc.128x128_conv.weight"].shape) == (64, 2, 3, 3) and tuple(sd["model.out_conv.weight"].shape) == (1, 64, 3, 3)


@pytest.mark.parametrize("tag", list(fx.COND_SAMPLER_CASES))
def test_sample_edm_golden(module, golden, monkeypatch, tag):
    g = golden("cond_edm.npz")
    h, u_noise, steps = fx.cond_sampler_inputs(tag)
    sp = cond_hparams(S_churn=fx.COND_SAMPLER_CASES[tag]).sampler
    real_randn = torch.randn
    monkeypatch.setattr(torch, "randn", lambda *a, **k: torch.stack(steps).cuda() if k.get("dtype") == torch.float64 else real_randn(*a, **k))
    xs = module.sample_edm(h.cuda(), u_noise.cuda(), sp, return_last=False)
    monkeypatch.undo()
    assert xs.dtype == torch.float64 and tuple(xs.shape) == (3, 19, 32, 32, 1)
    close(xs[:, -1:], g[f"{tag}_xs_last"], rtol=1e-4, atol=1e-5)
    close(xs[:, ::6], g[f"{tag}_xs_traj"], rtol=1e-4, atol=1e-5)


def test_training_step_golden(module, golden, monkeypatch):
    g = golden("cond_edm.npz")
    h, u, noise, rnd_normal = fx.cond_training_inputs()
    st = fx.TRAIN_NORM_STATS
    module.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    module.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: noise.cuda())
    monkeypatch.setattr(torch, "randn", lambda *a, **k: rnd_normal)
    loss = module.training_step((h.cuda(), None, None, u.cuda()), 0)
    monkeypatch.undo()
    close(loss, torch.as_tensor(g["loss"]), rtol=1e-4, atol=1e-4)
    loss.backward()
    grads = dict(module.model.named_parameters())
    for n in fx.COND_GRAD_NAMES:
        ref = torch.as_tensor(g[f"grad::{n}"])
        close(grads[n].grad, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
