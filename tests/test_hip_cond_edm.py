"""GPU, SURVEY.md section 8(f2): the single-task conditional EDM (reference models/ddim.py PlCondEdm) on the HIP path --
unmasked Heun sampler and training step through the drop-in ``mcedm_amd.ddim.PlCondEdm`` against the reference's golden
vectors (tests/golden/cond_edm.npz)."""
import numpy as np
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


@pytest.mark.parametrize("system", ["swe_per", "darcy"])
def test_sample_edm_pde_guidance_golden(module, golden, monkeypatch, system):
    """guide_dx=True in the single-task sampler (models/ddim.py:1577-1579, 1589-1591): the PDE-residual
    gradient is evaluated on the device after every denoiser call.  Golden = the reference's own guided trajectories."""
    m = module
    g = golden("guided.npz")
    st = fx.STEP_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    m.set_pde_loss_function(system, False)
    h, u_noise, steps = fx.cond_sampler_inputs("det")
    sp = cond_hparams(guide_dx=True).sampler
    xs = m.sample_edm(h.cuda(), u_noise.cuda(), sp, return_last=False, guide_dx=True)
    ref = torch.as_tensor(g[f"{system}_xs_last"])
    scale = float(ref.abs().max())
    err = float((xs[:, -1:].cpu() - ref).abs().max())
    x0 = m.sample_edm(h.cuda(), u_noise.cuda(), sp, return_last=True, guide_dx=False)
    moved = float((x0.cpu() - ref).abs().max())
    print(f"guided sampler {system}: max|d| = {err:.3e}, max|x| = {scale:.1f}, guidance moved the sample by {moved:.3e}")
    traj = torch.as_tensor(g[f"{system}_xs_traj"])
    per_step = [float((xs[:, 6 * k].cpu() - traj[:, k]).abs().max() / traj[:, k].abs().max()) for k in range(traj.shape[1])]
    print("  relative error at steps 0 / 6 / 12 / 18:", " ".join(f"{e:.2e}" for e in per_step))
    if system == "darcy":
        # The Darcy log-probability gradient is a step-like function of the residual (d/dL log(2 (1 - sigmoid(1e5 L)) + 1e-12):
        # zero, or O(1e5), with a transition 1e-5 wide), and with a random-weight network the guided dynamics are explosive
        # (the guidance moves the sample by O(|x|)): the reference's own trajectory is not reproducible beyond the first
        # steps at fp32, on any hardware.  The gradient itself is pinned at op level (tests/test_pde.py); here the early
        # trajectory must agree and the sampler must stay finite.
        # (in the reference run a single grid cell falls inside that transition, from trajectory entry 2 on; which cell does
        # is decided by the last bits of the residual)
        k = int(g["darcy_first_guided_entry"])
        assert k >= 1
        close(xs[:, :k], torch.as_tensor(g["darcy_xs_head"])[:, :k], rtol=1e-4, atol=1e-5 * float(np.abs(g["darcy_xs_head"]).max()))
        assert bool(torch.isfinite(xs).all())
        return
    close(xs[:, -1:], ref, rtol=1e-4, atol=1e-5 * scale)
    close(xs[:, ::6], traj, rtol=1e-4, atol=1e-5 * scale)
    assert moved > 10 * err, "the guided and unguided samples must differ by far more than the parity error"


def test_joint_model_guidance_is_rejected_like_the_reference(golden):
    """PlMcedm.sample_edm(guide_dx=True) raises in the reference (models/mcedm.py:500-518 slices the wrong axis); the
    drop-in refuses it too instead of inventing semantics."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd.mcedm import PlMcedm
    from tests.test_hip_module import hparams as joint_hparams
    assert int(golden("guided.npz")["joint_model_guidance_raises"]) == 1
    m = PlMcedm(joint_hparams(fx.CFG_P)).cuda()
    z = torch.zeros(2, 2, 32, 32, device="cuda")
    with pytest.raises(NotImplementedError, match="raises in the reference"):
        m.sample_edm(z, z, z, joint_hparams(fx.CFG_P).sampler, guide_dx=True)
