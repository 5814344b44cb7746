"""GPU, SURVEY.md section 8(f3), second clause: the network CONDITIONED on the PDE-residual gradient (hparams.model.dx_cond,
reference models/adm_blocks.py:233-280, 334-362; models/ddim.py:601-639, 1424-1450, 1571, 1584, 1672-1681) in both of its
forms -- cat_dx=True (dx concatenated to conv_in's input) and cat_dx=False (dx_enc = Conv3x3 -> GELU -> Conv3x3 and
combine_enc) -- through the drop-ins ``mcedm_amd.adm_blocks.DhariwalUNet`` / ``mcedm_amd.ddim.PlCondEdm`` against the
reference's own outputs (tests/golden/dxcond.npz, oracle/make_golden_dxcond.py) and the oracle."""
import dataclasses

import pytest
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc
from tests.test_hip_cond_edm import cond_hparams

from tests._tol import close_per_entry

pytestmark = pytest.mark.gpu
MODES = ("cat", "enc")


def cfg_of(mode):
    return dataclasses.replace(fx.CFG_C, dx_channels=1, dx_mode=mode)


def make_module(mode, golden, **sampler):
    import mcedm_amd  # noqa: F401
    from mcedm_amd.ddim import PlCondEdm
    hp = cond_hparams(**sampler)
    hp.model.update(dx_cond=True, cat_dx=mode == "cat", dx_norm="prob", dx_detach=True)
    m = PlCondEdm(hp).cuda()
    P = orc.make_params(cfg_of(mode), int(golden("dxcond.npz")["seed"]))
    assert [n for n, _ in m.model.named_parameters()] == list(P)          # state_dict order of the reference
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    return m, P


def close(got, ref, rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(got.detach().cpu(), torch.as_tensor(ref), rtol=rtol, atol=atol)


@pytest.mark.parametrize("mode", MODES)
def test_network_forward_with_dx_golden(golden, mode):
    """DhariwalUNet.forward(x, noise_labels, cond, dx=dx) and dx=None (zeros concatenated / zero dx features, NOT dx_enc(0))."""
    g = golden("dxcond.npz")
    m, P = make_module(mode, golden)
    x, cond, dx, sig = fx.dxcond_net_inputs()
    with torch.no_grad():
        for tag, d in (("dx", dx), ("none", None)):
            F = m.model(x.cuda(), (sig.log() / 4).cuda(), cond.cuda(), dx=None if d is None else d.cuda())
            close(F, g[f"{mode}_F_{tag}"])
            close(F, orc.unet_forward(P, cfg_of(mode), x, sig.log() / 4, cond, dx=d))
    assert float((torch.as_tensor(g[f"{mode}_F_dx"]) - torch.as_tensor(g[f"{mode}_F_none"])).abs().max()) > 1e-2


@pytest.mark.parametrize("mode", MODES)
def test_get_denoised_with_dx_and_classifier_free_branch_golden(golden, mode):
    """models/ddim.py:1745-1763: with w != 0 the second evaluation drops cond AND dx."""
    g = golden("dxcond.npz")
    m, _ = make_module(mode, golden)
    x, cond, dx, sig = fx.dxcond_net_inputs()
    D, F = m.get_denoised(m.model, x.double().cuda(), sig.cuda(), cond=cond.cuda(), dx=dx.cuda(), w=0.5)
    close(D, g[f"{mode}_D_w"])
    close(F, g[f"{mode}_F_w"])
    # model_precond (w = 0) against the oracle
    Dp = m.model_precond(x.cuda(), sig.cuda(), cond.cuda(), dx=dx.cuda())
    close(Dp, orc.model_precond(orc.make_params(cfg_of(mode), int(g["seed"])), cfg_of(mode), x, sig, cond, dx=dx))


SAMPLER_CASES = [("cat", "swe_per", False), ("cat", "darcy", False), ("enc", "swe_per", False), ("enc", "swe_per", True),
                 ("enc", "darcy", False)]


@pytest.mark.parametrize("mode,system,guided", SAMPLER_CASES)
def test_sample_edm_dx_cond_golden(golden, mode, system, guided):
    """PlCondEdm.sample_edm of a dx_cond model: dx_in = get_dx_input(h, x) on the current noisy state before every denoiser
    call, computed on the device inside mcedm_heun_sample_dxcond; golden = the reference's own trajectories."""
    g = golden("dxcond.npz")
    m, _ = make_module(mode, golden, guide_dx=guided)
    st = fx.STEP_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    m.set_pde_loss_function(system, False)
    h, u_noise, _ = fx.cond_sampler_inputs("det")
    sp = m.sparams
    xs = m.sample_edm(h.cuda(), u_noise.cuda(), sp, return_last=False, guide_dx=guided)
    key = f"{mode}_{system}{'_guided' if guided else ''}"
    ref, traj = torch.as_tensor(g[f"{key}_xs_last"]), torch.as_tensor(g[f"{key}_xs_traj"])
    scale = float(traj.abs().max())
    err = float((xs[:, -1:].cpu() - ref).abs().max())
    per_step = [float((xs[:, 6 * k].cpu() - traj[:, k]).abs().max() / traj[:, k].abs().max()) for k in range(traj.shape[1])]
    print(f"dx_cond sampler {key}: max|d| = {err:.3e} (max|x| {scale:.1f}); relative error at steps 0 / 6 / 12 / 18:",
          " ".join(f"{e:.2e}" for e in per_step))
    assert xs.dtype == torch.float64 and tuple(xs.shape) == (3, 19, 32, 32, 1)
    if system == "darcy":
        # The Darcy log-probability gradient is a step-like function of the residual (zero or O(1e5), 1e-5 wide transition,
        # see tests/test_hip_cond_edm.py): which grid cells sit inside the transition is decided by the last bits of the
        # residual, and a flipped cell changes the network INPUT there by O(1e5).  How much of that the REFERENCE ITSELF shows is
        # measured, not asserted (round 4, oracle/make_golden_dxcond_stability.py -> tests/golden/dxcond_stability.npz): the
        # reference run on 1 thread instead of 8, or with one input moved by one unit in the last place, differs from its own
        # 8-thread trajectory in 0.23-0.78 % of the final state's entries of the enc case (first at sampled step 12, never by
        # step 6) and in none of the cat case.  The device is held to that: the cat case at the bar throughout; the enc case at
        # the bar through step 6 and, in the final state, at most 4x the reference's own worst self-disagreement outside it
        # (the device differs from the reference in the summation order of every convolution, not of one).
        # Round 5: every sampled step is held to ITS OWN magnitude (atol 1e-5 x max|that state|, 60x tighter on the final state
        # than the trajectory-wide atol was), and the fixture measures the reference against itself at that bar (`_own` keys):
        # 25.6-28.3 % of the enc case's final state there, still never by step 6, none of the cat case.  The allowance is
        # 1.5x the reference's own worst.
        stab = golden("dxcond_stability.npz")
        ref_self = float(stab[f"{mode}_darcy::final_bad_frac_own"].max())
        first_own = stab[f"{mode}_darcy::first_bad_step_own"]
        assert int(first_own[first_own >= 0].min(initial=99)) > 6
        close_per_entry(xs[:, 0:7:6], traj[:, :2], what=f"{key} steps 0 / 6")
        bad = (xs[:, -1:].cpu() - ref).abs() > 1e-5 * float(ref.abs().max()) + 1e-4 * ref.abs()
        frac = float(bad.double().mean())
        print(f"   entries of the final state outside the bar: {frac * 100:.3f} % (reference vs itself: up to {ref_self * 100:.3f} %)")
        assert bool(torch.isfinite(xs).all())
        if ref_self == 0.0:
            close_per_entry(xs[:, -1:], ref, what=f"{key} final state")
            close_per_entry(xs[:, ::6], traj, what=f"{key} trajectory")
        else:
            assert frac <= 1.5 * ref_self, (frac, ref_self)
        return
    # every sampled step against ITS OWN magnitude (the final state is ~100x smaller than the sigma = 80 initial one)
    close_per_entry(xs[:, -1:], ref, what=f"{key} final state")
    close_per_entry(xs[:, ::6], traj, what=f"{key} trajectory")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("branch", ["on", "off"])
def test_training_step_dx_cond_golden(golden, monkeypatch, mode, branch):
    """PlCondEdm.training_step of a dx_cond model: the dx branch is taken when torch.rand(1) > 0.1 (models/ddim.py:1673), dx
    is evaluated on the noised target and carries no gradient; loss and gradients (dx_enc / combine_enc included) against
    the reference's, and every gradient's squared norm."""
    g = golden("dxcond.npz")
    m, _ = make_module(mode, golden)
    h, u, noise, rnd_normal = fx.cond_training_inputs()
    st = fx.TRAIN_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    m.set_pde_loss_function("swe_per", False)
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: noise.cuda())
    monkeypatch.setattr(torch, "randn", lambda *a, **k: rnd_normal)
    monkeypatch.setattr(torch, "rand", lambda *a, **k: torch.tensor([0.5 if branch == "on" else 0.05]))
    loss = m.training_step((h.cuda(), None, None, u.cuda()), 0)
    monkeypatch.undo()
    close(loss, torch.as_tensor(g[f"{mode}_loss_{branch}"]), rtol=1e-4, atol=1e-4)
    loss.backward()
    params = dict(m.model.named_parameters())
    for n in fx.DXCOND_GRAD_NAMES[mode]:
        ref = torch.as_tensor(g[f"{mode}_grad_{branch}::{n}"])
        close(params[n].grad, ref, rtol=1e-4, atol=1e-5 * max(float(ref.abs().max()), 1e-30))
    sq = torch.tensor([float((p.grad.double() ** 2).sum()) for p in params.values()])
    torch.testing.assert_close(sq, torch.as_tensor(g[f"{mode}_grad_sqnorm_{branch}"]).to(sq.dtype), rtol=2e-4, atol=1e-12)
    if mode == "enc" and branch == "off":        # dx None: zero features, no path to dx_enc (adm_blocks.py:355-357)
        assert all(float(params[n].grad.abs().max()) == 0.0 for n in params if n.startswith("dx_enc."))


def test_other_dx_norms_are_rejected_like_the_reference(golden):
    """PlCondEdm.get_dx_input fails to unpack get_dx_pde's 3-D result for every dx_norm other than 'prob' (pinned)."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd.ddim import PlCondEdm
    assert int(golden("dxcond.npz")["dx_norm_l2_raises"]) == 1
    hp = cond_hparams()
    hp.model.update(dx_cond=True, cat_dx=True, dx_norm="l2")
    with pytest.raises(NotImplementedError, match="raises in the reference"):
        PlCondEdm(hp)


def test_dx_cond_and_guided_sampler_replay_from_a_graph_bit_identically(golden, monkeypatch):
    """sample_edm of a dx_cond model with guide_dx on top: the HIP-graph replay (default) equals the eager launches bit for
    bit, and a second replay reproduces the first."""
    m, _ = make_module("enc", golden, guide_dx=True)
    st = fx.STEP_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    m.set_pde_loss_function("swe_per", False)
    h, u_noise, _ = fx.cond_sampler_inputs("det")
    a = m.sample_edm(h.cuda(), u_noise.cuda(), m.sparams, return_last=True, guide_dx=True)
    b = m.sample_edm(h.cuda(), u_noise.cuda(), m.sparams, return_last=True, guide_dx=True)
    assert len(m._graphs) == 1, "the call must have been captured"
    monkeypatch.setenv("MCEDM_HIP_GRAPH", "0")
    c = m.sample_edm(h.cuda(), u_noise.cuda(), m.sparams, return_last=True, guide_dx=True)
    assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("mode", MODES)
def test_fused_trainer_step_with_dx_equals_autograd_plus_adam(golden, monkeypatch, mode):
    """The fused data-parallel step (mcedm_amd.train.EdmTrainer: flat buffers, fused clip + Adam + EMA) of a dx_cond model:
    same batch and dx as PlCondEdm.training_step + clip_grad_norm_(1.0) + torch.optim.Adam -> the same parameters."""
    from mcedm_amd.train import EdmTrainer
    h, u, noise, rnd_normal = fx.cond_training_inputs()
    st = fx.TRAIN_NORM_STATS

    def prepared():
        m, _ = make_module(mode, golden)
        m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
        m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
        m.set_pde_loss_function("swe_per", False)
        return m

    a = prepared()
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: noise.cuda())
    monkeypatch.setattr(torch, "randn", lambda *a_, **k: rnd_normal)
    monkeypatch.setattr(torch, "rand", lambda *a_, **k: torch.tensor([0.5]))
    loss = a.training_step((h.cuda(), None, None, u.cuda()), 0)
    monkeypatch.undo()
    loss.backward()
    opt = torch.optim.Adam(a.model.parameters(), lr=2e-4)
    torch.nn.utils.clip_grad_norm_(a.model.parameters(), 1.0)
    opt.step()
    b = prepared()
    tr = EdmTrainer(b)
    hn = ((h - st[0]) / st[1]).permute(0, 3, 1, 2).contiguous().cuda()
    un = ((u - st[2]) / st[3]).permute(0, 3, 1, 2).contiguous().cuda()
    l2 = tr.step(un, hn, None, noise.cuda(), rnd_normal.cuda(), dx_fn=lambda c, xn: b.get_dx_input(c[:, 0:1], xn))
    close(l2.reshape(()), loss.detach().reshape(()).cpu(), rtol=1e-5, atol=1e-5)
    pa, pb = dict(a.model.named_parameters()), dict(b.model.named_parameters())
    for n in pa:
        close(pb[n], pa[n].detach().cpu(), rtol=1e-4, atol=2e-6)


def test_dx_cond_sampler_on_a_rectangular_grid_against_the_oracle(golden):
    """T != X (the SWE residual runs along W, its time axis along H): a 4-step dx-conditioned sampler on 2 x 64 x 32 states
    against the oracle's (no reference golden at this shape; the oracle is pinned on the square cases above)."""
    m, P = make_module("enc", golden, timesteps=4)
    st = fx.STEP_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    m.set_pde_loss_function("swe_per", False)
    B, T, X = 2, 64, 32
    h, u_noise = fx.randn("dxcond/rect/h", B, T, X, 1), fx.randn("dxcond/rect/u", B, T, X, 1)
    xs = m.sample_edm(h.cuda(), u_noise.cuda(), m.sparams, return_last=False)
    with torch.no_grad():
        xo = orc.sample_edm_cond(P, cfg_of("enc"), h.permute(0, 3, 1, 2), orc.SamplerParams(timesteps=4), u_noise.permute(0, 3, 1, 2),
                                 None, return_last=False, dx_input=lambda hh, d: orc.guidance_dx_cond("swe_per", hh, d, st))
    assert tuple(xs.shape) == (B, 5, T, X, 1)
    close(xs, xo, rtol=1e-4, atol=1e-5 * float(xo.abs().max()))
