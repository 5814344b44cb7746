"""GPU, VERDICT r3 item 9 / ADVICE r3: the optional widenings of the joint model's conditioning input
(models/mcedm.py:25-34, 241-252: add_cond_mask, add_xt) through the drop-in ``mcedm_amd.mcedm.PlMcedm``, and
PlCondEdm.training_step with the conditioning dropped (cond_p < 1, models/ddim.py:1683-1684), against the reference's own
outputs (tests/golden/cond_in.npz, oracle/make_golden_r4.py)."""
import pytest
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc
from tests.test_hip_module import hparams
from tests.test_hip_cond_edm import cond_hparams

pytestmark = pytest.mark.gpu


def close(got, ref, rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(got.detach().cpu(), torch.as_tensor(ref), rtol=rtol, atol=atol)


def build(tag, seed):
    import mcedm_amd  # noqa: F401
    from mcedm_amd.mcedm import PlMcedm
    add_mask, add_xt = fx.COND_IN_CASES[tag]
    hp = hparams(fx.CFG_P)
    hp.model.add_cond_mask, hp.model.add_xt = add_mask, add_xt
    m = PlMcedm(hp).cuda()
    cfg = fx.cond_in_cfg(tag)
    assert hp.model.cond_channels == cfg.cond_channels           # the constructor widened it, like models/mcedm.py:28-34
    P = orc.make_params(cfg, seed)
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    st = fx.TRAIN_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    return m, cfg


@pytest.mark.parametrize("tag", list(fx.COND_IN_CASES))
def test_get_cond_in_training_step_and_sampler_golden(golden, monkeypatch, tag):
    g = golden("cond_in.npz")
    add_mask, add_xt = fx.COND_IN_CASES[tag]
    m, cfg = build(tag, int(g["seed"]))
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    dx, dt = fx.cond_in_xt()
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: cond_noise.to(t.device))
    cond_in = m.get_cond_in(m.data_transform(h.cuda(), u.cuda()), mask.cuda(), dx.cuda(), dt.cuda())
    monkeypatch.undo()
    close(cond_in, g[f"{tag}::cond_in"], rtol=0, atol=0)
    # training step: loss and the reference's gradients
    queue = ([] if add_mask else [cond_noise]) + [noise]
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: queue.pop(0).to(t.device))
    monkeypatch.setattr(torch, "randn", lambda *a, **k: rnd_normal)
    loss = m.training_step((h.cuda(), dx.cuda(), dt.cuda(), u.cuda(), mask.cuda()), 0)
    monkeypatch.undo()
    assert not queue
    close(loss, torch.as_tensor(g[f"{tag}::loss"]), rtol=1e-4, atol=1e-4)
    loss.backward()
    grads = dict(m.model.named_parameters())
    for n in fx.TRAIN_GRAD_NAMES:
        ref = torch.as_tensor(g[f"{tag}::grad::{n}"])
        close(grads[n].grad, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
    # sampler: hu_known = the first h_ch + u_ch channels of the widened conditioning (models/mcedm.py:590)
    init = fx.randn(f"condin/{tag}/init", 4, 2, 32, 32)
    sp = hparams(fx.CFG_P).sampler
    mc = mask.permute(0, 3, 1, 2).contiguous()
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: init.to(t.device))
    xs = m.sample_edm(torch.zeros(4, 2, 32, 32).cuda(), cond_in.permute(0, 3, 1, 2).contiguous(), mc.cuda(), sp, return_last=True)
    monkeypatch.undo()
    close(xs, g[f"{tag}::xs_last"], rtol=1e-4, atol=1e-5)
    obs = (mc == 0).permute(0, 2, 3, 1)
    assert torch.equal(xs[:, 0].cpu()[obs], cond_in[..., :2].double().cpu()[obs])     # observed entries are preserved exactly


def test_cond_edm_training_step_drops_the_conditioning(golden, monkeypatch):
    """cond_p < 1 (PlCondEdm defaults to 0.8 when the key is absent).  With cond_p = 0 every batch trains
    unconditioned; with cond_p = 0.8 the draw decides, in the reference's order of random draws."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd.ddim import PlCondEdm
    g, g1 = golden("cond_in.npz"), golden("cond_edm.npz")
    hp = cond_hparams()
    hp.model.cond_p = 0.8
    m = PlCondEdm(hp).cuda()
    P = orc.make_params(fx.CFG_C, 13)
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
    h, u, noise, rnd_normal = fx.cond_training_inputs()
    st = fx.TRAIN_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    for draw, key, gg in ((0.9, "cond_drop::", g), (0.3, "", g1)):           # 0.9 >= 0.8: dropped; 0.3: conditioned (= the round-1 golden)
        monkeypatch.setattr(torch, "randn_like", lambda t, **k: noise.cuda())
        monkeypatch.setattr(torch, "randn", lambda *a, **k: rnd_normal)
        monkeypatch.setattr(torch, "rand", lambda *a, **k: torch.tensor([draw]))
        loss = m.training_step((h.cuda(), None, None, u.cuda()), 0)
        monkeypatch.undo()
        close(loss, torch.as_tensor(gg[f"{key}loss"]), rtol=1e-4, atol=1e-4)
        m.zero_grad()
        loss.backward()
        grads = dict(m.model.named_parameters())
        for n in fx.COND_GRAD_NAMES:
            ref = torch.as_tensor(gg[f"{key}grad::{n}"])
            close(grads[n].grad, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))


def test_cond_edm_node_type_channel_golden(golden, monkeypatch):
    """PlCondEdm with hparams.model.node_type (models/ddim.py:36-38, 1105-1114): the constructor widens cond_channels by one,
    get_cond_in appends the boundary flag; training step and sampler against the reference's outputs."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd.ddim import PlCondEdm
    g = golden("cond_in.npz")
    hp = cond_hparams()
    hp.model.node_type = True
    m = PlCondEdm(hp).cuda()
    assert hp.model.cond_channels == fx.CFG_NODE.cond_channels
    P = orc.make_params(fx.CFG_NODE, 31)
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    h, u, noise, rnd_normal = fx.cond_training_inputs()
    st = fx.TRAIN_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    assert (m.h_ch, m.u_ch) == (1, 1)         # the constructor's default does not count the node_type channel
    hn4 = ((h - st[0]) / st[1]).cuda()
    close(m.get_cond_in(hn4, ((u - st[2]) / st[3]).cuda(), None, None), g["node::cond_in"], rtol=0, atol=0)
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: noise.cuda())
    monkeypatch.setattr(torch, "randn", lambda *a, **k: rnd_normal)
    loss = m.training_step((h.cuda(), None, None, u.cuda()), 0)
    monkeypatch.undo()
    close(loss, torch.as_tensor(g["node::loss"]), rtol=1e-4, atol=1e-4)
    loss.backward()
    grads = dict(m.model.named_parameters())
    for n in fx.COND_GRAD_NAMES:
        ref = torch.as_tensor(g[f"node::grad::{n}"])
        close(grads[n].grad, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
    hs, u_noise, _ = fx.cond_sampler_inputs("det")
    xs = m.sample_edm(fx.node_cond(hs).cuda(), u_noise.cuda(), cond_hparams().sampler, return_last=True)
    close(xs, g["node::xs_last"], rtol=1e-4, atol=1e-5)
