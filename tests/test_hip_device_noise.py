"""GPU: the hot-path sampler with its per-step churn noise drawn ON THE DEVICE (mcedm_heun_sample_rng; models/mcedm.py:604-608 with
the reference's shipped configs/diff_sampler/edm_sampler.yaml: S_churn 15): equal, bit for bit, to mcedm_heun_sample fed with the
tensors mcedm_normal_fill writes for the same seed (the golden-tested plumbing), the masked entries untouched, moments of the
stream, HIP-graph replay == eager launches per seed, fresh noise per call, and seed_everything-style reproducibility through
PlMcedm.sample_edm(noise_source='device')."""
import math

import pytest
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    cfg = fx.CFG_P
    plan = L.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks, cfg.attn_resolutions,
                  cfg.resolution)
    P = orc.make_params(cfg, 7)
    packed = plan.pack({k: v.cuda() for k, v in P.items()})
    return L, plan, packed


def test_device_churn_noise_equals_the_materialised_draws(net):
    L, plan, packed = net
    cond, m, init, _ = fx.sampler_inputs("churn_u")
    cond, m, init = cond.cuda(), m.cuda(), init.cuda()
    N = 18
    sd = L.sampler_desc(orc.SamplerParams(timesteps=N, S_churn=15.0))
    seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
    steps = torch.stack([L.normal_fill(torch.empty(tuple(init.shape), dtype=torch.float64, device="cuda"), seed, i) for i in range(N)])
    want = plan.sample(packed, sd, cond, m, init, steps, return_last=False)
    got = plan.sample(packed, sd, cond, m, init, None, return_last=False, rng_seed=seed)
    assert torch.equal(got, want)
    other = plan.sample(packed, sd, cond, m, init, None, return_last=False, rng_seed=seed + 1)
    assert not torch.equal(other, want) and bool(torch.isfinite(other).all())
    # observed entries (mask == 0) are the clean conditioning at every step, whatever the noise (mcedm.py:597, 608, 618)
    known = cond[:, :2].permute(0, 2, 3, 1).double()
    obs = (m == 0).permute(0, 2, 3, 1)
    for t in range(got.shape[1]):
        assert torch.equal(got[:, t][obs], known[obs])
    with pytest.raises(RuntimeError):
        plan.sample(packed, sd, cond, m, init, steps, rng_seed=seed)          # both noise sources at once


def test_device_noise_stream_has_unit_normal_moments(net):
    L, _, _ = net
    seed = torch.tensor([77], dtype=torch.int64, device="cuda")
    z = torch.stack([L.normal_fill(torch.empty(1 << 18, dtype=torch.float64, device="cuda"), seed, d) for d in range(4)])
    n = z[0].numel()
    assert abs(float(z.mean())) < 5 / math.sqrt(4 * n) and abs(float(z.var()) - 1) < 5 * math.sqrt(2 / (4 * n))
    assert abs(float((z ** 4).mean()) - 3) < 0.05
    c = torch.corrcoef(z)
    assert float((c - torch.eye(4, device="cuda", dtype=torch.float64)).abs().max()) < 5 / math.sqrt(n)     # draws are independent


def test_module_sampler_with_device_noise_graph_equals_eager_and_follows_the_torch_seed(monkeypatch):
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    from mcedm_amd.mcedm import PlMcedm
    from tests.test_hip_module import hparams
    sp = hparams(fx.CFG_P, timesteps=6, S_churn=15.0).sampler
    m = PlMcedm(hparams(fx.CFG_P)).cuda()
    P = orc.make_params(fx.CFG_P, 7)
    with torch.no_grad():
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    m.noise_source = "device"
    cond, mk, _, _ = fx.sampler_inputs("churn_u")
    cond, mk = cond.cuda(), mk.cuda()
    hu = torch.zeros(4, 2, 32, 32).cuda()

    def run(seed):
        torch.manual_seed(seed)
        return m.sample_edm(hu, cond, mk, sp, return_last=True)
    real = torch.randn

    def no_big_draw(*a, **k):                  # the [N, B, 2, H, W] float64 tensor must not be materialised any more
        assert k.get("dtype") != torch.float64, "sample_edm materialised the per-step noise"
        return real(*a, **k)
    monkeypatch.setattr(torch, "randn", no_big_draw)
    a, b, c = run(3), run(3), run(4)
    assert len(m._graphs) == 1 and isinstance(next(iter(m._graphs.values())), L.GraphedSampler)
    assert next(iter(m._graphs.values())).step_noise is None
    assert torch.equal(a, b) and not torch.equal(a, c) and bool(torch.isfinite(a).all())
    monkeypatch.setenv("MCEDM_HIP_GRAPH", "0")
    e = run(3)
    assert torch.equal(a, e), "graph replay and eager launches differ for the same seed"
    m.noise_source = "elsewhere"
    with pytest.raises(RuntimeError):
        run(3)
