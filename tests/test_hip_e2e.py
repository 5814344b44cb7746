"""GPU: the pieces around the hot path driven end to end through libmcedm_hip.so (SURVEY.md section 8 e / f4):

  * a reference-layout Lightning checkpoint -> load_reference_checkpoint -> PlMcedm.sample_edm == the reference's
    sampler golden (tests/golden/sampler_P.npz);
  * an HDF5MaskDatamodule batch (npz store of the reference's sample layout) -> setup('fit') statistics -> test_step with
    the reference's dict keys, equal to the oracle's evaluation loop on the same batch;
  * the bucketed backward: when bucket k's event fires every gradient of bucket k is final; GradSync's side-stream
    ordering with an injected stream-ordered reduce;
  * the HIP-graph sampler == the eager call at S128, capture failure falls back to eager, a replaced nn.Parameter is seen.
"""
import io
import warnings

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc
from tests.test_hip_module import hparams

pytestmark = pytest.mark.gpu


def _module(cfg=fx.CFG_P, **sampler):
    import mcedm_amd  # noqa: F401
    from mcedm_amd.mcedm import PlMcedm
    return PlMcedm(hparams(cfg, **sampler)).cuda()


def _reference_state_dict(cfg, P):
    """The 412-entry state_dict layout of a reference PlMcedm checkpoint: model.*, ema_model.ma_model.*,
    the resample_filter buffers of the up / down blocks and the normaliser buffers."""
    sd = {}
    for n, v in P.items():
        sd[f"model.{n}"] = v
        sd[f"ema_model.ma_model.{n}"] = v.clone()
    for net in ("model.", "ema_model.ma_model."):
        for key in ("enc.64x64_down", "enc.32x32_down", "dec.64x64_up", "dec.128x128_up"):
            for conv in ("conv0", "skip"):
                sd[f"{net}{key}.{conv}.resample_filter"] = torch.full((1, 1, 2, 2), 0.25)
    sd.update({"normalizer_input.subtract": torch.tensor(1.4), "normalizer_input.divide": torch.tensor(0.2),
               "normalizer_target.subtract": torch.tensor(0.0), "normalizer_target.divide": torch.tensor(0.5)})
    return sd


class _Hyper(dict):
    """Stands in for the OmegaConf container a real Lightning checkpoint carries under 'hyper_parameters': a dict subclass
    with attributes, which torch.load(weights_only=True) refuses."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.flags = {"resolve": True}


def test_checkpoint_to_sampler_golden(golden, monkeypatch):
    from mcedm_amd import checkpoint as ck
    g = golden("sampler_P.npz")
    cfg = fx.CFG_P
    sd = _reference_state_dict(cfg, orc.make_params(cfg, int(g["seed"])))
    assert len(sd) == 412
    buf = io.BytesIO()
    # a Lightning file: extra entries that weights_only=True refuses (an object of a class this image may not have)
    torch.save({"state_dict": sd, "epoch": 9, "global_step": 77, "pytorch-lightning_version": "1.8.0",
                "hyper_parameters": _Hyper(lr=2e-4, name="adm_edm_mcedm")}, buf)
    buf.seek(0)
    m = _module()
    with torch.no_grad():                               # start from different weights: the load must matter
        for p in m.parameters():
            p.add_(0.01)
    meta = ck.load_reference_checkpoint(m, buf, strict=True)
    assert meta["epoch"] == 9 and float(m.normalizer_input.divide) == pytest.approx(0.2)
    cond, mk, init, _ = fx.sampler_inputs("det_u")
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: init.to(t.device))
    xs = m.sample_edm(torch.zeros(4, 2, 32, 32).cuda(), cond.cuda(), mk.cuda(), hparams(cfg).sampler, return_last=True)
    monkeypatch.undo()
    torch.testing.assert_close(xs.cpu(), torch.as_tensor(g["det_u_xs_last"]), rtol=1e-4, atol=1e-5)
    # and a round trip through save_checkpoint gives the same sampler output
    out = io.BytesIO()
    ck.save_checkpoint(m, out, epoch=10)
    out.seek(0)
    m2 = _module()
    ck.load_reference_checkpoint(m2, out, strict=True)
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: init.to(t.device))
    xs2 = m2.sample_edm(torch.zeros(4, 2, 32, 32).cuda(), cond.cuda(), mk.cuda(), hparams(cfg).sampler, return_last=True)
    monkeypatch.undo()
    assert torch.equal(xs, xs2)


class _Trainer:
    def __init__(self, dm):
        self.datamodule = dm


def test_datamodule_batch_through_test_step(monkeypatch):
    """datamodules/pl_datamodule.py:221-317 + h5_dataset.py:306-393 -> models/mcedm.py:343-441 on the device."""
    from mcedm_amd import data as D
    store = D.NpzStore(fx.data_tree_flat())
    dm = D.HDF5MaskDatamodule(store, store, store, return_abs_coords=True, return_grid=True, batch_size=2, test_batch_size=2,
                              dataset_cls=D.HDF5TimeMaskDataset, dataset_kwargs=dict(add_time_masks=True))
    dm.setup()
    n = 2
    m = _module(n_samples=n)
    P = orc.make_params(fx.CFG_P, 7)
    with torch.no_grad():
        for k, p in m.model.named_parameters():
            p.copy_(P[k])
        for k, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[k])
    m.trainer = _Trainer(dm)
    m.setup("fit")                                         # normaliser statistics from the datamodule (mcedm.py:106-123)
    m.set_test_sampler_params(hparams(fx.CFG_P, n_samples=n).sampler)
    m.set_pde_loss_function("swe_per", False)
    logs = {}
    m.log = lambda name, value, **k: logs.__setitem__(name, float(value))
    h, tg, xg, u, masks = next(iter(dm.test_dataloader()))
    assert list(masks) == ["hu", "u", "h"]
    B, T, X = h.shape[0], h.shape[1], h.shape[2]
    noises = {k: (fx.randn(f"e2e/{k}/cond", B, T, X, 2), fx.randn(f"e2e/{k}/init", n * B, 2, T, X)) for k in masks}
    queue = []
    for k in masks:
        queue += [noises[k][0], torch.zeros(n * B, 2, T, X), noises[k][1]]

    def randn_like(t, **kw):
        v = queue.pop(0)
        assert tuple(v.shape) == tuple(t.shape)
        return v.to(device=t.device, dtype=kw.get("dtype", t.dtype))
    monkeypatch.setattr(torch, "randn_like", randn_like)
    res = m.test_step((h.cuda(), tg.cuda(), xg.cuda(), u.cuda(), {k: v.cuda() for k, v in masks.items()}), 0)
    monkeypatch.undo()
    assert not queue
    want = set()
    for k in masks:
        want |= {f"loss_{k}", f"loss_{k}_un", f"traj_{k}", f"gt_{k}"}
    assert set(res) == want                               # what the plotting callbacks read (custom_callbacks.py:146-161)
    st = (float(dm.input_mean), float(dm.input_std), float(dm.target_mean), float(dm.target_std))
    with torch.no_grad():
        o = orc.eval_test_step(P, fx.CFG_P, h, u, masks, noises, st, orc.SamplerParams(), n, "swe_per", 1)
    for k in sorted(want):
        got, ref = res[k].detach().cpu(), o[k]
        assert got.shape == ref.shape and bool(torch.isfinite(got).all()), k
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5 * max(1.0, float(ref.abs().max())), msg=lambda s: f"{k}: {s}")
    assert tuple(res["traj_u"].shape) == (B, 1, T, X, n, 2) and res["traj_u"].dtype == torch.float64
    assert all(np.isfinite(v) for v in logs.values()) and "test_pde_loss_u" in logs and "test_mae_hu_un" in logs


def _training_call(L, cfg, B, seed):
    from tests.test_hip_backward import make_plan
    g = torch.Generator().manual_seed(seed)
    H = W = 32
    P = orc.make_params(cfg, 3)
    plan = make_plan(L, cfg)
    params = {k: v.cuda() for k, v in P.items()}
    packed = plan.pack(params)
    x = torch.randn(B, 2, H, W, generator=g).cuda()
    mc = (torch.rand(B, 2, H, W, generator=g) > 0.5).float().cuda()
    cond = (x * (1 - mc) + torch.randn(B, 2, H, W, generator=g).cuda() * mc).contiguous()
    noise = torch.randn(B, 2, H, W, generator=g).cuda()
    rnd = torch.randn(B, generator=g).cuda()
    ws = L.Workspace()

    def fwd():
        x_noise, sigma = L.edm_noise_inputs(x, mc, noise, rnd)
        D = plan.denoise(packed, x_noise, sigma, cond=cond, ws=ws, training=True)
        _, dD = L.edm_loss(D, x, mc, sigma)
        return x_noise, sigma, dD
    return plan, params, packed, cond, ws, fwd


def _flat_views(plan, params):
    tens = [params[n] for n in plan.param_names]
    flat = torch.full((sum(t.numel() for t in tens),), float("nan"), device="cuda")
    views, off = [], 0
    for t in tens:
        views.append(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    return flat, views, [t.numel() for t in tens]


def _close_grads(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    lim = 1e-5 * b.abs().max() + 1e-4 * b.abs()           # the split-K sums of wgrad are atomic: last-bit run-to-run noise
    assert bool(((a - b).abs() <= lim).all()), float((a - b).abs().max())


def test_bucketed_backward_events_mark_final_gradients():
    """mcedm_edm_denoise_backward_bucketed (csrc/backward.hip): event k is recorded once every parameter >= bucket_first[k]
    has its FINAL gradient enqueued (wgrad accumulates into scratch by atomics and a later kernel writes the gradient).  A
    side stream waits for each event and snapshots that bucket's range immediately: every snapshot must equal the final
    buffer bit for bit, and the whole buffer must equal the unbucketed run up to the atomics' noise."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    from mcedm_amd.train import GradSync
    plan, params, packed, cond, ws, fwd = _training_call(L, fx.CFG_P, 16, 5)
    flat1, views1, numels = _flat_views(plan, params)
    x_noise, sigma, dD = fwd()
    plan.denoise_backward(packed, params, x_noise, sigma, cond, dD, views1, ws)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(flat1).all())

    firsts = plan.grad_buckets(4)
    assert len(firsts) >= 2 and firsts[-1] == 0 and all(a > b for a, b in zip(firsts, firsts[1:]))
    flat, views, _ = _flat_views(plan, params)
    sync = GradSync(flat, numels, firsts)
    snap = torch.full_like(flat, float("nan"))
    for rep in range(3):                                   # repeated: the snapshots race the rest of the backward
        flat.fill_(float("nan"))
        snap.fill_(float("nan"))
        x_noise, sigma, dD = fwd()
        torch.cuda.synchronize()
        plan.denoise_backward(packed, params, x_noise, sigma, cond, dD, views, ws, bucket_first=sync.bucket_first,
                              bucket_events=sync.events)
        for ev, (lo, hi) in zip(sync.events, sync.ranges):
            sync.side.wait_event(ev)
            with torch.cuda.stream(sync.side):
                snap[lo:hi].copy_(flat[lo:hi])
        torch.cuda.synchronize()
        covered = sorted(sync.ranges)
        assert covered[0][0] == 0 and covered[-1][1] == flat.numel() and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
        assert torch.equal(snap, flat), f"repetition {rep}: a gradient changed after its bucket's event fired"
        _close_grads(flat, flat1)


def test_gradsync_side_stream_order_with_injected_reduce(monkeypatch):
    """train.GradSync.launch()/join() as a 2-rank job would run them, with dist.all_reduce replaced by a stream-ordered
    stand-in (x2 on the current stream).  If a bucket were reduced before its gradients landed, the later writes would
    leave un-doubled values behind."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    from mcedm_amd import train as TR
    plan, params, packed, cond, ws, fwd = _training_call(L, fx.CFG_P, 16, 6)
    flat1, views1, numels = _flat_views(plan, params)
    x_noise, sigma, dD = fwd()
    plan.denoise_backward(packed, params, x_noise, sigma, cond, dD, views1, ws)
    flat, views, _ = _flat_views(plan, params)
    sync = TR.GradSync(flat, numels, plan.grad_buckets(4))
    sync.world = 2
    calls = []

    def fake_all_reduce(t, op=None):
        assert torch.cuda.current_stream() == sync.side
        calls.append(t.numel())
        t.mul_(2.0)
    monkeypatch.setattr(TR.dist, "all_reduce", fake_all_reduce)
    x_noise, sigma, dD = fwd()
    plan.denoise_backward(packed, params, x_noise, sigma, cond, dD, views, ws, bucket_first=sync.bucket_first,
                          bucket_events=sync.events)
    sync.launch()
    sync.join()
    sq = L.sqnorm(flat)                                    # the step's next kernel, on the main stream after join()
    torch.cuda.synchronize()
    assert len(calls) == len(sync.ranges) and sum(calls) == flat.numel()
    _close_grads(flat, 2.0 * flat1)
    assert float(sq) == pytest.approx(4.0 * float((flat1.double() ** 2).sum()), rel=1e-4)


def test_graphed_sampler_equals_eager_at_full_size():
    """The HIP-graph replay the bench times (lib.GraphedSampler) == the eager mcedm_heun_sample call, bit for bit, at S128."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    from tests.test_hip_fullsize import CFG, inputs
    plan = L.Plan(CFG.in_channels, CFG.cond_channels, CFG.out_ch, CFG.ch, CFG.ch_mult, CFG.num_res_blocks,
                  CFG.attn_resolutions, CFG.resolution)
    packed = plan.pack({k: v.cuda() for k, v in orc.make_params(CFG, 7).items()})
    B = 8
    cond, m, init = inputs(B, seed=11)
    sd = L.sampler_desc(orc.SamplerParams(timesteps=18))
    ws = L.Workspace()
    eager = plan.sample(packed, sd, cond.cuda(), m.cuda(), init.cuda(), None, ws=ws)
    gs = L.GraphedSampler(plan, packed, sd, B, 128, 128, ws=ws)
    out1 = gs(cond.cuda(), m.cuda(), init.cuda()).clone()
    cond2, m2, init2 = inputs(B, seed=12)
    out2 = gs(cond2.cuda(), m2.cuda(), init2.cuda()).clone()
    assert torch.equal(out1, eager) and bool(torch.isfinite(out1).all())
    assert torch.equal(out2, plan.sample(packed, sd, cond2.cuda(), m2.cuda(), init2.cuda(), None))
    assert not torch.equal(out1, out2)
    # the borrowed workspace may grow under the graph's feet: the instance keeps the buffer it captured with
    ws.get(ws.buf.numel() * 2, "cuda")
    assert torch.equal(gs(cond.cuda(), m.cuda(), init.cuda()), eager)


def test_sample_edm_graph_cache_and_capture_fallback(monkeypatch):
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    m = _module()
    sp = hparams(fx.CFG_P).sampler
    z = torch.zeros(2, 2, 32, 32).cuda()
    cond, mk, init, _ = fx.sampler_inputs("det_u", B=2)
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: init.to(t.device))
    graphed = m.sample_edm(z, cond.cuda(), mk.cuda(), sp)
    assert len(m._graphs) == 1 and isinstance(next(iter(m._graphs.values())), L.GraphedSampler)
    # a capture that fails (e.g. another thread's HIP call under a global capture mode) must not break the evaluation loop
    m2 = _module()
    m2.load_state_dict(m.state_dict())

    def boom(run, dev):
        raise RuntimeError("operation not permitted when stream is capturing")
    monkeypatch.setattr(L, "_capture", boom)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eager = m2.sample_edm(z, cond.cuda(), mk.cuda(), sp)
        again = m2.sample_edm(z, cond.cuda(), mk.cuda(), sp)          # remembered: no second capture attempt
    assert sum("capture" in str(x.message) for x in w) == 1
    assert list(m2._graphs.values()) == ["eager"] and torch.equal(eager, graphed) and torch.equal(again, graphed)
    monkeypatch.undo()
    # at most two graphs are kept per module
    for B in (1, 3, 4):
        c, k, i, _ = fx.sampler_inputs("det_u", B=B)
        m.sample_edm(torch.zeros(B, 2, 32, 32).cuda(), c.cuda(), k.cuda(), sp)
    assert len(m._graphs) == 2


def test_replaced_parameter_is_repacked():
    """packed_weights keys on the LIVE parameter objects: a new nn.Parameter put in by setattr must be used."""
    m = _module()
    net = m.model
    x, cond = fx.randn("unet_P/x", 2, 2, 32, 32).cuda(), fx.randn("unet_P/cond", 2, 2, 32, 32).cuda()
    lab = torch.tensor([0.3]).cuda()
    with torch.no_grad():
        net.out_conv.weight.copy_(fx.param("e2e/out", "conv.weight", tuple(net.out_conv.weight.shape)))
        y0 = net(x, lab, cond).clone()
        net.out_conv.weight = torch.nn.Parameter(net.out_conv.weight.detach() * 2.0)        # a NEW object
        y1 = net(x, lab, cond)
        b = net.out_conv.bias.detach().view(1, -1, 1, 1)
    torch.testing.assert_close(y1 - b, 2.0 * (y0 - b), rtol=1e-4, atol=1e-5)
