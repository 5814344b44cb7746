"""CPU, world_size 2 over gloo: the data-parallel host logic of the N > 1 path, driven through the SAME host code the
GPU trainer runs (mcedm_amd.train.GradSync and clip_adam_ema_): batch sharding, the bucketed sum all-reduce of the flat
gradient buffer in the backward's completion order, and 1/world scaling + clip-after-average + Adam + EMA, must equal
one full-batch step (Lightning: DDP mean -> clip_grad_norm_(1.0) -> Adam.step -> EmaModel.update).  The per-shard
gradients come from the oracle and the two device functions (squared norm, fused Adam/EMA) are replaced by their CPU
restatements; the HIP kernels themselves are covered by the -m gpu tests."""
import math
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    from mcedm_amd.train import GradSync, clip_adam_ema_, shard_range
    from oracle import fixtures as fx
    from oracle import mcedm_oracle as orc
    cfg = orc.UNetConfig(ch=32, ch_mult=(1, 1), attn_resolutions=(), resolution=16)
    P = orc.make_params(cfg, 5)
    names = [n for n, _ in orc.param_shapes(cfg)]
    B, H, W = 6, 8, 8
    x = fx.randn("par/x", B, 2, H, W) * 3          # large enough that the clip (max_norm 1) is active
    mk = torch.zeros(B, 2, H, W)
    mk[:, 1] = 1
    cond = x * (1 - mk) + fx.randn("par/c", B, 2, H, W) * mk
    noise, rnd = fx.randn("par/n", B, 2, H, W), fx.randn("par/r", B, 1, 1, 1)

    def grads_of(lo, hi):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        orc.training_loss(Pg, cfg, x[lo:hi], cond[lo:hi], mk[lo:hi], noise[lo:hi], rnd[lo:hi]).backward()
        return torch.cat([Pg[n].grad.reshape(-1) for n in names])

    # the plan's bucket split (host-only C call) and the product's reducer on CPU tensors
    plan = L.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks,
                  cfg.attn_resolutions, cfg.resolution)
    assert plan.param_names == names
    firsts = plan.grad_buckets(3)
    lo, hi = shard_range(B, rank, world)
    flat_g = grads_of(lo, hi)
    sync = GradSync(flat_g, [P[n].numel() for n in names], firsts)
    assert sorted(sync.ranges) == sorted(set(sync.ranges)) and sum(b - a for a, b in sync.ranges) == flat_g.numel()
    sync.launch()
    sync.join()
    full = grads_of(0, B)
    err_sum = float((flat_g / world - full).abs().max() / full.abs().max())

    # clip + Adam + EMA through the product's host function, device functions replaced by CPU restatements
    flat_p = torch.cat([P[n].reshape(-1) for n in names])
    st = dict(p=flat_p.clone(), m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p), e=flat_p.clone())
    seen = {}

    def sqnorm_fn(g, sq):
        sq[0] = float((g.double() ** 2).sum())

    def adam_fn(p, g, m, v, e, step, sqnorm_t, grad_scale, lr, beta1, beta2, eps, weight_decay, max_norm, ema_beta):
        total = math.sqrt(float(sqnorm_t)) * grad_scale               # norm of the AVERAGED gradient
        clip = min(1.0, max_norm / (total + 1e-6))
        seen["clip"] = clip
        p1, m1, v1, e1 = orc.adam_ema_step(p, g * grad_scale, m, v, e, step, lr, beta1, beta2, eps, clip, ema_beta)
        p.copy_(p1); m.copy_(m1); v.copy_(v1); e.copy_(e1)

    hp = dict(lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, max_norm=1.0, ema_beta=0.999)
    sq = torch.zeros(1, dtype=torch.float64)
    clip_adam_ema_(st["p"], flat_g, st["m"], st["v"], st["e"], 1, world, hp, sq, sqnorm_fn=sqnorm_fn, adam_fn=adam_fn)
    coef, _ = orc.clip_scale([full], 1.0)
    p_ref, _, _, e_ref = orc.adam_ema_step(flat_p, full, torch.zeros_like(flat_p), torch.zeros_like(flat_p), flat_p, 1, clip=coef)
    err_p = float((st["p"] - p_ref).abs().max())
    err_e = float((st["e"] - e_ref).abs().max())
    if rank == 0:
        out.put((err_sum, (lo, hi), firsts, seen["clip"], coef, err_p, err_e))
    dist.destroy_process_group()


def test_shard_ranges_cover_the_batch():
    sys.path.insert(0, ROOT)
    import mcedm_amd  # noqa: F401
    from mcedm_amd.train import shard_range
    for n in (1, 7, 8, 256):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_two_rank_bucketed_allreduce_clip_adam_equals_full_batch_step():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    err, rng, firsts, clip, clip_ref, err_p, err_e = out.get(timeout=5)
    assert rng == (0, 3) and err < 1e-5, (err, rng)
    assert len(firsts) >= 2 and firsts[-1] == 0 and all(a > b for a, b in zip(firsts, firsts[1:])), firsts
    assert clip_ref < 0.9 and abs(clip - clip_ref) < 1e-5 * clip_ref, (clip, clip_ref)     # the clip really was active
    assert err_p < 5e-7 and err_e < 5e-9, (err_p, err_e)       # one fp32 ulp of O(1) parameters


def test_fused_trainer_optimizer_state_round_trips_through_torch_adam():
    """Checkpoint / resume of the fused trainer: its Adam state exports as a torch.optim.Adam state_dict
    (what Lightning stores as checkpoint['optimizer_states'][0] for models/mcedm.py:139-161) that torch's own Adam loads, and
    the state of a torch Adam that has taken steps loads back into the flat buffers, hyper-parameters included."""
    sys.path.insert(0, ROOT)
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    from mcedm_amd.train import FlatTrainState
    from oracle import mcedm_oracle as orc
    cfg = orc.UNetConfig(ch=32, ch_mult=(1, 1), attn_resolutions=(), resolution=16)
    P = orc.make_params(cfg, 3)
    plan = L.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks, cfg.attn_resolutions,
                  cfg.resolution)
    ts = FlatTrainState(plan, P, lr=3e-4, beta1=0.8, eps=1e-7)
    assert ts.optimizer_state_dict()["state"] == {}                       # a fresh optimiser has no state yet
    g = torch.Generator().manual_seed(0)
    ts.flat_m.copy_(torch.randn(ts.flat_m.shape, generator=g))
    ts.flat_v.copy_(torch.rand(ts.flat_v.shape, generator=g))
    ts.step_count = 7
    sd = ts.optimizer_state_dict()
    params = [torch.nn.Parameter(P[n].clone()) for n in plan.param_names]
    opt = torch.optim.Adam(params, lr=1.0)
    opt.load_state_dict(sd)                                               # torch accepts it as its own
    grp = opt.param_groups[0]
    assert (grp["lr"], grp["betas"], grp["eps"], grp["weight_decay"]) == (3e-4, (0.8, 0.999), 1e-7, 0.0)
    off = 0
    for i, p in enumerate(params):
        st = opt.state[p]
        assert float(st["step"]) == 7
        torch.testing.assert_close(st["exp_avg"].reshape(-1), ts.flat_m[off:off + p.numel()], rtol=0, atol=0)
        torch.testing.assert_close(st["exp_avg_sq"].reshape(-1), ts.flat_v[off:off + p.numel()], rtol=0, atol=0)
        off += p.numel()
    # a torch Adam that has stepped -> the fused trainer
    for p in params:
        p.grad = torch.randn(p.shape, generator=g)
    opt.step()
    t2 = FlatTrainState(plan, P)
    t2.load_optimizer_state_dict(opt.state_dict())
    assert t2.step_count == 8 and t2.hp["lr"] == 3e-4 and t2.hp["beta1"] == 0.8
    torch.testing.assert_close(t2.flat_m, torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in params]), rtol=0, atol=0)
    torch.testing.assert_close(t2.flat_v, torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1) for p in params]), rtol=0, atol=0)
    bad = opt.state_dict()
    bad["state"][3]["step"] = torch.tensor(5.0)
    with pytest.raises(RuntimeError, match="disagree on the step"):
        t2.load_optimizer_state_dict(bad)


class _ItemwiseSampler:
    """Stand-in for a module's sample_edm on the CPU: a deterministic per-item map of (cond, mask) with the reference's return
    layout 'b t h w c' float64 (models/mcedm.py:636-638).  What the test pins is the sharding and the gather order, not the sampler
    (whose batch-invariance on the device is tests/test_hip_fullsize.py's subject)."""

    def __init__(self):
        self.calls = []

    def sample_edm(self, hu, cond, hu_mask, sparams, return_last=True, guide_dx=False):
        self.calls.append(tuple(hu.shape))
        x = (cond[:, :2].double() * (1 - hu_mask.double()) + torch.sin(3.0 * cond[:, :2].double()) * hu_mask.double())
        xs = x.permute(0, 2, 3, 1).unsqueeze(1)                           # b 1 h w c
        return xs if return_last else torch.cat([xs * 0.5, xs], dim=1)


def _shard_worker(rank, world, port, out, n, return_last):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mcedm_amd  # noqa: F401
    from mcedm_amd.train import sample_edm_sharded, shard_range
    g = torch.Generator().manual_seed(11)
    cond = torch.randn(n, 2, 6, 4, generator=g)
    mask = (torch.rand(n, 2, 6, 4, generator=g) > 0.5).float()
    hu = torch.zeros(n, 2, 6, 4)
    mod = _ItemwiseSampler()
    got = sample_edm_sharded(mod, hu, cond, mask, None, return_last=return_last)
    own = sample_edm_sharded(mod, hu, cond, mask, None, return_last=return_last, gather=False)
    lo, hi = shard_range(n, rank, world)
    ref = _ItemwiseSampler().sample_edm(hu, cond, mask, None, return_last=return_last)
    ok = torch.equal(got, ref) and got.dtype == torch.float64
    ok_own = (own is None and hi == lo) or (own is not None and torch.equal(own, ref[lo:hi]))
    out.put((rank, ok, ok_own, mod.calls, (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,return_last", [(6, True), (5, False), (1, True)])
def test_sharded_sampling_gathers_in_the_callers_order(n, return_last):
    """SURVEY.md 8e / models/mcedm.py:356-385: the (n_samples * B) axis split over two ranks and gathered back must equal the
    unsharded call bit for bit, in the '(n b)' order test_step reshapes -- even shards (6), ragged ones (5 = 3 + 2) and a rank
    without any item (1 = 1 + 0)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 31500 + (os.getpid() + 7 * n) % 2000
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, out, n, return_last)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(out.get(timeout=5) for _ in range(2))
    for rank, ok, ok_own, calls, (lo, hi) in res:
        assert ok, f"rank {rank}: gathered states differ from the unsharded call"
        assert ok_own, f"rank {rank}: gather=False must return the rank's own shard"
        assert all(c[0] == hi - lo for c in calls) and (len(calls) == (2 if hi > lo else 0)), (rank, calls, lo, hi)
    assert res[0][4][0] == 0 and res[0][4][1] == res[1][4][0] and res[1][4][1] == n
