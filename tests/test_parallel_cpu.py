"""CPU, world_size 2 over gloo: the data-parallel host logic of the N > 1 path -- batch sharding and the single
flat-buffer gradient all-reduce -- gives the full-batch gradient (sum of per-shard gradients == full-batch gradient,
SURVEY.md section 4 item 4).  The per-shard gradients come from the oracle; the HIP kernels themselves are covered by the
-m gpu tests."""
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
    from mcedm_amd.train import allreduce_mean_, shard_range
    from oracle import fixtures as fx
    from oracle import mcedm_oracle as orc
    cfg = orc.UNetConfig(ch=32, ch_mult=(1, 1), attn_resolutions=(), resolution=16)
    P = orc.make_params(cfg, 5)
    B, H, W = 6, 8, 8
    x = fx.randn("par/x", B, 2, H, W)
    mk = torch.zeros(B, 2, H, W)
    mk[:, 1] = 1
    cond = x * (1 - mk) + fx.randn("par/c", B, 2, H, W) * mk
    noise, rnd = fx.randn("par/n", B, 2, H, W), fx.randn("par/r", B, 1, 1, 1)

    def grads_of(lo, hi):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        orc.training_loss(Pg, cfg, x[lo:hi], cond[lo:hi], mk[lo:hi], noise[lo:hi], rnd[lo:hi]).backward()
        return torch.cat([Pg[n].grad.reshape(-1) for n, _ in orc.param_shapes(cfg)])

    lo, hi = shard_range(B, rank, world)
    flat = grads_of(lo, hi)
    allreduce_mean_(flat)                       # ONE message: the flat gradient buffer
    full = grads_of(0, B)
    err = float((flat - full).abs().max() / full.abs().max())
    if rank == 0:
        out.put((err, (lo, hi)))
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


def test_two_rank_gradient_allreduce_equals_full_batch():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    err, rng = out.get(timeout=5)
    assert rng == (0, 3) and err < 1e-5, (err, rng)
