"""GPU, the N > 1 training path rehearsed on ONE card: two ranks (two processes, both on cuda:0, process group over gloo --
RCCL refuses two ranks on one device) run the data-parallel step of mcedm_amd.train.FlatTrainState on their halves of a batch:
the REAL multi-bucket backward (mcedm_edm_denoise_backward_dx records one HIP event per gradient bucket while it runs), the
side stream that waits for each event and all-reduces that bucket's slice of the flat gradient buffer, the join, and the fused
clip + Adam + EMA with the 1 / world factor.  The result must equal one single-process step on the full batch (Lightning:
DDP mean -> clip_grad_norm_(1.0) -> Adam.step -> EmaModel.update).  What this cannot cover is RCCL itself; everything around
the collective call is the code the 8-GPU run executes."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs():
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    xc, cond_in, mc = fx.training_nchw(h, u, mask, cond_noise)
    return xc, cond_in, mc, noise, rnd_normal.reshape(-1)


def _state(lo, hi, steps=2):
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    from mcedm_amd.train import FlatTrainState
    cfg = fx.CFG_P
    plan = L.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks, cfg.attn_resolutions,
                  cfg.resolution)
    P = {k: v.cuda() for k, v in orc.make_params(cfg, 7).items()}
    ts = FlatTrainState(plan, P, ema={k: v.clone() for k, v in P.items()}, max_buckets=4)
    ins = [t[lo:hi].contiguous().cuda() for t in _inputs()]
    for _ in range(steps):
        loss = ts.step(*ins)
    torch.cuda.synchronize()
    return ts, float(loss)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcedm_amd.train import shard_range
    lo, hi = shard_range(4, rank, world)
    ts, loss = _state(lo, hi)
    assert ts.world == 2 and len(ts.sync.ranges) > 1, "the bucketed, overlapped path must be the one that ran"
    torch.save({"p": ts.flat_p.cpu(), "ema": ts.flat_ema.cpu(), "m": ts.flat_m.cpu(), "loss": loss, "buckets": len(ts.sync.ranges)},
               f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_card_equal_the_full_batch_step():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r")
        mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
        r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["buckets"] >= 2
    for k in ("p", "ema", "m"):
        assert torch.equal(r0[k], r1[k]), f"replicas diverged in {k}"          # identical replicas after the exchange
    full, loss = _state(0, 4)
    # per-rank losses are shard means; the full-batch loss is their mean
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - loss) <= 1e-5 * abs(loss)
    scale = float(full.flat_p.abs().max())
    torch.testing.assert_close(r0["p"], full.flat_p.cpu(), rtol=1e-4, atol=1e-6 * scale)
    torch.testing.assert_close(r0["ema"], full.flat_ema.cpu(), rtol=1e-4, atol=1e-6 * scale)
    torch.testing.assert_close(r0["m"], full.flat_m.cpu(), rtol=1e-3, atol=1e-6 * float(full.flat_m.abs().max()))


def _module():
    import mcedm_amd  # noqa: F401
    from mcedm_amd.mcedm import PlMcedm
    from tests.test_hip_module import hparams
    m = PlMcedm(hparams(fx.CFG_P)).cuda()
    P = orc.make_params(fx.CFG_P, 7)
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    return m


def _sampling_inputs(n):
    cond, mask, init, _ = fx.sampler_inputs("det_u", n, 32, 32)
    return cond.cuda(), mask.cuda(), init.float().cuda()


def _sample_worker(rank, world, port, out, n):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcedm_amd.train import sample_edm_sharded, shard_range
    m = _module()
    cond, mask, init = _sampling_inputs(n)
    lo, hi = shard_range(n, rank, world)
    # the sampler's own draw (torch.randn_like(hu), models/mcedm.py:576) replaced by this rank's slice of ONE fixed noise tensor, so
    # that the sharded and the unsharded call integrate the same initial states
    real = torch.randn_like
    torch.randn_like = lambda t, **k: init[lo:hi].to(k.get("dtype", t.dtype))
    try:
        xs = sample_edm_sharded(m, torch.zeros_like(cond[:, :2]), cond, mask, m.sparams, return_last=True)
    finally:
        torch.randn_like = real
    torch.save(xs.cpu(), f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sampling_on_two_ranks_equals_the_unsharded_call_bit_for_bit():
    """SURVEY.md 8e, models/mcedm.py:356-385: PlMcedm.sample_edm with the batch axis split over two ranks (both on cuda:0, gloo) and
    one all_gather of the final float64 states: every rank ends up with the unsharded call's tensor, bit for bit and in its order
    (3 = 2 + 1 items: ragged shards)."""
    import socket
    n = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "s")
        mp.spawn(_sample_worker, args=(2, port, out, n), nprocs=2, join=True)
        r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    m = _module()
    cond, mask, init = _sampling_inputs(n)
    real = torch.randn_like
    torch.randn_like = lambda t, **k: init.to(k.get("dtype", t.dtype))
    try:
        full = m.sample_edm(torch.zeros_like(cond[:, :2]), cond, mask, m.sparams, return_last=True).cpu()
    finally:
        torch.randn_like = real
    assert full.dtype == torch.float64 and tuple(full.shape) == (n, 1, 32, 32, 2)
    assert torch.equal(r0, r1), "the ranks disagree on the gathered tensor"
    assert torch.equal(r0, full), f"sharded + gathered differs from the unsharded call: max |d| {float((r0 - full).abs().max()):.3e}"
