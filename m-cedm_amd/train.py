"""Data-parallel EDM training step on flat parameter buffers.

One process per GPU.  Per step (models/mcedm.py:254-281 + Lightning's DDP / clip / Adam / EmaModel.update):

    x_noise, sigma      <- mcedm_edm_noise_inputs            (HIP)
    D                   <- mcedm_edm_denoise(training)        (HIP, activations kept in the workspace)
    loss, dD            <- mcedm_edm_loss                     (HIP)
    grads (flat)        <- mcedm_edm_denoise_backward_bucketed (HIP; records one event per gradient bucket)
    grads               <- sum all-reduce, bucket by bucket, on a side stream as soon as a bucket's event fires, i.e.
                           UNDER the rest of the backward (RCCL over xGMI through torch.distributed; gloo on CPU)
    |g|^2               <- mcedm_sqnorm                        (HIP, after the side stream is joined)
    params, m, v, ema   <- mcedm_adam_ema_step                 (HIP: 1/world scaling + clip + Adam + EMA fused)

The flat layout is the parameter order of ``DhariwalUNet.state_dict()``; each ``nn.Parameter`` of the model (and of the
EMA copy) is re-pointed to a view of the flat buffer, so ``state_dict()`` / checkpoints are unchanged.

Reproducibility: no kernel of the step uses atomics -- the weight-gradient kernel's split-K partial blocks are stored per
split and reduced in a fixed order (csrc/wgrad_mfma.hip) -- so the same inputs give the same bits, run after run
(tests/test_hip_backward.py::test_training_step_is_bitwise_reproducible); inference / sampling is additionally
bit-identical under batch sharding (tests/test_hip_fullsize.py).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import lib as _lib


def shard_range(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sample_edm_sharded(module, hu, cond, hu_mask, sparams, return_last=True, guide_dx=False, gather=True, group=None):
    """``module.sample_edm`` with the ``(n_samples * B)`` axis split over the ranks (BASELINE config 4 is
    ``n_samples = 16`` on 8 GPUs): rank r samples the contiguous items ``shard_range(n, r, world)`` -- batch items are
    independent and the sigma schedule is shared, so there is no collective inside the loop -- and, with ``gather=True``, one
    ``all_gather`` of the final ``[b, t, h, w, c]`` float64 states puts them back in the caller's order, i.e. the reference's
    ``(n b)`` layout that ``test_step`` reshapes to ``n b ...`` (models/mcedm.py:356-385).  Shards of unequal size (n not a
    multiple of the world size) are padded for the exchange; a rank without items contributes nothing.  With world size 1, or
    ``gather=False`` (the rank keeps its own shard), no communication happens.  Every rank must call it with the same n.

    The noise the sampler draws (``torch.randn_like`` inside ``sample_edm``) comes from each rank's own generator, like every DDP
    rank of the reference; a sample's bits do not depend on the batch it is computed in (tests/test_hip_fullsize.py), so with the
    same noise the gathered tensor equals the unsharded call bit for bit (tests/test_parallel_cpu.py, tests/test_hip_multirank.py)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    n = hu.shape[0]
    lo, hi = shard_range(n, rank, world)
    own = None
    if hi > lo:
        own = module.sample_edm(hu[lo:hi], cond[lo:hi], hu_mask[lo:hi], sparams, return_last=return_last, guide_dx=guide_dx)
    if world == 1 or not gather:
        return own
    # shape of one item's trajectory: from this rank's own result, else from a rank that has one (n < world leaves ranks empty)
    meta = torch.tensor(list(own.shape[1:]) if own is not None else [0, 0, 0, 0], dtype=torch.int64, device=hu.device)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    item = next((tuple(int(v) for v in m.tolist()) for m in metas if int(m.sum()) > 0), None)
    if item is None:
        raise RuntimeError("sample_edm_sharded: no rank has an item to sample (n = 0)")
    sizes = [shard_range(n, r, world) for r in range(world)]
    pad = max(b - a for a, b in sizes)
    send = torch.zeros((pad,) + item, dtype=torch.float64, device=hu.device)
    if own is not None:
        send[:hi - lo] = own.to(torch.float64)
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send, group=group)
    return torch.cat([p[:b - a] for p, (a, b) in zip(parts, sizes)], dim=0)


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_mean_(flat: torch.Tensor, world: Optional[int] = None, average: bool = True) -> torch.Tensor:
    """In-place sum all-reduce of one flat buffer (RCCL on GPUs, gloo on CPU), optionally divided by world."""
    if _world() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.div_(world or dist.get_world_size())
    return flat


def flatten_params_(params: Sequence[torch.nn.Parameter]) -> torch.Tensor:
    """Move the parameters into one contiguous fp32 buffer and re-point each one at its slice."""
    params = list(params)
    dev = params[0].device
    flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
    return flat


def views_like(flat: torch.Tensor, params: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    out, off = [], 0
    for p in params:
        out.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    return out


class GradSync:
    """Sum all-reduce of the flat gradient buffer in buckets that follow the backward's completion order.

    ``bucket_first`` (from ``Plan.grad_buckets``) are first-parameter indices, decreasing, ending with 0; bucket k is
    the element range of parameters [bucket_first[k], bucket_first[k-1]).  On a GPU, ``launch()`` makes a side stream
    wait for bucket k's event (recorded by the library inside the backward) and issues that bucket's all-reduce there,
    so the exchange of the decoder's gradients overlaps the encoder's backward; ``join()`` makes the caller's stream
    wait for the side stream.  On CPU tensors (gloo, tests) the same bucket ranges are reduced in order, synchronously.
    With world size 1 nothing is launched."""

    def __init__(self, flat_g: torch.Tensor, numels: Sequence[int], bucket_first: Sequence[int]):
        self.flat_g = flat_g
        offs = [0]
        for n in numels:
            offs.append(offs[-1] + int(n))
        self.bucket_first = [int(f) for f in bucket_first]
        if not self.bucket_first or self.bucket_first[-1] != 0 or any(a <= b for a, b in zip(self.bucket_first, self.bucket_first[1:])):
            raise RuntimeError("bucket_first must decrease and end with 0")
        his = [len(numels)] + self.bucket_first[:-1]
        self.ranges = [(offs[lo], offs[hi]) for lo, hi in zip(self.bucket_first, his)]      # element ranges, completion order
        self.world = _world()
        self.events: Optional[List[torch.cuda.Event]] = None
        self.side: Optional[torch.cuda.Stream] = None
        if flat_g.is_cuda:
            self.side = torch.cuda.Stream(device=flat_g.device)
            self.events = [torch.cuda.Event(enable_timing=False) for _ in self.ranges]
            for e in self.events:
                e.record()                 # torch creates the HIP event lazily; the library needs the handle

    def launch(self) -> None:
        if self.world == 1:
            return
        if self.side is None:
            for lo, hi in self.ranges:
                dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM)
            return
        for ev, (lo, hi) in zip(self.events, self.ranges):
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM)

    def join(self) -> None:
        if self.world > 1 and self.side is not None:
            torch.cuda.current_stream(self.flat_g.device).wait_stream(self.side)


def clip_adam_ema_(flat_p, flat_g, flat_m, flat_v, flat_ema, step: int, world: int, hp: Dict[str, float], sq: torch.Tensor,
                   sqnorm_fn: Callable = _lib.sqnorm, adam_fn: Callable = _lib.adam_ema_step) -> None:
    """What follows the all-reduce, as Lightning orders it (DDP mean -> clip_grad_norm_(1.0) -> Adam.step -> EmaModel.update):
    ``flat_g`` holds the SUM over ranks; the 1/world of DDP's mean is folded into the fused kernel as ``grad_scale`` and the
    clip factor is formed from the norm of the scaled gradient.  The two device functions are injectable so that the
    world_size-2 CPU test drives this very function with reference implementations."""
    sqnorm_fn(flat_g, sq)
    adam_fn(flat_p, flat_g, flat_m, flat_v, flat_ema, step, sqnorm_t=sq, grad_scale=1.0 / world, **hp)


class FlatTrainState:
    """Flat parameter / gradient / Adam / EMA buffers of one network and the fused optimisation step on them."""

    def __init__(self, plan: "_lib.Plan", params: Dict[str, torch.Tensor], packed: Optional[torch.Tensor] = None,
                 ema: Optional[Dict[str, torch.Tensor]] = None, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                 clip=1.0, ema_beta=0.999, P_mean=-1.2, P_std=1.2, sigma_data=1.0, max_buckets: int = 4):
        self.plan = plan
        names = plan.param_names
        tens = [params[n] for n in names]
        self.hp = dict(lr=lr, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, max_norm=clip, ema_beta=ema_beta)
        self.P_mean, self.P_std, self.sigma_data = P_mean, P_std, sigma_data
        self.flat_p = torch.cat([t.detach().reshape(-1) for t in tens]).contiguous()
        self.pviews = dict(zip(names, views_like(self.flat_p, tens)))
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.flat_ema = None
        if ema is not None:
            self.flat_ema = torch.cat([ema[n].detach().reshape(-1) for n in names]).contiguous()
        self.grad_views = views_like(self.flat_g, tens)
        self.sq = torch.zeros(1, dtype=torch.float64, device=self.flat_p.device)
        # scratch of the two fixed-order reductions (loss, gradient norm): owned by this state, so two trainers on two streams
        # of one device never share any (ABI 3; the calls of one step are ordered on its stream and may share one area)
        self.red_scratch = (torch.empty(_lib.REDUCE_SCRATCH_BYTES, dtype=torch.uint8, device=self.flat_p.device)
                            if self.flat_p.is_cuda else None)
        self.step_count = 0
        self.ws = _lib.Workspace()
        self.packed = packed
        self.world = _world()
        self.sync = GradSync(self.flat_g, [t.numel() for t in tens], plan.grad_buckets(max_buckets if self.world > 1 else 1))
        self.use_graph = os.environ.get("MCEDM_TRAIN_GRAPH", "1") != "0"
        self._graph = None

    def _forward_backward(self, x, cond_in, mask, noise, rnd_normal, dx_fn=None):
        """pack -> noise -> denoise(training) -> loss -> backward into the flat gradient buffer; returns the loss tensor.
        dx_fn (dx_cond models, models/ddim.py:1672-1681): callable (cond_in, x_noise) -> the network's dx input (no gradient)."""
        plan = self.plan
        self.packed = plan.pack(self.pviews, self.packed)
        x_noise, sigma = _lib.edm_noise_inputs(x, mask, noise, rnd_normal.reshape(-1).contiguous(), self.P_mean, self.P_std)
        dx = dx_fn(cond_in, x_noise) if dx_fn is not None else None
        D = plan.denoise(self.packed, x_noise, sigma, cond=cond_in, ws=self.ws, training=True, sigma_data=self.sigma_data, dx=dx)
        loss, dD = _lib.edm_loss(D, x, mask, sigma, sigma_data=self.sigma_data, want_grad=True, scratch=self.red_scratch)
        multi = self.world > 1             # bucket events only where a side stream waits for them
        plan.denoise_backward(self.packed, self.pviews, x_noise, sigma, cond_in, dD, self.grad_views, self.ws,
                              sigma_data=self.sigma_data, bucket_first=self.sync.bucket_first if multi else None,
                              bucket_events=self.sync.events if multi else None, dx=dx)
        return loss

    def _graphed_forward_backward(self, x, cond_in, mask, noise, rnd_normal):
        """Single-GPU steps replay the ~300 launches of the forward + backward from ONE HIP graph (the library never
        allocates or synchronises, every scalar of these kernels is shape-derived): inputs are copied into static buffers,
        the loss comes back in a static tensor.  Kept per input shape; a failed capture falls back to eager launches.
        (Multi-rank steps stay eager: the bucket events and the side-stream all-reduce are issued around the backward.)"""
        ins = (x, cond_in, mask, noise, rnd_normal)
        key = tuple((tuple(t.shape), t.dtype) if t is not None else None for t in ins)
        if self._graph is None or self._graph[0] != key:
            self._graph = None
            static = [torch.empty_like(t) if t is not None else None for t in ins]
            for d, t in zip(static, ins):
                if d is not None:
                    d.copy_(t)
            holder = {}

            def run():
                holder["loss"] = self._forward_backward(*static)
            try:
                self.packed = self.plan.pack(self.pviews, self.packed)         # allocated outside the capture
                self.ws.get(self.plan.workspace_bytes(x.shape[0], x.shape[2], x.shape[3], True), x.device)
                g = _lib._capture(run, x.device)
            except RuntimeError as e:                                          # pragma: no cover - needs a capture failure
                import warnings
                warnings.warn(f"HIP-graph capture of the training step failed ({str(e)[:200]}); launching it eagerly")
                torch.cuda.synchronize()
                self._graph = (key, None, None, None)
                return self._forward_backward(*ins)
            self._graph = (key, g, static, holder["loss"])
        _, g, static, loss = self._graph
        if g is None:
            return self._forward_backward(*ins)
        for d, t in zip(static, ins):
            if d is not None:
                d.copy_(t)
        g.replay()
        return loss

    # ---- checkpoint / resume (configs/callbacks/callbacks_ddim.yaml:1-10, run.py:68-72): the optimiser half -------------
    def optimizer_state_dict(self) -> dict:
        """The Adam state in ``torch.optim.Adam.state_dict()`` form -- what Lightning stores as
        ``checkpoint['optimizer_states'][0]`` for ``configure_optimizers`` (models/mcedm.py:139-161): per-parameter ``step``,
        ``exp_avg``, ``exp_avg_sq`` in ``model.parameters()`` order (= the plan's parameter table) and one param group.  A run
        of the fused trainer can therefore be resumed by the reference's Lightning loop and vice versa."""
        names = self.plan.param_names
        shapes = [self.pviews[n] for n in names]
        state = {}
        if self.step_count > 0:
            for i, (m, v) in enumerate(zip(views_like(self.flat_m, shapes), views_like(self.flat_v, shapes))):
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
        hp = self.hp
        group = {"lr": hp["lr"], "betas": (hp["beta1"], hp["beta2"]), "eps": hp["eps"], "weight_decay": hp["weight_decay"],
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        """Inverse of optimizer_state_dict: accepts the ``optimizer_states[0]`` entry of a reference checkpoint (one param
        group over ``model.parameters()``, no amsgrad).  Hyper-parameters of the group (lr, betas, eps, weight decay) are taken
        over; an empty state resets the moments."""
        names = self.plan.param_names
        groups = sd["param_groups"]
        if len(groups) != 1 or list(groups[0]["params"]) != list(range(len(names))):
            raise RuntimeError("optimizer state: expected one param group over model.parameters() in registration order")
        g = groups[0]
        if g.get("amsgrad", False):
            raise NotImplementedError("amsgrad state (the reference configures amsgrad=False, configs/model/*.yaml)")
        self.hp.update(lr=float(g["lr"]), beta1=float(g["betas"][0]), beta2=float(g["betas"][1]), eps=float(g["eps"]),
                       weight_decay=float(g["weight_decay"]))
        state = sd["state"]
        if not state:
            self.flat_m.zero_(); self.flat_v.zero_(); self.step_count = 0
            return
        if sorted(int(k) for k in state) != list(range(len(names))):
            raise RuntimeError("optimizer state: every parameter needs an entry")
        shapes = [self.pviews[n] for n in names]
        steps = set()
        with torch.no_grad():
            for i, (m, v) in enumerate(zip(views_like(self.flat_m, shapes), views_like(self.flat_v, shapes))):
                e = state[i] if i in state else state[str(i)]
                if tuple(e["exp_avg"].shape) != tuple(m.shape):
                    raise RuntimeError(f"optimizer state of parameter {names[i]}: shape {tuple(e['exp_avg'].shape)} != {tuple(m.shape)}")
                m.copy_(e["exp_avg"]); v.copy_(e["exp_avg_sq"])
                steps.add(int(float(e["step"])))
        if len(steps) != 1:
            raise RuntimeError(f"optimizer state: parameters disagree on the step count ({sorted(steps)})")
        self.step_count = steps.pop()

    def step(self, x, cond_in, mask, noise, rnd_normal, dx_fn=None, exchange: bool = True):
        """One optimisation step on this rank's shard (all tensors NCHW fp32 on the device). Returns the local loss (a
        tensor that the next step overwrites when the step is graph-replayed).  dx_fn: see _forward_backward (such steps are
        launched eagerly: the callable is the caller's code).  exchange=False skips the gradient all-reduce (measurement
        only -- bench.py separates compute from exchange with it; the replicas then drift apart)."""
        if self.use_graph and self.world == 1 and x.is_cuda and dx_fn is None:
            loss = self._graphed_forward_backward(x, cond_in, mask, noise, rnd_normal)
        else:
            loss = self._forward_backward(x, cond_in, mask, noise, rnd_normal, dx_fn)
        if exchange:
            self.sync.launch()             # bucketed sum all-reduce, overlapping the tail of the backward
            self.sync.join()
        self.step_count += 1
        sqn = _lib.sqnorm if self.red_scratch is None else (lambda g, out: _lib.sqnorm(g, out, scratch=self.red_scratch))
        clip_adam_ema_(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.flat_ema, self.step_count, self.world, self.hp,
                       self.sq, sqnorm_fn=sqn)
        return loss


class EdmTrainer(FlatTrainState):
    """Fused training step for a ``mcedm_amd.mcedm.PlMcedm`` (or anything with ``.model`` / ``.ema_model``): the module's
    parameters (and its EMA copy's) are re-pointed at the flat buffers, so the module sees every update."""

    def __init__(self, module, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, clip=1.0, ema_beta=0.999,
                 P_mean=-1.2, P_std=1.2, sigma_data=1.0, max_buckets: int = 4):
        self.module = module
        self.net = module.model
        net = self.net
        params = dict(net.named_parameters())
        ema_net = module.ema_model.ma_model if getattr(module, "ema_model", None) is not None else None
        ema = dict(ema_net.named_parameters()) if ema_net is not None else None
        super().__init__(net.plan, params, packed=None, ema=ema, lr=lr, beta1=beta1, beta2=beta2, eps=eps,
                         weight_decay=weight_decay, clip=clip, ema_beta=ema_beta, P_mean=P_mean, P_std=P_std,
                         sigma_data=sigma_data, max_buckets=max_buckets)
        with torch.no_grad():
            for n, p in params.items():
                p.data = self.pviews[n]
            if ema is not None:
                for p, v in zip(ema.values(), views_like(self.flat_ema, list(ema.values()))):
                    p.data = v

    def step(self, x, cond_in, mask, noise, rnd_normal, dx_fn=None):
        loss = super().step(x, cond_in, mask, noise, rnd_normal, dx_fn)
        # the kernel wrote parameters (and the EMA copy) in place behind autograd's back: drop the modules' packed copies
        self.net.invalidate_packed()
        if self.flat_ema is not None:
            self.module.ema_model.ma_model.invalidate_packed()
        return loss
