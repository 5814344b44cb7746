"""GPU, BASELINE.json full sizes (S128: 128x128 fields, ch=128, 4 levels, attention at 16^2, 32 states per GPU): the CPU
oracle needs ~20 s per state here, so parity is checked through size-independent properties of the path instead.

  * sampling is independent per batch item: one B=32 call == two B=16 calls, bit for bit (this is what makes the
    multi-GPU batch sharding of SURVEY.md 8e exact);
  * the forward path has no atomics: repeated calls are bitwise identical;
  * observed entries (mask == 0) come out exactly equal to the conditioning values (mcedm.py:597,618,628);
  * the fused conv is linear in its input and the network output is finite and non-trivial;
  * one oracle spot check on a single state of the same size for the whole 18-step sampler.
"""
import pytest
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc

from tests._tol import close_per_entry

pytestmark = pytest.mark.gpu

CFG = orc.UNetConfig(ch=128, ch_mult=(1, 1, 1, 1), attn_resolutions=(16,))
H = W = 128


@pytest.fixture(scope="module")
def net():
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib
    assert torch.cuda.is_available()
    plan = lib.Plan(CFG.in_channels, CFG.cond_channels, CFG.out_ch, CFG.ch, CFG.ch_mult, CFG.num_res_blocks,
                    CFG.attn_resolutions, CFG.resolution)
    P = orc.make_params(CFG, 7)
    packed = plan.pack({k: v.cuda() for k, v in P.items()})
    return lib, plan, packed, P


def inputs(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    state = torch.randn(B, 2, H, W, generator=g)
    m = fx.task_mask("h_time", B, H, W)
    cond = state * (1 - m) + torch.randn(B, 2, H, W, generator=g) * m
    init = torch.randn(B, 2, H, W, generator=g)
    return cond, m, init


def test_sampler_batch_shard_invariance_determinism_and_observed_entries(net):
    lib, plan, packed, _ = net
    B = 32
    cond, m, init = inputs(B)
    sd = lib.sampler_desc(orc.SamplerParams(timesteps=18))
    ws = lib.Workspace()
    full = plan.sample(packed, sd, cond.cuda(), m.cuda(), init.cuda(), None, ws=ws)
    again = plan.sample(packed, sd, cond.cuda(), m.cuda(), init.cuda(), None, ws=ws)
    assert torch.equal(full, again), "forward path must be bitwise reproducible"
    halves = [plan.sample(packed, sd, cond[s].contiguous().cuda(), m[s].contiguous().cuda(), init[s].contiguous().cuda(), None)
              for s in (slice(0, 16), slice(16, 32))]
    assert torch.equal(full, torch.cat(halves)), "batch items must be independent (exact shardability)"
    assert torch.isfinite(full).all() and full.dtype == torch.float64 and tuple(full.shape) == (B, 1, H, W, 2)
    obs = (m == 0).permute(0, 2, 3, 1)
    assert torch.equal(full[:, 0].cpu()[obs], cond.permute(0, 2, 3, 1).double()[obs])
    assert float(full[:, 0].cpu()[~obs].std()) > 1e-3, "sampled entries must not be degenerate"


def test_eight_wave_conv_kernel_is_bit_identical(net):
    # the experimental 8-wave kernel (fused GroupNorm partial sums included) must not change a single bit
    lib, plan, packed, _ = net
    B = 8
    cond, m, init = inputs(B, seed=3)
    sd = lib.sampler_desc(orc.SamplerParams(timesteps=2))
    base = plan.sample(packed, sd, cond.cuda(), m.cuda(), init.cuda(), None)
    lib.set_conv8(1)
    try:
        eight = plan.sample(packed, sd, cond.cuda(), m.cuda(), init.cuda(), None)
    finally:
        lib.set_conv8(-1)
    assert torch.equal(base, eight)


def test_conv_linearity_full_size(net):
    lib, *_ = net
    g = torch.Generator().manual_seed(1)
    B, C = 8, 128
    a = torch.randn(B, C, H, W, generator=g).cuda()
    b = torch.randn(B, C, H, W, generator=g).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda()
    wpk, _ = lib.op_pack_conv(w, None)
    ya, yb = lib.op_conv(a, None, wpk, None, C, 3), lib.op_conv(b, None, wpk, None, C, 3)
    yab = lib.op_conv(a + 2 * b, None, wpk, None, C, 3)
    torch.testing.assert_close(yab, ya + 2 * yb, rtol=1e-4, atol=1e-4)


def test_three_states_against_oracle_full_size(net):
    """The whole 18-step sampler (35 evaluations of the S128 network, Winograd kernels included) against the oracle on three
    states with three different masks (one state was thin for the kernel that is 75 % of the run time):
    'h_time' (h missing, u missing for t >= H / 2), 'u' (h observed) and nothing observed at all."""
    lib, plan, packed, P = net
    g = torch.Generator().manual_seed(3)
    state = torch.randn(3, 2, H, W, generator=g)
    m = torch.cat([fx.task_mask("h_time", 1, H, W), fx.task_mask("u", 1, H, W), torch.ones(1, 2, H, W)])
    cond = state * (1 - m) + torch.randn(3, 2, H, W, generator=g) * m
    init = torch.randn(3, 2, H, W, generator=g)
    sd = lib.sampler_desc(orc.SamplerParams(timesteps=18))
    xs = plan.sample(packed, sd, cond.cuda(), m.cuda(), init.cuda(), None)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    with torch.no_grad():
        ref = orc.sample_edm(P, CFG, cond, m, orc.SamplerParams(timesteps=18), init)
    for i, kind in enumerate(("h_time", "u", "none observed")):
        print(f"S128 state {i} ({kind}): max|d| = {float((xs[i].cpu() - ref[i]).abs().max()):.3e} on max|x| = {float(ref[i].abs().max()):.2f}")
    torch.testing.assert_close(xs.cpu(), ref, rtol=1e-4, atol=1e-5)
    obs = (m == 0).permute(0, 2, 3, 1)
    assert torch.equal(xs[:, 0].cpu()[obs], cond.permute(0, 2, 3, 1).double()[obs])


# ---- BASELINE config 5 at full size: the DDPM U-Net (configs/model/ddim_res32.yaml at 128 x 128: levels 128 / 64 / 32,
# single-head attention over T = 1024 tokens at 32^2, stride-2 convs 128 -> 64 -> 32) under the RePaint sampler ------------
from oracle import ddpm_oracle as dorc  # noqa: E402

CFG_D128 = dorc.DdpmConfig()            # resolution 128, ch 64, ch_mult (1, 1, 1), attn_resolutions (32,), self_cond


def test_winograd_and_direct_kernels_agree_full_size(net):
    """The S128 network with the Winograd F(2x2, 3x3) kernels (default: 26 of its 34 3x3 launches, csrc/conv_wino.hip) against
    the same network on the direct kernels only: one forward and one whole 18-step sampler call (35 evaluations), at the
    north_star tolerance; and the direct-kernel run is the one the oracle spot check above pins from the other side."""
    lib, plan, packed, _ = net
    B = 2
    cond, m, init = inputs(B, seed=3)
    sd = lib.sampler_desc(orc.SamplerParams(timesteps=18))
    lab = torch.tensor([0.7]).cuda()
    names = []
    lib.prof_enable(True)
    try:
        Fw = plan.forward(packed, init.cuda(), lab, cond=cond.cuda())
        torch.cuda.synchronize()
        names = [r["name"] for r in lib.prof_report()]
    finally:
        lib.prof_enable(False)
    assert any(n.startswith("conv_wino_kernel<WinoCfg<4>") for n in names), names
    xw = plan.sample(packed, sd, cond.cuda(), m.cuda(), init.cuda(), None)
    lib.set_conv_wino(0)
    try:
        lib.prof_enable(True)
        Fd = plan.forward(packed, init.cuda(), lab, cond=cond.cuda())
        torch.cuda.synchronize()
        names = [r["name"] for r in lib.prof_report()]
        lib.prof_enable(False)
        assert not any(n.startswith("conv_wino") for n in names), names
        xd = plan.sample(packed, sd, cond.cuda(), m.cuda(), init.cuda(), None)
    finally:
        lib.prof_enable(False)
        lib.set_conv_wino(-1)
    err = (Fw - Fd).abs()
    assert bool((err <= 1e-5 + 1e-4 * Fd.abs()).all()), f"forward: max |winograd - direct| = {float(err.max()):.3e} at |F| <= {float(Fd.abs().max()):.3e}"
    scale = float(xd.abs().max())
    err = (xw - xd).abs()
    assert bool((err <= 1e-5 * max(1.0, scale) + 1e-4 * xd.abs()).all()), f"sampler: max |winograd - direct| = {float(err.max()):.3e} (scale {scale:.3e})"
    obs = (m == 0).permute(0, 2, 3, 1)
    assert torch.equal(xw[:, 0].cpu()[obs], xd[:, 0].cpu()[obs]), "observed entries are exact in both"


@pytest.fixture(scope="module")
def ddpm_net():
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib
    c = CFG_D128
    plan = lib.DdpmPlan(c.in_channels, c.out_ch, c.ch, c.ch_mult, c.num_res_blocks, c.attn_resolutions, c.resolution)
    P = dorc.make_params(c, 21)
    packed = plan.pack({k: v.cuda() for k, v in P.items()}, dorc.timestep_freqs(c.ch).cuda())
    return lib, plan, packed, P


def repaint_inputs(B, N, R, seed):
    g = torch.Generator().manual_seed(seed)
    hu = torch.randn(B, 2, H, W, generator=g)
    init = torch.randn(B, 2, H, W, generator=g)
    steps = torch.randn(N, B, 2, H, W, generator=g).double()
    reps = torch.randn(N, R - 1, B, 2, H, W, generator=g).double()
    return hu, init, steps, reps


def repaint_desc(lib, sp):
    betas = dorc.betas_of(CFG_D128)
    return lib.repaint_desc(sp, dorc.edm_steps_of(betas), dorc.alphas_ext_of(betas), 1, 1)


def test_repaint_full_size_shard_invariance_determinism_and_known_entries(ddpm_net):
    """config 5's shape (n_time_h = 0, n_time_u = 64) with 3 resampling loops and churn: one B = 8 call == two B = 4 calls
    bit for bit, reruns are bitwise identical, known rows come out as the clean data, the rest is finite and non-trivial."""
    lib, plan, packed, _ = ddpm_net
    B, N, R = 8, 4, 3
    sp = dorc.RepaintParams(timesteps=N, n_repeat=R, S_churn=15.0, n_time_h=0, n_time_u=64)
    rd, keep = repaint_desc(lib, sp)
    hu, init, steps, reps = repaint_inputs(B, N, R, 0)
    ws = lib.Workspace()
    run = lambda s: plan.repaint_sample(packed, rd, hu[s].contiguous().cuda(), init[s].contiguous().cuda(),
                                        steps[:, s].contiguous().cuda(), reps[:, :, s].contiguous().cuda(), return_last=False,
                                        ws=ws)
    full = run(slice(0, B))
    assert torch.equal(full, run(slice(0, B))), "the RePaint path must be bitwise reproducible"
    halves = torch.cat([run(slice(0, 4)), run(slice(4, 8))])
    assert torch.equal(full, halves), "batch items must be independent (exact shardability across GPUs)"
    assert full.dtype == torch.float64 and tuple(full.shape) == (B, N + 1, H, W, 2) and bool(torch.isfinite(full).all())
    last = full[:, -1].cpu()
    assert torch.equal(last[:, :64, :, 1], hu[:, 1, :64].double()), "known rows of u must be the clean data"
    assert float((last[:, :, :, 0] - hu[:, 0].double()).abs().max()) > 1e-3, "h is generated, not copied"
    assert float(last[:, 64:, :, 1].std()) > 1e-3


def test_repaint_one_state_against_oracle_full_size(ddpm_net):
    """one state, 4 steps x 2 resampling loops (14 evaluations of the 128 x 128 network) against the CPU oracle."""
    lib, plan, packed, P = ddpm_net
    N, R = 4, 2
    sp = dorc.RepaintParams(timesteps=N, n_repeat=R, S_churn=0.0, n_time_h=0, n_time_u=64)
    rd, keep = repaint_desc(lib, sp)
    hu, init, steps, reps = repaint_inputs(1, N, R, 4)
    xs = plan.repaint_sample(packed, rd, hu.cuda(), init.cuda(), None, reps.cuda(), return_last=False)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    with torch.no_grad():
        ref = dorc.sample_edm_repaint(P, CFG_D128, hu, sp, init, [s for s in steps], [[r for r in rr] for rr in reps],
                                      return_last=False)
    scale = float(ref.abs().max())
    print(f"repaint 128^2 one state: max|d| = {float((xs.cpu() - ref).abs().max()):.3e} on max|x| = {scale:.1f}")
    close_per_entry(xs, ref, what="repaint 128^2 one state")      # each step's state against its own magnitude


def test_ddpm_forward_full_size_against_oracle(ddpm_net):
    """Model.forward at 128 x 128 (T = 1024 single-head attention, stride-2 convs) for two samples."""
    lib, plan, packed, P = ddpm_net
    x = fx.randn("fullsize/ddpm/x", 2, 2, H, W)
    y = plan.forward(packed, x.cuda(), 417.0)
    with torch.no_grad():
        ref = dorc.model_forward(P, CFG_D128, x, torch.tensor([417.0]))
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-5 * max(1.0, float(ref.abs().max())))


@pytest.mark.parametrize("mode", ["cat", "enc"])
def test_dx_conditioned_network_at_full_size(mode):
    """dx_cond (second clause) at the BASELINE config-4 size (single-task Darcy model: 1 + 1 -> 1 channels,
    ch = 128, 128 x 128): the dx-conditioned sampler is reproducible bit for bit; dx = zeros is dx = None exactly in the cat_dx
    form and differs from it in the dx_enc form (zero FEATURES, not dx_enc(0)); the training backward returns finite gradients
    for every parameter, non-zero for the dx head.  (NOT batch-shard invariant, by the reference's own definition: dx is the
    gradient of the residual's MEAN over the whole batch, models/pde_loss.py:231-236, so it scales with 1 / B -- a sharded call
    sees twice the dx of the unsharded one, exactly like one DDP rank of the reference; checked below.)"""
    import dataclasses
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib
    cfg = dataclasses.replace(CFG, in_channels=1, cond_channels=1, out_ch=1, dx_channels=1, dx_mode=mode)
    plan = lib.Plan(1, 1, 1, cfg.ch, cfg.ch_mult, cfg.num_res_blocks, cfg.attn_resolutions, cfg.resolution, dx_channels=1,
                    dx_mode=lib.DX_CAT if mode == "cat" else lib.DX_ENC)
    P = {k: v.cuda() for k, v in orc.make_params(cfg, 11).items()}
    assert plan.param_names == list(P)
    packed = plan.pack(P)
    B = 8
    g = torch.Generator().manual_seed(5)
    cond = (torch.randn(B, 1, H, W, generator=g) * 0.3).cuda()
    init = torch.randn(B, 1, H, W, generator=g).cuda()
    sd = lib.sampler_desc(orc.SamplerParams(timesteps=3))
    # Darcy residual of (a = cond * 0.05 + 1.4, u = x * 0.1): the statistics of fx.STEP_NORM_STATS
    gd = lib.GuidanceDesc(2, 0.0, 0.0, float(torch.tensor(2 * (1.0 / H), dtype=torch.float32)), 1.4, 0.05, 0.0, 0.1, 5.0)
    full = plan.sample(packed, sd, cond, None, init, None, dx_input=gd)
    again = plan.sample(packed, sd, cond, None, init, None, dx_input=gd)
    halves = [plan.sample(packed, sd, cond[s].contiguous(), None, init[s].contiguous(), None, dx_input=gd) for s in (slice(0, 4), slice(4, 8))]
    assert torch.equal(full, again) and bool(torch.isfinite(full).all())
    assert float((full - torch.cat(halves)).abs().max()) > 0 and all(bool(torch.isfinite(h_).all()) for h_ in halves)      # 1 / B in the batch-mean gradient
    plain = plan.sample(packed, sd, cond, None, init, None)                   # the same network without its dx input
    assert float((plain - full).abs().max()) > 0
    # network level: dx = zeros against dx = None
    x, sig = init * 0.5, torch.full((B,), 0.7, device="cuda")
    z = torch.zeros_like(x)
    D0, Dz = plan.denoise(packed, x, sig, cond=cond), plan.denoise(packed, x, sig, cond=cond, dx=z)
    assert torch.equal(D0, Dz) if mode == "cat" else float((D0 - Dz).abs().max()) > 1e-6
    # training: forward that keeps the activations + backward with dx
    dx = torch.randn(B, 1, H, W, generator=g).cuda() * 0.5
    ws = lib.Workspace()
    D = plan.denoise(packed, x, sig, cond=cond, ws=ws, training=True, dx=dx)
    assert torch.allclose(D, plan.denoise(packed, x, sig, cond=cond, dx=dx), rtol=1e-5, atol=1e-5)
    grads = [torch.full_like(P[n], float("nan")) for n in plan.param_names]
    plan.denoise_backward(packed, P, x, sig, cond, torch.ones_like(D), grads, ws=ws, dx=dx)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(g_).all()) for g_ in grads)
    if mode == "enc":
        for n, g_ in zip(plan.param_names, grads):
            if n.startswith(("dx_enc.", "combine_enc.")):
                assert float(g_.abs().max()) > 0, n


# ---- config 2 at ITS batch (64 states of 32 x 32 on one GPU), config 5 with ITS 32 resampling loops -----
def test_config2_batch_64_against_oracle():
    """BASELINE config 2 as bench.py runs it -- the reference's ch = 64 network on 64 states of 32 x 32, 18 Heun steps -- against
    the oracle on the SAME batch (every state, every mask kind), not on a 4- or 6-state stand-in: the kernels the batch of 64
    selects (one round of `WinoCfg<2>` workgroups at 32^2, the input-resident tiles below) are the ones compared."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib
    cfg = fx.CFG_P
    plan = lib.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks, cfg.attn_resolutions,
                    cfg.resolution)
    P = orc.make_params(cfg, 7)
    packed = plan.pack({k: v.cuda() for k, v in P.items()})
    B, S = 64, 32
    g = torch.Generator().manual_seed(64)
    state = torch.randn(B, 2, S, S, generator=g)
    kinds = ("u", "h", "h_time")
    m = torch.cat([fx.task_mask(kinds[i % 3], 1, S, S) for i in range(B)])
    cond = state * (1 - m) + torch.randn(B, 2, S, S, generator=g) * m
    init = torch.randn(B, 2, S, S, generator=g)
    sp = orc.SamplerParams(timesteps=18)
    xs = plan.sample(packed, lib.sampler_desc(sp), cond.cuda(), m.cuda(), init.cuda(), None)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    with torch.no_grad():
        ref = orc.sample_edm(P, cfg, cond, m, sp, init)
    print(f"config 2, 64 states: max|d| = {float((xs.cpu() - ref).abs().max()):.3e} on max|x| = {float(ref.abs().max()):.2f}")
    torch.testing.assert_close(xs.cpu(), ref, rtol=1e-4, atol=1e-5)
    obs = (m == 0).permute(0, 2, 3, 1)
    assert torch.equal(xs[:, 0].cpu()[obs], cond.permute(0, 2, 3, 1).double()[obs])


def test_repaint_with_all_32_resampling_loops_against_oracle():
    """BASELINE config 5's loop count -- n_repeat = 32 resampling loops per Heun step (README.md:60-61, configs/diff_sampler/
    edm_sampler_inv.yaml) -- on the DDPM U-Net at 32 x 32 so that the oracle finishes in seconds: 3 steps x 32 loops = 160
    evaluations, every intermediate state of the outer loop against its own magnitude."""
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib
    c = dorc.DdpmConfig(resolution=32, attn_resolutions=(8,))
    plan = lib.DdpmPlan(c.in_channels, c.out_ch, c.ch, c.ch_mult, c.num_res_blocks, c.attn_resolutions, c.resolution)
    P = dorc.make_params(c, 23)
    packed = plan.pack({k: v.cuda() for k, v in P.items()}, dorc.timestep_freqs(c.ch).cuda())
    B, N, R, S = 2, 3, 32, 32
    sp = dorc.RepaintParams(timesteps=N, n_repeat=R, S_churn=0.0, n_time_h=0, n_time_u=16)
    betas = dorc.betas_of(c)
    rd, keep = lib.repaint_desc(sp, dorc.edm_steps_of(betas), dorc.alphas_ext_of(betas), 1, 1)
    g = torch.Generator().manual_seed(32)
    hu = torch.randn(B, 2, S, S, generator=g)
    init = torch.randn(B, 2, S, S, generator=g)
    steps = torch.randn(N, B, 2, S, S, generator=g).double()
    reps = torch.randn(N, R - 1, B, 2, S, S, generator=g).double()
    xs = plan.repaint_sample(packed, rd, hu.cuda(), init.cuda(), None, reps.cuda(), return_last=False)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    with torch.no_grad():
        ref = dorc.sample_edm_repaint(P, c, hu, sp, init, [s for s in steps], [[r for r in rr] for rr in reps], return_last=False)
    print(f"repaint, 32 loops: max|d| = {float((xs.cpu() - ref).abs().max()):.3e} on max|x| = {float(ref.abs().max()):.1f}")
    close_per_entry(xs, ref, what="repaint with 32 resampling loops")
