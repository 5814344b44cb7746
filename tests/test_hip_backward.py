"""GPU parity of the backward path (through the C ABI): weight/data gradients of the fused conv, GroupNorm/FiLM/SiLU
backward, attention backward, and the complete training step (loss + every parameter gradient) against autograd
through the oracle and against the golden vectors the reference produced (tests/golden/training_P.npz).

Tolerance: rtol 1e-4 with atol = 1e-5 x max|ref| per tensor (gradients span several orders of magnitude).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    assert torch.cuda.is_available()
    L.load()
    return L


def dev(t):
    return t.detach().contiguous().cuda()


def close(got, ref, rtol=1e-4, rel_atol=1e-5, what=""):
    got = got.detach().cpu().double()
    ref = torch.as_tensor(ref).detach().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs()
    lim = rel_atol * ref.abs().max() + rtol * ref.abs()
    bad = err > lim
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {err.max():.3e} (max ref {ref.abs().max():.3e})"


def apply_coef(x, coef, act):
    y = (x - coef[:, :, 0, None, None]) * coef[:, :, 1, None, None] + coef[:, :, 2, None, None]
    return F.silu(y) if act else y


def rand_coef(tag, B, C):
    return torch.stack([fx.randn(tag + "/mean", B, C) * 0.3, 1 + 0.3 * fx.randn(tag + "/scale", B, C),
                        0.2 * fx.randn(tag + "/off", B, C), torch.zeros(B, C)], dim=-1)


CONV_SHAPES = [
    # (B, Ca, Cb, Cout, H, W, k, resample, act)
    (2, 64, 0, 64, 32, 32, 3, 0, 1), (2, 72, 64, 128, 20, 24, 3, 0, 1), (2, 64, 0, 64, 16, 16, 3, 1, 1),
    (2, 64, 0, 64, 16, 16, 3, 2, 1), (3, 2, 2, 64, 32, 32, 3, 0, 0), (2, 64, 0, 2, 32, 32, 3, 0, 1),
    (2, 128, 0, 64, 16, 16, 1, 0, 0), (1, 8, 0, 8, 4, 4, 3, 0, 1), (2, 40, 0, 24, 8, 8, 1, 0, 1),
    (4, 128, 0, 128, 64, 64, 3, 0, 1),
    # thin shapes (wgrad_thin_kernel: at most 4 channels on one side -- conv_in, out_conv): ragged widths, 1 - 4 channels, concat
    (3, 64, 0, 1, 12, 20, 3, 0, 1), (2, 32, 32, 3, 9, 70, 3, 0, 0), (2, 128, 0, 4, 16, 130, 3, 0, 1), (2, 128, 0, 2, 128, 128, 3, 0, 1),
    (3, 1, 0, 32, 10, 24, 3, 0, 0), (2, 2, 1, 64, 7, 66, 3, 0, 1), (2, 2, 2, 128, 128, 128, 3, 0, 0),
    # power-of-two widths take wgrad_thin4_kernel (four pixels per lane): more row streams than rows, a single stream, odd heights
    (2, 64, 0, 3, 6, 16, 3, 0, 1), (1, 3, 0, 32, 5, 256, 3, 0, 0), (3, 32, 0, 4, 33, 64, 3, 0, 1),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv_wgrad_and_dgrad(lib, shape):
    B, Ca, Cb, Cout, H, W, k, rs, act = shape
    tag = "t/bwd/conv/" + "_".join(map(str, shape))
    Cin = Ca + Cb
    Hs, Ws = (H // 2, W // 2) if rs == 1 else ((H * 2, W * 2) if rs == 2 else (H, W))
    xa = fx.randn(tag + "/xa", B, Ca, Hs, Ws)
    xb = fx.randn(tag + "/xb", B, Cb, Hs, Ws) if Cb else None
    w = fx.param(tag, "conv.weight", (Cout, Cin, k, k)).requires_grad_(True)
    b = fx.param(tag, "conv.bias", (Cout,)).requires_grad_(True)
    coef = rand_coef(tag, B, Cin)
    dy = fx.randn(tag + "/dy", B, Cout, H, W)
    x = torch.cat([xa, xb], 1) if Cb else xa
    u = apply_coef(x, coef, act)
    if rs == 1:
        u = orc.resample_up(u)
    elif rs == 2:
        u = orc.resample_down(u)
    u = u.detach().requires_grad_(True)
    y = F.conv2d(u, w, b, padding=k // 2)
    gu, gw, gb = torch.autograd.grad(y, (u, w, b), dy)
    dw, db = lib.op_conv_wgrad(dev(dy), dev(xa), dev(xb) if Cb else None, k, coef=dev(coef), act=act, resample=rs)
    close(dw, gw, what="dW")
    close(db, gb, what="db")
    wpk, _ = lib.op_pack_conv(dev(w), None, dgrad=True)
    du = lib.op_conv(dev(dy), None, wpk, None, Cin, k)
    close(du, gu, what="d(conv input)")


@pytest.mark.parametrize("B,Ca,Cb,Cout,H,W,use_coef", [
    (4, 128, 0, 128, 128, 128, False),      # one tensor read in place; 1024 stages = 4 per split
    (2, 128, 128, 128, 128, 128, False),    # the decoder's skip projection: cat(x, skip) read in place from two tensors
    (5, 128, 0, 256, 64, 128, True),        # a transformed (materialised) operand, two output-channel blocks, ragged splits
])
def test_wgrad_1x1_gemm_form(lib, B, Ca, Cb, Cout, H, W, use_coef):
    """wgrad_gemm1_kernel (round 5): the 1x1 weight gradient of 128-channel blocks as a GEMM on the Winograd weight-gradient kernel's
    stage machinery, its slices finished by the direct kernel's reduction.  Against fp64 autograd, bit-reproducible."""
    tag = f"t/bwd/gemm1/{B}_{Ca}_{Cb}_{Cout}_{H}_{W}"
    Cin = Ca + Cb
    xa = fx.randn(tag + "/xa", B, Ca, H, W)
    xb = fx.randn(tag + "/xb", B, Cb, H, W) if Cb else None
    dy = fx.randn(tag + "/dy", B, Cout, H, W)
    coef = torch.stack([fx.randn(tag + "/m", B, Cin) * 0.3, 1 + 0.3 * fx.randn(tag + "/s", B, Cin), 0.2 * fx.randn(tag + "/o", B, Cin),
                        torch.zeros(B, Cin)], dim=-1) if use_coef else None
    x = (torch.cat([xa, xb], 1) if Cb else xa).double()
    if use_coef:
        c = coef.double()
        x = F.silu((x - c[:, :, 0, None, None]) * c[:, :, 1, None, None] + c[:, :, 2, None, None])
    w = torch.zeros(Cout, Cin, 1, 1, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    gw, gb = torch.autograd.grad(F.conv2d(x, w, b), (w, b), dy.double())
    kw = dict(coef=dev(coef), act=1) if use_coef else {}
    lib.prof_enable(True)
    try:
        dw, db = lib.op_conv_wgrad(dev(dy), dev(xa), dev(xb) if Cb else None, 1, **kw)
        torch.cuda.synchronize()
        names = {r["name"] for r in lib.prof_report()}
    finally:
        lib.prof_enable(False)
    assert "wgrad_gemm1_kernel" in names and not any(n.startswith("wgrad_kernel") for n in names), names
    assert ("act_materialize_kernel" in names) == use_coef, names
    close(dw, gw, what="dW")
    close(db, gb, what="db")
    dw2, db2 = lib.op_conv_wgrad(dev(dy), dev(xa), dev(xb) if Cb else None, 1, **kw)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("B,H,W,gemm", [(8, 64, 64, True), (2, 16, 16, False)])
def test_qkv_weight_gradient_rows_in_packed_order(lib, B, H, W, gemm):
    """The attention projection's weight gradient (adm_blocks.py:176): dy arrives with its rows in the packed (head, {q,k,v}, c) order of
    the attention kernels and the parameter wants (head, c, {q,k,v}).  Both 1x1 forms -- the direct kernel and, with enough pixels,
    wgrad_gemm1_kernel -- leave the row permutation to wgrad_reduce_kernel."""
    heads, Cin = 2, 128
    Cout = heads * 192
    tag = f"t/bwd/qkvwg/{B}_{H}_{W}"
    x = fx.randn(tag + "/x", B, Cin, H, W)
    dy = fx.randn(tag + "/dy", B, Cout, H, W)                 # packed row order
    dwp = torch.einsum("bohw,bihw->oi", dy.double(), x.double())
    dbp = dy.double().sum(dim=(0, 2, 3))
    per, d = Cout // heads, Cout // heads // 3
    idx = torch.empty(Cout, dtype=torch.long)                  # reference row co <- packed row cp
    for cp in range(Cout):
        hh, rr = divmod(cp, per)
        which, c = divmod(rr, d)
        idx[hh * per + c * 3 + which] = cp
    lib.prof_enable(True)
    try:
        dw, db = lib.op_conv_wgrad(dev(dy), dev(x), None, 1, qkv_heads=heads)
        torch.cuda.synchronize()
        names = {r["name"] for r in lib.prof_report()}
    finally:
        lib.prof_enable(False)
    assert ("wgrad_gemm1_kernel" in names) == gemm, names
    close(dw.reshape(Cout, Cin), dwp[idx], what="dW")
    close(db, dbp[idx], what="db")


@pytest.mark.parametrize("B,Ca,Cb,Cout,H,W,k", [(2, 64, 64, 128, 32, 32, 1), (3, 128, 128, 128, 16, 48, 1), (2, 32, 96, 64, 12, 24, 3)])
def test_wgrad_reads_an_untransformed_concat_in_place(lib, B, Ca, Cb, Cout, H, W, k):
    """The decoder's skip projections (adm_blocks.py:150-151): their weight gradient's operand is cat(x, skip) itself -- no transform,
    no activation --, which the direct kernel now reads from the two tensors in place (round 5: the 537 MB copy at 128^2 is gone)."""
    tag = f"t/bwd/catwg/{B}_{Ca}_{Cb}_{Cout}_{H}_{W}_{k}"
    xa, xb = fx.randn(tag + "/xa", B, Ca, H, W), fx.randn(tag + "/xb", B, Cb, H, W)
    dy = fx.randn(tag + "/dy", B, Cout, H, W)
    w = torch.zeros(Cout, Ca + Cb, k, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    gw, gb = torch.autograd.grad(F.conv2d(torch.cat([xa, xb], 1).double(), w, b, padding=k // 2), (w, b), dy.double())
    lib.prof_enable(True)
    try:
        dw, db = lib.op_conv_wgrad(dev(dy), dev(xa), dev(xb), k)
        torch.cuda.synchronize()
        names = {r["name"] for r in lib.prof_report()}
    finally:
        lib.prof_enable(False)
    assert "act_materialize_kernel" not in names and any(n.startswith("wgrad_kernel") for n in names), names
    close(dw, gw, what="dW")
    close(db, gb, what="db")


@pytest.mark.parametrize("case", [
    # (B, Ca, Cb, Hs, Ws, resample, act, film, add_mode, accumulate)
    (2, 64, 0, 8, 8, 0, 1, True, 0, False), (2, 64, 64, 8, 12, 0, 1, False, 1, False), (2, 64, 0, 8, 8, 1, 1, False, 2, True),
    (2, 64, 0, 16, 16, 2, 1, False, 2, False), (3, 128, 0, 6, 6, 0, 0, False, 1, True), (2, 256, 0, 4, 4, 0, 1, True, 0, False),
    # slabs of up to 8192 elements: gn_bwd_reg_kernel (round 5: the slab in registers; 1 / 4 / 8 / 2 / 8 quads per thread; the first
    # case of this list is one too: 16 lanes per channel) ...
    (2, 128, 0, 32, 32, 0, 1, True, 1, False), (1, 64, 0, 32, 64, 0, 0, True, 0, False), (2, 64, 64, 16, 32, 0, 1, False, 1, True),
    (3, 256, 0, 16, 16, 0, 1, True, 0, True), (2, 32, 0, 32, 32, 0, 1, False, 0, False),
    # ... a larger slab on the two-pass kernel (no counters given) ...
    (2, 64, 64, 64, 64, 0, 1, False, 1, True),
    # ... and on gn_bwd_lds_kernel (round 5: pieces of 4096 elements stay in LDS between the passes; 16 / 16 / 8 / 4 workgroups per
    # slab exchange their sums through `sync`)
    (2, 64, 0, 128, 128, 0, 1, True, 1, False, "sync"), (3, 128, 0, 128, 128, 0, 1, False, 0, True, "sync"),
    (2, 128, 128, 64, 64, 0, 1, True, 1, False, "sync"), (2, 64, 64, 64, 64, 0, 0, False, 1, True, "sync"),
    # ... resampled inputs too: the gradient is mapped back through the 2x up-sampling / 2x2 mean on its way into LDS
    (2, 128, 0, 64, 64, 1, 1, False, 2, True, "sync"), (2, 64, 0, 128, 128, 2, 1, True, 2, False, "sync"),
    (1, 128, 0, 64, 64, 2, 1, False, 1, False, "sync"),
])
def test_gn_film_silu_backward(lib, case):
    use_sync = len(case) > 10
    B, Ca, Cb, Hs, Ws, rs, act, use_film, add_mode, accumulate = case[:10]
    tag = "t/bwd/gn/" + "_".join(map(str, case))
    C = Ca + Cb
    Hc, Wc = (Hs * 2, Ws * 2) if rs == 1 else ((Hs // 2, Ws // 2) if rs == 2 else (Hs, Ws))
    xa = (fx.randn(tag + "/xa", B, Ca, Hs, Ws) * 1.3 + 0.4).requires_grad_(True)
    xb = (fx.randn(tag + "/xb", B, Cb, Hs, Ws) * 0.7).requires_grad_(True) if Cb else None
    gamma = fx.param(tag, "norm.weight", (C,)).requires_grad_(True)
    beta = fx.param(tag, "norm.bias", (C,)).requires_grad_(True)
    film = (fx.randn(tag + "/film", B, 2 * C) * 0.3).requires_grad_(True) if use_film else None
    dact = fx.randn(tag + "/dact", B, C, Hc, Wc)
    add = None
    if add_mode == 1:
        add = fx.randn(tag + "/add", B, C, Hs, Ws)
    elif add_mode == 2:
        add = fx.randn(tag + "/add", B, C, Hc, Wc)
    init = (fx.randn(tag + "/ia", B, Ca, Hs, Ws), fx.randn(tag + "/ib", B, Cb, Hs, Ws) if Cb else None) if accumulate else None
    # reference
    x = torch.cat([xa, xb], 1) if Cb else xa
    t = orc.group_norm(x, gamma, beta)
    if use_film:
        t = torch.addcmul(film[:, C:, None, None], t, film[:, :C, None, None] + 1)
    u = F.silu(t) if act else t
    res = (lambda v: orc.resample_up(v)) if rs == 1 else ((lambda v: orc.resample_down(v)) if rs == 2 else (lambda v: v))
    loss = (res(u) * dact).sum()
    if add_mode == 1:
        loss = loss + (x * add).sum()
    elif add_mode == 2:
        loss = loss + (res(x) * add).sum()
    ins = [xa, gamma, beta] + ([xb] if Cb else []) + ([film] if use_film else [])
    gr = torch.autograd.grad(loss, ins)
    gxa, gg, gb = gr[0], gr[1], gr[2]
    gxb = gr[3] if Cb else None
    gfilm = gr[-1] if use_film else None
    if accumulate:
        gxa = gxa + init[0]
        gxb = gxb + init[1] if Cb else None
    # HIP
    coef, stats = lib.op_gn_coef(dev(xa), dev(xb) if Cb else None, dev(gamma), dev(beta), film=dev(film) if use_film else None,
                                 film_batch=1, film_stride=2 * C, want_stats=True)
    sync = torch.zeros(lib.GN_SYNC_WORDS * B * min(32, C // 4), dtype=torch.int32, device="cuda") if use_sync else None
    lib.prof_enable(True)
    dxa, dxb, dg, dbt, dfilm = lib.op_gn_bwd(dev(dact), dev(xa), dev(xb) if Cb else None, coef, stats, dev(gamma), dev(beta),
                                             film=dev(film) if use_film else None, film_batch=1, film_stride=2 * C, act=act,
                                             resample=rs, add=dev(add) if add is not None else None, add_mode=add_mode,
                                             dx_init=(dev(init[0]), dev(init[1]) if Cb else None) if accumulate else None,
                                             sync=sync)
    torch.cuda.synchronize()
    names = {r["name"] for r in lib.prof_report()}
    lib.prof_enable(False)
    on_chip = use_sync and (rs == 0 or Ws % 4 == 0) and (Hs * Ws) % 4096 == 0 and (C // min(32, C // 4)) * Hs * Ws >= 16384
    cpg, hw = C // min(32, C // 4), Hs * Ws
    in_regs = (not on_chip and rs == 0 and hw % 4 == 0 and cpg * hw <= 8192
               and ((hw // 4) % 64 == 0 or (16 <= hw // 4 < 64 and (hw // 4) & (hw // 4 - 1) == 0)))
    want = "gn_bwd_lds_kernel" if on_chip else "gn_bwd_reg_kernel" if in_regs else "gn_bwd_kernel"
    assert {n_ for n_ in names if n_.startswith("gn_bwd")} == {want}, (names, want, case)
    if use_sync:
        assert int(sync.abs().sum()) == 0, "the exchange area must be left zero"
    close(dxa, gxa, what="dxa")
    if Cb:
        close(dxb, gxb, what="dxb")
    close(dg, gg, what="dgamma")
    close(dbt, gb, what="dbeta")
    if use_film:
        close(dfilm, gfilm, what="dfilm")


def test_gn_backward_partner_wait_timeout_gives_the_same_bits(lib, tmp_path):
    """gn_bwd_lds_kernel, groups split over workgroups: a workgroup whose (bounded) wait for its partners' channel sums runs out
    computes them itself from global memory.  With MCEDM_GN_BWD_SPIN_US=0 (read once per process: a child process) most workgroups
    take that path; the result must equal the waiting run bit for bit, and the counters must come back zero."""
    import os
    import subprocess
    import sys
    script = r'''
import importlib, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
L = importlib.import_module("m-cedm_amd.lib"); L.load()
g = torch.Generator().manual_seed(3)
B, C, H, W = 4, 128, 128, 128
x = (torch.randn(B, C, H, W, generator=g) * 1.3 + 0.4).cuda()
dact = torch.randn(B, C, H, W, generator=g).cuda()
gamma, beta = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
coef, stats = L.op_gn_coef(x, None, gamma, beta, want_stats=True)
out = {}
for rep in range(2):
    sync = torch.zeros(L.GN_SYNC_WORDS * B * 32, dtype=torch.int32, device="cuda")
    dxa, _, dg, dbt, _ = L.op_gn_bwd(dact, x, None, coef, stats, gamma, beta, act=1, sync=sync)
    torch.cuda.synchronize()
    assert int(sync.abs().sum()) == 0
    out[f"dx{rep}"] = dxa.cpu().numpy(); out[f"dg{rep}"] = dg.cpu().numpy(); out[f"db{rep}"] = dbt.cpu().numpy()
np.savez(sys.argv[2], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for us in ("100", "0"):
        path = str(tmp_path / f"spin{us}.npz")
        r = subprocess.run([sys.executable, "-c", script, root, path], env=dict(os.environ, MCEDM_GN_BWD_SPIN_US=us), capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[us] = dict(np.load(path))
    for k in res["100"]:
        assert np.isfinite(res["100"][k]).all()
        assert np.array_equal(res["100"][k], res["0"][k]), k
    assert np.array_equal(res["100"]["dx0"], res["100"]["dx1"]) and np.array_equal(res["0"]["dx0"], res["0"]["dx1"])


def test_gn_backward_on_chip_under_contention(lib):
    """gn_bwd_lds_kernel's workgroups wait (bounded) for partners that may not be resident yet.  Run it on one stream while another
    stream keeps the CUs busy with large convolutions, twenty times: every result must equal the quiet run bit for bit (whichever
    mix of "partner arrived" / "recomputed" each workgroup saw), nothing may hang, and the exchange area must come back zero."""
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 8, 128, 128, 128
    x = (torch.randn(B, C, H, W, generator=g) * 1.3 + 0.4).cuda()
    dact = torch.randn(B, C, H, W, generator=g).cuda()
    gamma, beta = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    coef, stats = lib.op_gn_coef(x, None, gamma, beta, want_stats=True)
    sync = torch.zeros(lib.GN_SYNC_WORDS * B * 32, dtype=torch.int32, device="cuda")
    quiet = lib.op_gn_bwd(dact, x, None, coef, stats, gamma, beta, act=1, sync=sync)[0].clone()
    torch.cuda.synchronize()
    # the contender: 3x3 convs of the same size on a second stream
    wc = (torch.randn(128, 128, 3, 3, generator=g) / 34.0).cuda()
    wino = lib.op_pack_conv_wino(wc)
    side = torch.cuda.Stream()
    ycon = torch.empty(B, 128, H, W, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(40):
            lib.op_conv_wino(x, None, wino, None, 128, out=ycon)
    for rep in range(20):
        got = lib.op_gn_bwd(dact, x, None, coef, stats, gamma, beta, act=1, sync=sync)[0]
        assert torch.equal(got, quiet), rep
    torch.cuda.synchronize()
    assert int(sync.abs().sum()) == 0


def packed_qkv(qkv, heads):
    B, C3, H, W = qkv.shape
    d = C3 // heads // 3
    return qkv.reshape(B, heads, d, 3, H, W).permute(0, 1, 3, 2, 4, 5).reshape(B, C3, H, W).contiguous()


def unpacked_qkv(p, heads):
    B, C3, H, W = p.shape
    d = C3 // heads // 3
    return p.reshape(B, heads, 3, d, H, W).permute(0, 1, 3, 2, 4, 5).reshape(B, C3, H, W).contiguous()


@pytest.mark.parametrize("T", list(fx.ATTN_CASES))
def test_attention_backward_golden(lib, golden, T):
    g = golden("ops.npz")
    B, hw = fx.ATTN_CASES[T]
    qkv = fx.randn(f"ops/attn{T}/qkv", B, 384, *hw)
    da = fx.randn(f"ops/attn{T}/da", B, 128, *hw)
    pq = dev(packed_qkv(qkv, 2))
    a = lib.op_attention(pq, 2)
    dqkv = lib.op_attention_bwd(pq, a, dev(da), 2)
    close(unpacked_qkv(dqkv.cpu(), 2), g[f"attn{T}_dqkv"], what=f"dqkv T={T}")


@pytest.mark.parametrize("B,heads,hw", [(1, 1, (6, 6)), (2, 1, (32, 32)), (1, 2, (2, 2)),
                                        (2, 2, (16, 16)), (1, 1, (8, 16)), (3, 1, (16, 24))])      # T = 256, 128 (LDS-staged kernels), 384
def test_attention_backward_vs_oracle(lib, B, heads, hw):
    qkv = fx.randn(f"t/bwd/attn/{B}{heads}{hw}", B, heads * 192, *hw).requires_grad_(True)
    da = fx.randn(f"t/bwd/attn/da/{B}{heads}{hw}", B, heads * 64, *hw)
    a = orc.attention(qkv, heads)
    (ref,) = torch.autograd.grad(a, qkv, da)
    pq = dev(packed_qkv(qkv.detach(), heads))
    dqkv = lib.op_attention_bwd(pq, lib.op_attention(pq, heads), dev(da), heads)
    close(unpacked_qkv(dqkv.cpu(), heads), ref, what="dqkv")


@pytest.mark.parametrize("hw,scale", [((16, 16), 4.0), ((32, 32), 6.0), ((6, 6), 6.0)])
def test_attention_backward_with_large_scores(lib, hw, scale):
    """The LDS-staged dq kernel normalises the scores ONLINE (running maximum and sum, rescaled when the maximum moves) and in base 2:
    with q, k scaled so that the scores span +-100 and more (softmax rows nearly one-hot, maxima that move many times along the key axis)
    nothing may overflow, and the gradients must still match fp64 autograd.  T = 256 / 1024: LDS kernels; T = 36: the two-pass statistics."""
    B, heads = 2, 1
    qkv = fx.randn(f"t/bwd/attn/big/{hw}", B, heads * 192, *hw) * scale
    da = fx.randn(f"t/bwd/attn/big/da/{hw}", B, heads * 64, *hw)
    q64 = qkv.double().requires_grad_(True)
    a = orc.attention(q64, heads)
    (ref,) = torch.autograd.grad(a, q64, da.double())
    pq = dev(packed_qkv(qkv, heads))
    dqkv = lib.op_attention_bwd(pq, lib.op_attention(pq, heads), dev(da), heads)
    got = unpacked_qkv(dqkv.cpu(), heads)
    assert torch.isfinite(got).all()
    close(got, ref, what="dqkv (large scores)")


def make_plan(L, cfg):
    return L.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks,
                  cfg.attn_resolutions, cfg.resolution)


def run_training_step(L, cfg, P, xc, cond_in, mc, noise, rnd_normal, variants=()):
    plan = make_plan(L, cfg)
    for which, value in variants:
        plan.set_variant(which, value)
    params = {k: dev(v) for k, v in P.items()}
    packed = plan.pack(params)
    ws = L.Workspace()
    x_noise, sigma = L.edm_noise_inputs(dev(xc), dev(mc), dev(noise), dev(rnd_normal.flatten()))
    D = plan.denoise(packed, x_noise, sigma, cond=dev(cond_in), ws=ws, training=True)
    loss, dD = L.edm_loss(D, dev(xc), dev(mc), sigma)
    grads = [torch.full_like(params[n], float("nan")) for n in plan.param_names]
    plan.denoise_backward(packed, params, x_noise, sigma, dev(cond_in), dD, grads, ws)
    return plan, loss, dict(zip(plan.param_names, grads))


def test_training_step_golden(lib, golden):
    """training_step loss and gradients (config P: B=4, 32x32, ch=64) vs the reference's own numbers."""
    g = golden("training_P.npz")
    P = orc.make_params(fx.CFG_P, int(g["seed"]))
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    xc, cond_in, mc = fx.training_nchw(h, u, mask, cond_noise)
    plan, loss, grads = run_training_step(lib, fx.CFG_P, P, xc, cond_in, mc, noise, rnd_normal)
    close(loss, torch.as_tensor(g["loss"]).reshape(1), what="loss")
    for n in fx.TRAIN_GRAD_NAMES:
        close(grads[n], g[f"grad::{n}"], rtol=1e-4, rel_atol=1e-5, what=f"grad {n}")
    sq = sum(float((v.double() ** 2).sum()) for v in grads.values())
    assert abs(sq - float(g["grad_sqnorm_total"])) <= 1e-3 * float(g["grad_sqnorm_total"])
    each = np.array([float((grads[n].double() ** 2).sum()) for n in plan.param_names])
    np.testing.assert_allclose(each, g["grad_sqnorm_each"], rtol=5e-3)


def test_training_step_golden_winograd_weight_gradients(lib, golden):
    """The same golden with the plan's 'wgrad_wino' variant forced on: the 64 -> 64 and 128 -> 64 convs of the 32^2 and 16^2 levels
    take the 64-channel Winograd weight-gradient kernel (at this batch the default leaves them to the direct kernel: too few
    tiles per split), and the reference's gradients still come out."""
    g = golden("training_P.npz")
    P = orc.make_params(fx.CFG_P, int(g["seed"]))
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    xc, cond_in, mc = fx.training_nchw(h, u, mask, cond_noise)
    lib.prof_enable(True)
    try:
        plan, loss, grads = run_training_step(lib, fx.CFG_P, P, xc, cond_in, mc, noise, rnd_normal, variants=[("wgrad_wino", 1)])
        rows = {r["name"]: int(r["launches"]) for r in lib.prof_report()}
    finally:
        lib.prof_enable(False)
    assert rows.get("wgrad_wino64_kernel", 0) >= 12 and rows.get("wgrad_wino_finish_kernel", 0) >= 12, rows
    close(loss, torch.as_tensor(g["loss"]).reshape(1), what="loss")
    for n in fx.TRAIN_GRAD_NAMES:
        close(grads[n], g[f"grad::{n}"], rtol=1e-4, rel_atol=1e-5, what=f"grad {n}")


CFG_M = orc.UNetConfig(ch=64, ch_mult=(1, 2, 4), attn_resolutions=(8,))


@pytest.mark.parametrize("cfg_name", ["P", "W", "M"])
def test_training_step_all_grads_vs_oracle(lib, cfg_name):
    """every parameter gradient vs autograd through the oracle (P: ch=64 3 levels; W: ch=128, 4 levels, 2-head attention;
    M: channel multipliers (1, 2, 4): decoder concats 256+128 / 128+64 whose GroupNorm groups (12 / 6 channels)
    straddle the concat boundary, 4-head attention at 8x8)"""
    cfg = {"P": fx.CFG_P, "W": fx.CFG_W, "M": CFG_M}[cfg_name]
    B, H, W = (2, 16, 16) if cfg_name == "W" else (2, 32, 32)
    tag = f"t/bwd/train/{cfg_name}"
    P = orc.make_params(cfg, 3)
    xc = fx.randn(tag + "/x", B, 2, H, W)
    mc = torch.zeros(B, 2, H, W)
    mc[0, 1] = 1
    mc[1, 0] = 1
    mc[1, 1, : H // 2] = 1
    cond_in = xc * (1 - mc) + fx.randn(tag + "/cn", B, 2, H, W) * mc
    noise = fx.randn(tag + "/noise", B, 2, H, W)
    rnd_normal = fx.randn(tag + "/rnd", B, 1, 1, 1)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref_loss = orc.training_loss(Pg, cfg, xc, cond_in, mc, noise, rnd_normal)
    ref_loss.backward()
    plan, loss, grads = run_training_step(lib, cfg, P, xc, cond_in, mc, noise, rnd_normal)
    close(loss, ref_loss.detach().reshape(1), what="loss")
    for n in plan.param_names:
        close(grads[n], Pg[n].grad, rtol=1e-4, rel_atol=1e-5, what=f"grad {n}")


def test_s128_network_training_step_at_a_winograd_size_vs_oracle(lib):
    """the S128 network (ch = 128, ch_mult (1, 1, 1, 1), attention at 16^2) at 64 x 64, where the
    Winograd kernel serves the 64^2 and 32^2 levels: forward convs AND the data-gradient convs (WinoCfg<4>, plain and
    up-sampling variants, the RS_DOWN / RS_UP residual modes of the down / up blocks) are compared with autograd through
    the oracle, every one of the 260 gradients; the kernel names recorded by the profiler prove which kernels ran."""
    cfg = fx.CFG_W
    B, H, W = 2, 64, 64
    tag = "t/bwd/train/W64"
    P = orc.make_params(cfg, 11)
    xc = fx.randn(tag + "/x", B, 2, H, W)
    mc = torch.zeros(B, 2, H, W)
    mc[0, 1] = 1
    mc[1, 0] = 1
    mc[1, 1, : H // 2] = 1
    cond_in = xc * (1 - mc) + fx.randn(tag + "/cn", B, 2, H, W) * mc
    noise = fx.randn(tag + "/noise", B, 2, H, W)
    rnd_normal = fx.randn(tag + "/rnd", B, 1, 1, 1)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref_loss = orc.training_loss(Pg, cfg, xc, cond_in, mc, noise, rnd_normal)
    ref_loss.backward()
    lib.prof_enable(True)
    try:
        plan, loss, grads = run_training_step(lib, cfg, P, xc, cond_in, mc, noise, rnd_normal)
        rows = {r["name"]: int(r["launches"]) for r in lib.prof_report()}      # one row per kernel name
    finally:
        lib.prof_enable(False)
    names = rows
    n_plain = sum(v for k, v in rows.items() if k.startswith("conv_wino_kernel<WinoCfg<4>, false"))      # with / without activation
    n_up = sum(v for k, v in rows.items() if k.startswith("conv_wino_kernel<WinoCfg<4>, true"))
    assert rows.get("conv_wino_kernel<WinoCfg<4>, false, false>", 0) >= 8, rows      # the data-gradient convs: no activation
    assert rows.get("wgrad_wino_kernel", 0) >= 8, rows                              # the 128-channel 3x3 weight gradients
    assert rows.get("wgrad_thin4_kernel", 0) == 2, rows                              # conv_in and out_conv
    assert rows.get("conv1x1_reg_kernel", 0) >= 4, rows                              # the decoder's skip projections and their data gradients
    assert rows.get("gn_bwd_lds_kernel", 0) >= 8 and rows.get("gn_bwd_reg_kernel", 0) >= 8, rows      # GroupNorm backward: slabs in LDS (64^2) / in registers
    assert rows.get("act_materialize_kernel", 0) <= 9, rows                          # the skip projections' operand is read in place
    # forward convs of the 64^2 and 32^2 levels and, in the backward, their data-gradient convs
    assert n_plain >= 20 and n_up >= 1, (n_plain, n_up, names)
    close(loss, ref_loss.detach().reshape(1), what="loss")
    for n in plan.param_names:
        close(grads[n], Pg[n].grad, rtol=1e-4, rel_atol=1e-5, what=f"grad {n}")


def test_reductions_on_two_streams_do_not_share_state(lib):
    """VERDICT r3 item 8 / SURVEY 8(b) "re-entrant per plan": the fixed-order sums of mcedm_edm_loss and mcedm_sqnorm keep
    their partial sums and ticket in CALLER-owned scratch (ABI 3).  Two independent loss / norm computations are enqueued
    back to back on two streams, many times, each with its own scratch: every result equals the serial one, bit for bit."""
    torch.manual_seed(0)
    dev0 = torch.device("cuda", 0)
    B, C_, H, W = 8, 2, 128, 128
    sets = []
    for i in range(2):
        D = torch.randn(B, C_, H, W, device=dev0) * (1 + i)
        x = torch.randn(B, C_, H, W, device=dev0)
        m = (torch.rand(B, C_, H, W, device=dev0) > 0.5).float()
        sg = torch.rand(B, device=dev0) * 3 + 0.1
        g = torch.randn(6_000_000 + 1237 * i, device=dev0)
        sets.append((D, x, m, sg, g))
    serial = []
    for D, x, m, sg, g in sets:
        loss, dD = lib.edm_loss(D, x, m, sg)
        serial.append((loss.clone(), dD.clone(), lib.sqnorm(g).clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev0) for _ in range(2)]
    scratch = [torch.empty(lib.REDUCE_SCRATCH_BYTES, dtype=torch.uint8, device=dev0) for _ in range(2)]
    outs = [[], []]
    for it in range(40):
        for i, (D, x, m, sg, g) in enumerate(sets):
            with torch.cuda.stream(streams[i]):
                loss, dD = lib.edm_loss(D, x, m, sg, scratch=scratch[i])
                outs[i].append((loss, lib.sqnorm(g, scratch=scratch[i])))
    torch.cuda.synchronize()
    for i in range(2):
        for loss, sq in outs[i]:
            assert torch.equal(loss, serial[i][0]) and torch.equal(sq, serial[i][2])
    with pytest.raises(RuntimeError, match="MCEDM_REDUCE_SCRATCH_BYTES"):
        lib.sqnorm(sets[0][4], scratch=torch.empty(64, dtype=torch.uint8, device=dev0))


def test_inference_forward_wide_multipliers_vs_oracle(lib):
    """config M through the inference path (GroupNorm rows derived inside the consuming conv, groups straddling the
    concat boundary) against the oracle's preconditioned forward"""
    P = orc.make_params(CFG_M, 5)
    plan = make_plan(lib, CFG_M)
    packed = plan.pack({k: dev(v) for k, v in P.items()})
    B, H, W = 3, 32, 32
    x = fx.randn("t/fwd/M/x", B, 2, H, W) * 2.0
    cond = fx.randn("t/fwd/M/c", B, 2, H, W)
    sigma = torch.tensor([0.7, 2.0, 11.0])
    D = plan.denoise(packed, dev(x), dev(sigma), cond=dev(cond))
    with torch.no_grad():
        ref = orc.model_precond(P, CFG_M, x, sigma, cond)
    close(D, ref, what="denoised (config M)")


def test_training_step_is_bitwise_reproducible(lib):
    """No atomics anywhere in the step (wgrad stores one partial block per split and reduces them in a fixed order): the same
    batch gives the same loss and the same gradients, bit for bit, run after run -- at a size where the split-K is wide
    (ch = 128, 64 x 64: 128 splits per conv)."""
    cfg = fx.CFG_W
    B, H, W = 4, 64, 64
    tag = "t/bwd/repro"
    P = orc.make_params(cfg, 3)
    xc = fx.randn(tag + "/x", B, 2, H, W)
    mc = torch.zeros(B, 2, H, W)
    mc[:, 1] = 1
    cond_in = xc * (1 - mc) + fx.randn(tag + "/cn", B, 2, H, W) * mc
    noise = fx.randn(tag + "/noise", B, 2, H, W)
    rnd_normal = fx.randn(tag + "/rnd", B, 1, 1, 1)
    runs = [run_training_step(lib, cfg, P, xc, cond_in, mc, noise, rnd_normal) for _ in range(3)]
    plan, loss0, g0 = runs[0]
    for _, loss, g in runs[1:]:
        assert torch.equal(loss, loss0)
        for n in plan.param_names:
            assert torch.equal(g[n], g0[n]), n
    assert all(bool(torch.isfinite(v).all()) for v in g0.values())
