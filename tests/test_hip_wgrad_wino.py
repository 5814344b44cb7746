"""GPU parity of the Winograd F(3x3, 2x2) weight-gradient kernel (csrc/wgrad_wino.hip) through the C ABI (mcedm_op_conv_wgrad):
against fp64 autograd of the reference's convolution (adm_blocks.py:78-79), against the direct split-K kernel it replaces, run to
run (bitwise), and with the profiler's kernel names proving which kernel served the call.

Tolerance: rtol 1e-4 with atol = 1e-5 x max|ref| per tensor, as for every gradient (tests/test_hip_backward.py)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures as fx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    assert torch.cuda.is_available()
    L.load()
    yield L
    L.set_wgrad_wino(-1)


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


def kernels_of(L, fn):
    L.prof_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
        names = {r["name"] for r in L.prof_report()}
    finally:
        L.prof_enable(False)
    return out, names


def reference(xa, xb, coef, act, dy, Cout):
    """fp64 autograd of conv2d(act(coef(cat(xa, xb)))) w.r.t. weight and bias"""
    x = torch.cat([xa, xb], 1) if xb is not None else xa
    x = x.double()
    c = coef.double()
    u = (x - c[:, :, 0, None, None]) * c[:, :, 1, None, None] + c[:, :, 2, None, None]
    if act:
        u = F.silu(u)
    w = torch.zeros(Cout, x.shape[1], 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(u, w, b, padding=1)
    return torch.autograd.grad(y, (w, b), dy.double())


def rand_coef(tag, B, C):
    return torch.stack([fx.randn(tag + "/mean", B, C) * 0.3, 1 + 0.3 * fx.randn(tag + "/scale", B, C),
                        0.2 * fx.randn(tag + "/off", B, C), torch.zeros(B, C)], dim=-1)


SHAPES = [
    # (B, Ca, Cb, Cout, H, W, act)
    (2, 128, 0, 128, 32, 32, 1),      # one 32-pixel segment per row: left and right image borders in every stage
    (1, 128, 0, 128, 2, 32, 1),       # one tile row: top and bottom borders in the same stage; a single stage per split
    (2, 128, 128, 128, 64, 64, 1),    # channel concat, two input-channel blocks
    (3, 128, 0, 256, 32, 64, 0),      # two output-channel blocks, rectangular image, odd batch, no activation
    (2, 256, 0, 128, 128, 128, 1),    # the S128 decoder shape at full resolution: four segments per row
    (5, 128, 0, 128, 6, 96, 1),       # stage count not divisible by the split count: ragged splits, clamped prefetch
    # the 64-channel form (all 16 positions per workgroup, 16-pixel segments): the reference's own ch = 64 network
    (2, 64, 0, 64, 32, 32, 1),        # two segments per row
    (1, 64, 0, 64, 2, 16, 1),         # one stage in all: every border in the same stage
    (2, 64, 64, 64, 64, 64, 1),       # the decoder's concat conv: two input-channel blocks
    (3, 64, 0, 128, 16, 48, 0),       # two output-channel blocks, rectangular, odd batch, no activation
    (2, 128, 0, 128, 16, 16, 1),      # 128 channels on a 16-pixel row: too narrow for the 128-channel form
    (5, 64, 0, 64, 6, 80, 1),         # ragged splits
    (2, 64, 0, 64, 128, 128, 1),      # the reference network's top level
]


def served_by(names):
    return {n for n in names if n in ("wgrad_wino_kernel", "wgrad_wino64_kernel")}


def expected_kernel(Cin, Cout, W):
    return "wgrad_wino_kernel" if Cin % 128 == 0 and Cout % 128 == 0 and W % 32 == 0 else "wgrad_wino64_kernel"


@pytest.mark.parametrize("shape", SHAPES)
def test_wgrad_wino_vs_fp64_and_direct(lib, shape):
    B, Ca, Cb, Cout, H, W, act = shape
    tag = "t/wgw/" + "_".join(map(str, shape))
    Cin = Ca + Cb
    xa = fx.randn(tag + "/xa", B, Ca, H, W)
    xb = fx.randn(tag + "/xb", B, Cb, H, W) if Cb else None
    coef = rand_coef(tag, B, Cin)
    dy = fx.randn(tag + "/dy", B, Cout, H, W)
    gw, gb = reference(xa, xb, coef, act, dy, Cout)
    args = (dev(dy), dev(xa), dev(xb) if Cb else None, 3)
    kw = dict(coef=dev(coef), act=act)
    lib.set_wgrad_wino(1)
    (dw, db), names = kernels_of(lib, lambda: lib.op_conv_wgrad(*args, **kw))
    assert served_by(names) == {expected_kernel(Cin, Cout, W)} and not any(n.startswith("wgrad_kernel") for n in names), names
    close(dw, gw, what="dW (Winograd)")
    close(db, gb, what="db (Winograd)")
    dw2, db2 = lib.op_conv_wgrad(*args, **kw)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "the Winograd weight gradient is not bitwise reproducible"
    lib.set_wgrad_wino(0)
    (dwd, dbd), names = kernels_of(lib, lambda: lib.op_conv_wgrad(*args, **kw))
    assert not served_by(names), names
    lib.set_wgrad_wino(-1)
    close(dwd, gw, what="dW (direct)")
    # the two kernels against each other, and their errors against fp64 side by side (the Winograd form must stay within 4x)
    ew = float((dw.cpu().double() - gw).abs().max()), float((dwd.cpu().double() - gw).abs().max())
    assert ew[0] <= 4 * ew[1] + 1e-7 * float(gw.abs().max()), f"Winograd error {ew[0]:.3e} vs direct {ew[1]:.3e}"
    close(dw, dwd.cpu(), rtol=2e-5, rel_atol=2e-6, what="Winograd vs direct")


@pytest.mark.parametrize("C,W", [(128, 64), (64, 48)])
def test_wgrad_wino_is_exact_on_small_integers(lib, C, W):
    """With small-integer inputs every product and partial sum of both forms is exact in fp32 (and the halves of G are powers of
    two), so the Winograd kernel must reproduce the reference bit for bit.  The kernel's first tile of a row starts one column
    left of the image and its last one ends one column right of it -- a neighbouring row's data, or memory in front of / behind
    the tensor: anything leaking in from there (or a wrong border mask) changes an integer."""
    B, H = 2, 8
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (B, C, H, W), generator=g).float()
    dy = torch.randint(-3, 4, (B, C, H, W), generator=g).float()
    ident = torch.tensor([0.0, 1.0, 0.0, 0.0]).repeat(B, C, 1)
    gw, gb = reference(x, None, ident, 0, dy, C)
    lib.set_wgrad_wino(1)
    try:
        (dw, db), names = kernels_of(lib, lambda: lib.op_conv_wgrad(dev(dy), dev(x), None, 3))
    finally:
        lib.set_wgrad_wino(-1)
    assert served_by(names) == {expected_kernel(C, C, W)}, names
    assert torch.equal(dw.cpu().double(), gw), f"max |diff| {float((dw.cpu().double() - gw).abs().max())}"
    assert torch.equal(db.cpu().double(), gb)


def test_wgrad_wino_not_taken_for_unserved_shapes(lib):
    lib.set_wgrad_wino(1)
    try:
        for (B, Cin, Cout, H, W) in [(2, 128, 128, 8, 8), (2, 96, 128, 32, 32), (2, 128, 32, 32, 32), (2, 128, 128, 32, 24), (2, 64, 64, 5, 16)]:
            tag = f"t/wgw/no/{Cin}_{Cout}_{H}_{W}"
            x = fx.randn(tag + "/x", B, Cin, H, W)
            dy = fx.randn(tag + "/dy", B, Cout, H, W)
            (dw, db), names = kernels_of(lib, lambda: lib.op_conv_wgrad(dev(dy), dev(x), None, 3))
            assert not served_by(names), (names, Cin, Cout, H, W)
            gw, gb = reference(x, None, torch.tensor([0.0, 1.0, 0.0, 0.0]).repeat(B, Cin, 1), 0, dy, Cout)
            close(dw, gw, what="dW")
    finally:
        lib.set_wgrad_wino(-1)
