"""GPU: the register-direct 1x1 GEMM kernel (csrc/conv1x1_reg.hip; the decoder blocks' skip projections, adm_blocks.py:150-151, 171)
through mcedm_op_conv: against an fp64 convolution, against the tiled MFMA kernel it replaces, bit-invariance under batch
sharding, and the shapes it must leave to the other kernels.  rtol 1e-4 / atol 1e-5."""
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures as fx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    L.load()
    yield L
    L.set_conv1x1_reg(-1)


def dev(t):
    return t.detach().contiguous().cuda()


def run(L, fn):
    L.prof_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
        names = {r["name"] for r in L.prof_report()}
    finally:
        L.prof_enable(False)
    return out, names


@pytest.mark.parametrize("B,Ca,Cb,Cout,H,W,use_res", [
    (2, 128, 128, 128, 32, 32, False),      # the decoder shape: channel concat, two tiles per image
    (3, 64, 0, 128, 32, 32, True),          # two tiles per image, odd batch, residual
    (2, 128, 0, 256, 64, 64, False),        # the skip projection's data gradient: two output-channel blocks
    (1, 32, 32, 128, 16, 64, True),         # the smallest K (two stages of 16 channels per source), two tiles
    (5, 256, 0, 128, 32, 32, False),        # tile count not divisible by the workgroup count
    (2, 64, 64, 64, 64, 64, False),         # the ch = 64 network's skip projection: 64-channel workgroups, 1024-pixel tiles
    (3, 128, 0, 64, 64, 128, True),         # the same with a residual, rectangular, odd batch
    (2, 64, 0, 192, 64, 64, False),         # three 64-channel blocks
])
def test_conv1x1_reg_vs_fp64_and_tiled_kernel(lib, B, Ca, Cb, Cout, H, W, use_res):
    tag = f"c1/{B}_{Ca}_{Cb}_{Cout}_{H}_{W}"
    Cin = Ca + Cb
    xa = fx.randn(tag + "/xa", B, Ca, H, W)
    xb = fx.randn(tag + "/xb", B, Cb, H, W) if Cb else None
    w = fx.randn(tag + "/w", Cout, Cin, 1, 1) / Cin ** 0.5
    b = fx.randn(tag + "/b", Cout) * 0.1
    res = fx.randn(tag + "/r", B, Cout, H, W) if use_res else None
    x = torch.cat([xa, xb], 1) if Cb else xa
    ref = F.conv2d(x.double(), w.double(), b.double())
    if use_res:
        ref = ref + res.double()
    wpk, bpk = lib.op_pack_conv(dev(w), dev(b))
    args = (dev(xa), dev(xb) if Cb else None, wpk, bpk, Cout, 1)
    kw = dict(res=dev(res)) if use_res else {}
    lib.set_conv1x1_reg(1)
    got, names = run(lib, lambda: lib.op_conv(*args, **kw))
    assert "conv1x1_reg_kernel" in names, names
    torch.testing.assert_close(got.cpu().double(), ref, rtol=1e-4, atol=1e-5)
    lib.set_conv1x1_reg(0)
    old, names = run(lib, lambda: lib.op_conv(*args, **kw))
    lib.set_conv1x1_reg(-1)
    assert "conv1x1_reg_kernel" not in names, names
    torch.testing.assert_close(got, old, rtol=1e-5, atol=2e-6)
    # a sample's bits do not depend on the batch it is computed in
    one = lib.op_conv(dev(xa[B - 1:]), dev(xb[B - 1:]) if Cb else None, wpk, bpk, Cout, 1, **(dict(res=dev(res[B - 1:])) if use_res else {}))
    assert torch.equal(one[0], got[B - 1])


def test_conv1x1_reg_leaves_other_shapes_alone(lib):
    lib.set_conv1x1_reg(1)
    try:
        for (Cin, Cout, H, W, coef) in [(128, 128, 16, 16, False), (128, 64, 32, 32, False), (128, 96, 64, 64, False), (40, 128, 32, 32, False), (128, 128, 32, 32, True)]:
            tag = f"c1/no/{Cin}_{Cout}_{H}_{W}_{coef}"
            x = fx.randn(tag + "/x", 2, Cin, H, W)
            w = fx.randn(tag + "/w", Cout, Cin, 1, 1) / Cin ** 0.5
            wpk, bpk = lib.op_pack_conv(dev(w), None)
            cf = torch.tensor([0.1, 1.2, -0.1, 0.0]).repeat(2, Cin, 1) if coef else None
            y, names = run(lib, lambda: lib.op_conv(dev(x), None, wpk, None, Cout, 1, coef=dev(cf) if coef else None))
            assert "conv1x1_reg_kernel" not in names, (names, Cin, Cout, H, W)
            xr = (x - 0.1) * 1.2 - 0.1 if coef else x
            torch.testing.assert_close(y.cpu().double(), F.conv2d(xr.double(), w.double()), rtol=1e-4, atol=1e-5)
    finally:
        lib.set_conv1x1_reg(-1)
