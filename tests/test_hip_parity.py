"""GPU parity: every HIP kernel and schedule, called through the C ABI (ctypes), against the oracle on the
same seeded inputs and against the committed golden vectors (outputs of the reference itself).

Tolerance (north_star): rtol 1e-4 / atol 1e-5 in fp32 for kernels, blocks, the whole network AND the
35-evaluation sampler (the CPU noise floor of that sampler is 7-10e-7 on states of magnitude 4.7, the HIP path
measures 1.3e-6: DESIGN.md section 4); its observed entries must be preserved bit-exactly.
"""
import math

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5


@pytest.fixture(scope="module")
def lib():
    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib as L
    assert torch.cuda.is_available(), "these tests need the MI355X"
    L.load()
    return L


def dev(t):
    return t.contiguous().cuda()


def close(got, ref, rtol=RTOL, atol=ATOL, what=""):
    got = got.detach().cpu()
    ref = torch.as_tensor(ref)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got.double() - ref.double()).abs()
    lim = atol + rtol * ref.double().abs()
    bad = err > lim
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {err.max():.3e} (max ref {ref.abs().max():.3e})"


def apply_coef(x, coef, act=False):
    y = (x - coef[:, :, 0, None, None]) * coef[:, :, 1, None, None] + coef[:, :, 2, None, None]
    return torch.nn.functional.silu(y) if act else y


# ------------------------------------------------------------------ K1 GroupNorm coefficients
@pytest.mark.parametrize("C", [64, 128, 256])
def test_gn_coef_golden(lib, golden, C):
    x = fx.randn(f"ops/gn{C}/x", 2, C, 8, 8) * 1.5 + 0.3
    g, b = fx.param(f"ops/gn{C}", "norm.weight", (C,)), fx.param(f"ops/gn{C}", "norm.bias", (C,))
    coef, stats = lib.op_gn_coef(dev(x), None, dev(g), dev(b), want_stats=True)
    y = apply_coef(x, coef.cpu())
    close(y, golden("ops.npz")[f"gn{C}_y"], what=f"gn{C}")
    G = min(32, C // 4)
    xr = x.reshape(2, G, -1).double()
    close(stats[..., 0], xr.mean(-1).float(), what="mean")
    close(stats[..., 1], (1 / (xr.var(-1, unbiased=False) + 1e-5).sqrt()).float(), what="rstd")


def test_gn_coef_concat_film_ragged(lib):
    # virtual concat (Ca=64 | Cb=64 -> 32 groups of 4), FiLM rows per sample, HW not a multiple of 4
    xa, xb = fx.randn("t/gn/xa", 3, 64, 5, 3) + 2.0, fx.randn("t/gn/xb", 3, 64, 5, 3) * 3
    g, b = fx.param("t/gn", "norm.weight", (128,)), fx.param("t/gn", "norm.bias", (128,))
    film = fx.randn("t/gn/film", 3, 300) * 0.3
    coef = lib.op_gn_coef(dev(xa), dev(xb), dev(g), dev(b), film=dev(film[:, 20:]), film_batch=1, film_stride=280)
    x = torch.cat([xa, xb], 1)
    sc, sh = film[:, 20:148, None, None], film[:, 148:276, None, None]
    ref = torch.addcmul(sh, orc.group_norm(x, g, b), sc + 1)
    close(apply_coef(x, coef.cpu()), ref, what="gn concat+film")
    coef1 = lib.op_gn_coef(dev(xa), dev(xb), dev(g), dev(b), film=dev(film[1:2, 20:]), film_batch=0, film_stride=280)
    ref1 = torch.addcmul(sh[1:2], orc.group_norm(x, g, b), sc[1:2] + 1)
    close(apply_coef(x, coef1.cpu()), ref1, what="gn film broadcast")


# ------------------------------------------------------------------ K2/K3/K4 convolutions
@pytest.mark.parametrize("tag", ["k3", "k3up", "k3down", "k1"])
def test_conv_golden(lib, golden, tag):
    kw = fx.CONV_CASES[tag]
    cin, cout = fx.conv_channels(tag)
    k = kw["kernel"]
    w = fx.param(f"ops/conv_{tag}", "conv.weight", (cout, cin, k, k))
    b = fx.param(f"ops/conv_{tag}", "conv.bias", (cout,))
    x = fx.randn(f"ops/conv_{tag}/x", 2, cin, 8, 12)
    wpk, bpk = lib.op_pack_conv(dev(w), dev(b))
    rs = lib.RS_UP if kw.get("up") else (lib.RS_DOWN if kw.get("down") else lib.RS_NONE)
    y = lib.op_conv(dev(x), None, wpk, bpk, cout, k, resample=rs)
    close(y, golden("ops.npz")[f"conv_{tag}_y"], what=tag)


@pytest.mark.parametrize("tag", ["k0up", "k0down"])
def test_resample_only_skip_golden(lib, golden, tag):
    # Conv2d(kernel=0, up/down) is the skip of up/down blocks; on the HIP path it is the residual mode of conv1
    x = fx.randn(f"ops/conv_{tag}/x", 2, 8, 8, 12)
    ref = torch.as_tensor(golden("ops.npz")[f"conv_{tag}_y"])
    H, W = ref.shape[2:]
    z = fx.randn(f"t/{tag}/z", 2, 8, H, W)
    w = torch.zeros(8, 8, 3, 3)
    wpk, bpk = lib.op_pack_conv(dev(w), dev(torch.zeros(8)))
    mode = lib.RS_UP if tag == "k0up" else lib.RS_DOWN
    y = lib.op_conv(dev(z), None, wpk, bpk, 8, 3, res=dev(x), res_mode=mode)
    close(y, ref, what=tag)


@pytest.mark.parametrize("shape", [
    # (B, Ca, Cb, Cout, H, W, k, resample, act)
    (2, 64, 0, 64, 32, 32, 3, 0, 1), (1, 128, 128, 128, 16, 16, 3, 0, 1), (2, 64, 64, 64, 8, 8, 3, 0, 1),
    (2, 64, 0, 64, 16, 16, 3, 1, 1), (2, 64, 0, 64, 32, 32, 3, 2, 1), (3, 2, 2, 64, 32, 32, 3, 0, 0),
    (2, 64, 0, 2, 32, 32, 3, 0, 1), (1, 128, 0, 384, 16, 16, 1, 0, 0), (2, 64, 64, 64, 32, 32, 1, 0, 0),
    (1, 8, 0, 8, 4, 4, 3, 0, 1), (1, 8, 0, 8, 2, 2, 3, 0, 1), (2, 16, 0, 24, 20, 12, 3, 0, 1),
    (5, 64, 0, 64, 64, 64, 3, 0, 1), (1, 256, 0, 128, 40, 24, 3, 0, 1), (6, 128, 0, 128, 64, 64, 3, 0, 1),
])
def test_conv_fused_vs_oracle(lib, shape):
    B, Ca, Cb, Cout, H, W, k, rs, act = shape
    tag = "t/conv/" + "_".join(map(str, shape))
    Cin = Ca + Cb
    Hs, Ws = (H // 2, W // 2) if rs == 1 else ((H * 2, W * 2) if rs == 2 else (H, W))
    xa = fx.randn(tag + "/xa", B, Ca, Hs, Ws)
    xb = fx.randn(tag + "/xb", B, Cb, Hs, Ws) if Cb else None
    w = fx.param(tag, "conv.weight", (Cout, Cin, k, k))
    b = fx.param(tag, "conv.bias", (Cout,))
    coef = torch.stack([fx.randn(tag + "/mean", B, Cin) * 0.3, 1 + 0.3 * fx.randn(tag + "/scale", B, Cin),
                        0.2 * fx.randn(tag + "/off", B, Cin), torch.zeros(B, Cin)], dim=-1)
    res = fx.randn(tag + "/res", B, Cout, H, W)
    wpk, bpk = lib.op_pack_conv(dev(w), dev(b))
    y = lib.op_conv(dev(xa), dev(xb) if Cb else None, wpk, bpk, Cout, k, coef=dev(coef), act=act, resample=rs,
                    res=dev(res), res_mode=0)
    x = torch.cat([xa, xb], 1) if Cb else xa
    ref = orc.conv2d(apply_coef(x, coef, act), w, b, up=(rs == 1), down=(rs == 2)) + res
    close(y, ref, what=tag)


@pytest.mark.parametrize("shape", [
    # (B, Cin, Cout, Hs, Ws, with_transform): 8x32 / 8x16 / 8x8 pixel tiles, channel tile 64 and 32, ragged edges
    (2, 64, 64, 128, 128, 0), (2, 64, 64, 64, 64, 0), (3, 64, 64, 16, 16, 0), (2, 40, 24, 20, 12, 1), (1, 128, 128, 48, 40, 1),
    (2, 64, 64, 104, 72, 0),
])
def test_conv_stride2_ddpm_downsample(lib, shape):
    """Downsample of the DDPM U-Net (models/ddim_blocks.py:85-104): pad (0, 1, 0, 1) then 3x3 stride 2, on the 4-phase
    LDS tile of conv_s2_mfma_kernel; optionally with the fused input transform."""
    B, Cin, Cout, Hs, Ws, tr = shape
    tag = "t/s2/" + "_".join(map(str, shape))
    x = fx.randn(tag + "/x", B, Cin, Hs, Ws)
    w, b = fx.param(tag, "conv.weight", (Cout, Cin, 3, 3)), fx.param(tag, "conv.bias", (Cout,))
    coef = None
    xin = x
    if tr:
        coef = torch.stack([fx.randn(tag + "/mean", B, Cin) * 0.3, 1 + 0.3 * fx.randn(tag + "/scale", B, Cin),
                            0.2 * fx.randn(tag + "/off", B, Cin), torch.zeros(B, Cin)], dim=-1)
        xin = apply_coef(x, coef, True)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xin, (0, 1, 0, 1)), w, b, stride=2)
    wpk, bpk = lib.op_pack_conv(dev(w), dev(b))
    y = lib.op_conv(dev(x), None, wpk, bpk, Cout, 3, coef=dev(coef) if tr else None, act=tr, resample=lib.RS_S2)
    assert tuple(y.shape) == tuple(ref.shape)
    close(y, ref, what=tag)


TILES = [(128, 8, 32), (64, 8, 32), (32, 8, 32), (128, 16, 16), (64, 16, 16), (32, 16, 16), (128, 8, 16), (64, 8, 16),
         (128, 8, 8), (64, 8, 8), (32, 8, 8), (128, 16, 32)]


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("k", [3, 1])
def test_conv_every_tile_configuration(lib, tile, k):
    # each template instantiation of conv_mfma_kernel on the same ragged problem (40x24 image: partial tiles
    # in both directions; 136 input channels: a zero-padded last K chunk; virtual concat 72|64)
    if tile == (128, 16, 32) and k == 1:
        pytest.skip("the 8-wave kernel is 3x3 only")
    B, Ca, Cb, Cout, H, W = 2, 72, 64, 128, 40, 24
    tag = f"t/tiles/k{k}"
    xa, xb = fx.randn(tag + "/xa", B, Ca, H, W), fx.randn(tag + "/xb", B, Cb, H, W)
    w, b = fx.param(tag, "conv.weight", (Cout, Ca + Cb, k, k)), fx.param(tag, "conv.bias", (Cout,))
    coef = torch.stack([fx.randn(tag + "/mean", B, Ca + Cb) * 0.3, 1 + 0.3 * fx.randn(tag + "/scale", B, Ca + Cb),
                        0.2 * fx.randn(tag + "/off", B, Ca + Cb), torch.zeros(B, Ca + Cb)], dim=-1)
    res = fx.randn(tag + "/res", B, Cout, H, W)
    wpk, bpk = lib.op_pack_conv(dev(w), dev(b))
    ref = orc.conv2d(apply_coef(torch.cat([xa, xb], 1), coef, True), w, b) + res
    lib.set_conv_tile(*tile)
    try:
        y = lib.op_conv(dev(xa), dev(xb), wpk, bpk, Cout, k, coef=dev(coef), act=1, res=dev(res))
    finally:
        lib.set_conv_tile()
    close(y, ref, what=f"tile {tile} k={k}")
    if tile == (128, 16, 32):     # same accumulation order as the 4-wave kernel: identical bits
        lib.set_conv_tile(128, 8, 32)
        try:
            y4 = lib.op_conv(dev(xa), dev(xb), wpk, bpk, Cout, k, coef=dev(coef), act=1, res=dev(res))
        finally:
            lib.set_conv_tile()
        assert torch.equal(y, y4)


@pytest.mark.parametrize("shape", [
    # (B, Ca, Cb, Cout, H, W, k, resample, res_mode): the shapes conv_resident_kernel serves (<= 32 x 32 images, Cout % 64 == 0)
    (3, 64, 0, 64, 8, 8, 3, 0, 0), (2, 64, 64, 64, 8, 8, 3, 0, 0), (2, 64, 0, 64, 16, 16, 3, 0, 0), (1, 128, 128, 128, 16, 16, 3, 0, 0),
    (2, 64, 0, 64, 16, 16, 3, 1, 1), (2, 64, 0, 64, 8, 8, 3, 0, 2), (3, 2, 2, 64, 32, 32, 3, 0, 0), (2, 64, 0, 64, 32, 32, 3, 0, 0),
    (2, 64, 64, 64, 32, 32, 3, 0, 0), (2, 64, 0, 64, 32, 32, 3, 1, 1), (2, 64, 0, 192, 8, 8, 1, 0, 0), (1, 128, 0, 384, 16, 16, 1, 0, 0),
    (2, 64, 0, 64, 16, 16, 1, 0, 0), (2, 24, 12, 64, 12, 10, 3, 0, 0), (3, 72, 0, 64, 8, 8, 3, 0, 0), (2, 40, 24, 96, 6, 8, 3, 0, 0),
    (5, 128, 128, 128, 8, 8, 3, 0, 1), (2, 64, 0, 2, 32, 32, 3, 0, 0), (3, 128, 0, 3, 24, 32, 3, 0, 0), (1, 72, 0, 64, 28, 30, 3, 0, 0), (2, 256, 0, 128, 16, 16, 3, 0, 0), (1, 320, 64, 64, 16, 16, 3, 0, 0), (2, 128, 128, 128, 32, 32, 3, 0, 0),
    # big tiles (>= 64 x 64 images, one 64-channel output tile): 8 x 32 pixels at 64^2, 16 x 32 at 128^2; multi-pass K,
    # virtual concat, ragged height, padded last chunk, residual at the output / half / double resolution
    (2, 64, 0, 64, 64, 64, 3, 0, 0), (2, 64, 64, 64, 64, 64, 3, 0, 2), (1, 64, 0, 64, 128, 128, 3, 0, 0), (1, 64, 64, 64, 128, 128, 3, 0, 1),
    (1, 72, 0, 64, 60, 72, 3, 0, 0), (1, 40, 24, 64, 136, 128, 3, 0, 0), (2, 4, 0, 64, 128, 128, 3, 0, 0),
])
def test_conv_resident_kernel_is_bit_identical_to_the_tiled_one(lib, shape):
    """conv_resident.hip (whole K extent of the tile in LDS, DMA weight stream) against conv_mfma_kernel on the same tile
    configuration: identical bits (same accumulation order), and both against the oracle.  Ragged images, padded last
    chunks, virtual concat, 2x up-sampled input, residual at the output / half / double resolution."""
    B, Ca, Cb, Cout, H, W, k, rs, rm = shape
    tag = "t/resident/" + "_".join(map(str, shape))
    Cin = Ca + Cb
    Hs, Ws = (H // 2, W // 2) if rs == 1 else (H, W)
    xa = fx.randn(tag + "/xa", B, Ca, Hs, Ws)
    xb = fx.randn(tag + "/xb", B, Cb, Hs, Ws) if Cb else None
    w, b = fx.param(tag, "conv.weight", (Cout, Cin, k, k)), fx.param(tag, "conv.bias", (Cout,))
    coef = torch.stack([fx.randn(tag + "/mean", B, Cin) * 0.3, 1 + 0.3 * fx.randn(tag + "/scale", B, Cin),
                        0.2 * fx.randn(tag + "/off", B, Cin), torch.zeros(B, Cin)], dim=-1)
    Hr, Wr = (H // 2, W // 2) if rm == 1 else ((2 * H, 2 * W) if rm == 2 else (H, W))
    res = fx.randn(tag + "/res", B, Cout, Hr, Wr)
    wpk, bpk = lib.op_pack_conv(dev(w), dev(b))
    args = (dev(xa), dev(xb) if Cb else None, wpk, bpk, Cout, k)
    kw = dict(coef=dev(coef), act=1, resample=rs, res=dev(res), res_mode=rm)
    lib.set_conv_resident(1)
    lib.prof_enable(True)
    try:
        y = lib.op_conv(*args, **kw)
        names = [r["name"] for r in lib.prof_report()]
    finally:
        lib.prof_enable(False)
        lib.set_conv_resident(-1)
    assert any(n.startswith("conv_resident_kernel") for n in names), names
    lib.set_conv_resident(0)
    try:
        y0 = lib.op_conv(*args, **kw)
    finally:
        lib.set_conv_resident(-1)
    if k == 3 and H <= 8 and W <= 8 and rs == 0:      # K-split tiles: the K sum is grouped per wave, so only close to the tiled kernel
        assert any("ResCfg<32, 4, 8, 1, 1, 9, 8, 4>" in n for n in names), names
        close(y, y0.cpu(), rtol=1e-5, atol=1e-5, what=tag + " (K-split vs tiled)")
    else:
        assert torch.equal(y, y0), f"{tag}: max |d| = {(y - y0).abs().max().item():.3e}"
    x = torch.cat([xa, xb], 1) if Cb else xa
    r = res
    if rm == 1:
        r = res.repeat_interleave(2, 2).repeat_interleave(2, 3)
    elif rm == 2:
        r = torch.nn.functional.avg_pool2d(res, 2)
    close(y, orc.conv2d(apply_coef(x, coef, True), w, b, up=(rs == 1)) + r, what=tag)


def test_unet_forward_resident_on_off(lib, net_P):
    """Whole 32 x 32 U-Net (fused GroupNorm statistics, folded 1x1 skip projections): the forward with the input-resident
    kernels against the forward with conv_mfma_kernel everywhere.  (Bit-identical per tile configuration except at the
    8 x 8 level, where the resident kernel splits the K loop over its waves.)"""
    plan, packed, P = net_P
    x, cond = fx.randn("unet_P/x", 4, 2, 32, 32), fx.randn("unet_P/cond", 4, 2, 32, 32)
    labels = dev(fx.UNET_LABELS["nB"])
    F1 = plan.forward(packed, dev(x), labels, cond=dev(cond))
    lib.set_conv_resident(0)
    try:
        F0 = plan.forward(packed, dev(x), labels, cond=dev(cond))
    finally:
        lib.set_conv_resident(-1)
    close(F1, F0.cpu(), rtol=1e-5, atol=2e-6, what="resident on / off")


def test_unet_forward_winograd_on_off(lib, golden, net_P):
    """The reference's ch = 64 network at 32 x 32: its 32 x 32 level runs on the 64-channel Winograd kernel by default
    (csrc/conv_wino.hip, WinoCfg<2>).  Both that forward and the direct-kernel forward must meet the reference's golden output
    at the north_star tolerance, and each other."""
    plan, packed, P = net_P
    x, cond = fx.randn("unet_P/x", 4, 2, 32, 32), fx.randn("unet_P/cond", 4, 2, 32, 32)
    labels = dev(fx.UNET_LABELS["nB"])
    lib.prof_enable(True)
    try:
        F1 = plan.forward(packed, dev(x), labels, cond=dev(cond))
        torch.cuda.synchronize()
        names = [r["name"] for r in lib.prof_report()]
    finally:
        lib.prof_enable(False)
    assert any(n.startswith("conv_wino_kernel<WinoCfg<2>") for n in names), names
    lib.set_conv_wino(0)
    try:
        F0 = plan.forward(packed, dev(x), labels, cond=dev(cond))
    finally:
        lib.set_conv_wino(-1)
    g = golden("unet_P.npz")["F_nB"]
    close(F1, g, what="winograd forward vs the reference")
    close(F0, g, what="direct forward vs the reference")
    close(F1, F0.cpu(), what="winograd on / off")


def test_unet_32x32_batch_shard_invariance(lib, net_P):
    """Every kernel of the 32 x 32 network (input-resident convs in all three tilings incl. the K-split one, multi-pass
    staging, fused attention block, fused statistics) is chosen by shape only and computes each sample on its own: a batch
    and its shards give identical bits (exact multi-GPU batch sharding), and a rerun is reproducible."""
    plan, packed, P = net_P
    x, cond = fx.randn("t/shard32/x", 6, 2, 32, 32), fx.randn("t/shard32/cond", 6, 2, 32, 32)
    lab = dev(torch.tensor([0.3]))
    full = plan.forward(packed, dev(x), lab, cond=dev(cond))
    again = plan.forward(packed, dev(x), lab, cond=dev(cond))
    assert torch.equal(full, again)
    parts = [plan.forward(packed, dev(x[i:j]), lab, cond=dev(cond[i:j])) for i, j in ((0, 1), (1, 3), (3, 6))]
    assert torch.equal(full, torch.cat(parts))


def test_unet_forward_fused_attention_block_vs_three_launches(lib, net_P):
    """attn_block64_kernel (GroupNorm + qkv + softmax attention + proj + residual of an 8 x 8 x 64 block in one launch) against
    the qkv conv / attention kernel / proj conv sequence on the whole 32 x 32 U-Net, and both against the oracle."""
    plan, packed, P = net_P
    x, cond = fx.randn("unet_P/x", 4, 2, 32, 32), fx.randn("unet_P/cond", 4, 2, 32, 32)
    labels = dev(fx.UNET_LABELS["nB"])
    lib.prof_enable(True)
    try:
        F1 = plan.forward(packed, dev(x), labels, cond=dev(cond))
        names = [r["name"] for r in lib.prof_report()]
    finally:
        lib.prof_enable(False)
    assert "attn_block64_kernel" in names and "attention_kernel" not in names, names
    lib.set_attn_fused(0)
    try:
        F0 = plan.forward(packed, dev(x), labels, cond=dev(cond))
    finally:
        lib.set_attn_fused(-1)
    close(F1, F0.cpu(), what="fused attention block vs three launches")
    close(F1, orc.unet_forward(P, fx.CFG_P, x, fx.UNET_LABELS["nB"], cond), what="fused attention block vs oracle")


@pytest.mark.parametrize("Cout", [1, 2, 3, 4])
def test_output_conv_direct_kernel(lib, Cout):
    # ch -> out_channels 3x3 conv on a large image takes the direct (non-MFMA) kernel: ragged 72 x 88 image,
    # 44 input channels (zero-padded last chunk), GroupNorm/SiLU transform rows, bias, no residual
    B, Cin, H, W = 2, 44, 72, 88
    tag = f"t/outconv/{Cout}"
    x = fx.randn(tag + "/x", B, Cin, H, W)
    w, b = fx.param(tag, "conv.weight", (Cout, Cin, 3, 3)), fx.param(tag, "conv.bias", (Cout,))
    coef = torch.stack([fx.randn(tag + "/mean", B, Cin) * 0.3, 1 + 0.3 * fx.randn(tag + "/scale", B, Cin),
                        0.2 * fx.randn(tag + "/off", B, Cin), torch.zeros(B, Cin)], dim=-1)
    wpk, bpk = lib.op_pack_conv(dev(w), dev(b))
    y = lib.op_conv(dev(x), None, wpk, bpk, Cout, 3, coef=dev(coef), act=1)
    close(y, orc.conv2d(apply_coef(x, coef, True), w, b), what=f"output conv Cout={Cout}")
    lib.set_conv_tile(32, 8, 32)      # the matrix path on the same problem
    try:
        y2 = lib.op_conv(dev(x), None, wpk, bpk, Cout, 3, coef=dev(coef), act=1)
    finally:
        lib.set_conv_tile()
    close(y, y2.cpu(), what="direct vs MFMA")


def test_conv_null_source_is_zero(lib):
    # cond=None => the cond half of cat(cond, x) reads as zeros (adm_blocks.py:328-331)
    x = fx.randn("t/null/x", 2, 2, 16, 16)
    w, b = fx.param("t/null", "conv.weight", (64, 4, 3, 3)), fx.param("t/null", "conv.bias", (64,))
    wpk, bpk = lib.op_pack_conv(dev(w), dev(b))
    lb = lib._bind_ops()
    out = torch.empty(2, 64, 16, 16, device="cuda")
    lib.check(lb.mcedm_op_conv(None, lib._ptr(dev(x)), 2, 2, None, 0, 0, 0, 16, 16, 16, 16, lib._ptr(wpk), lib._ptr(bpk),
                               None, 0, lib._ptr(out), 64, 2, 3, lib._stream()))
    close(out, orc.conv2d(torch.cat([torch.zeros_like(x), x], 1), w, b), what="null cond")


# ------------------------------------------------------------------ K5 attention
def packed_qkv(qkv, heads):
    B, C3, H, W = qkv.shape
    d = C3 // heads // 3
    return qkv.reshape(B, heads, d, 3, H, W).permute(0, 1, 3, 2, 4, 5).reshape(B, C3, H, W).contiguous()


@pytest.mark.parametrize("T", list(fx.ATTN_CASES))
def test_attention_golden(lib, golden, T):
    B, hw = fx.ATTN_CASES[T]
    qkv = fx.randn(f"ops/attn{T}/qkv", B, 384, *hw)
    a = lib.op_attention(dev(packed_qkv(qkv, 2)), 2)
    close(a, golden("ops.npz")[f"attn{T}_a"], what=f"attn{T}")


@pytest.mark.parametrize("B,heads,hw,scale", [(2, 1, (32, 32), 1.0), (1, 2, (4, 4), 1.0), (1, 1, (6, 6), 1.0),
                                              (1, 1, (2, 2), 1.0), (2, 1, (8, 8), 6.0), (1, 2, (22, 26), 1.0),
                                              (1, 1, (32, 20), 2.0), (1, 1, (23, 23), 1.0)])
def test_attention_vs_oracle(lib, B, heads, hw, scale):
    # ragged token counts (36, 4), one full 1024-token case, and a peaked-softmax case (scale 6); 572 and 640 tokens take the
    # LDS-staged long-sequence kernel with a ragged last key tile / query block, 529 (not a multiple of 4) must not
    qkv = fx.randn(f"t/attn/{B}{heads}{hw}{scale}", B, heads * 192, *hw) * scale
    a = lib.op_attention(dev(packed_qkv(qkv, heads)), heads)
    close(a, orc.attention(qkv, heads), what="attention")


def test_attention_online_softmax_rescale(lib):
    # force the running max to jump in a late key tile (guide rule 26): one huge key in the last tile
    qkv = fx.randn("t/attn/spike", 1, 192, 8, 16)
    q, k, v = qkv.reshape(1, 64, 3, 128).unbind(2)
    k[0, :, 120] = q[0, :, 5] * 4.0
    a = lib.op_attention(dev(packed_qkv(qkv, 1)), 1)
    close(a, orc.attention(qkv, 1), what="attention spike")
    # the same in the long-sequence kernel: the spike sits in the last of 20 key tiles
    qkv = fx.randn("t/attn/spike640", 1, 192, 20, 32)
    q, k, v = qkv.reshape(1, 64, 3, 640).unbind(2)
    k[0, :, 630] = q[0, :, 133] * 4.0
    a = lib.op_attention(dev(packed_qkv(qkv, 1)), 1)
    close(a, orc.attention(qkv, 1), what="attention spike, 640 tokens")


# ------------------------------------------------------------------ whole blocks out of the ops
def hip_block(L, P, spec, x, emb):
    """adm_blocks.py:159-181 composed from the kernel-level entry points (mirrors csrc/plan.hip run_block)."""
    k = spec.key
    g = lambda n: dev(P[f"{k}.{n}"])
    n_emb = emb.shape[0]
    film = dev(orc.linear(emb, P[f"{k}.affine.weight"], P[f"{k}.affine.bias"]))
    rs = L.RS_UP if spec.up else (L.RS_DOWN if spec.down else L.RS_NONE)
    xd = dev(x)
    c0 = L.op_gn_coef(xd, None, g("norm0.weight"), g("norm0.bias"))
    w0, b0 = L.op_pack_conv(g("conv0.weight"), g("conv0.bias"))
    h = L.op_conv(xd, None, w0, b0, spec.cout, 3, coef=c0, act=1, resample=rs)
    c1 = L.op_gn_coef(h, None, g("norm1.weight"), g("norm1.bias"), film=film, film_batch=int(n_emb > 1),
                      film_stride=2 * spec.cout)
    res, mode = xd, L.RS_NONE
    if spec.skip_kernel == 1:
        ws, bs = L.op_pack_conv(g("skip.weight"), g("skip.bias"))
        res = L.op_conv(xd, None, ws, bs, spec.cout, 1, resample=rs)
    elif spec.skip_kernel == 0:
        mode = rs
    w1, b1 = L.op_pack_conv(g("conv1.weight"), g("conv1.bias"))
    y = L.op_conv(h, None, w1, b1, spec.cout, 3, coef=c1, act=1, res=res, res_mode=mode)
    if not spec.attn:
        return y
    c2 = L.op_gn_coef(y, None, g("norm2.weight"), g("norm2.bias"))
    wq, bq = L.op_pack_conv(g("qkv.weight"), g("qkv.bias"), qkv_heads=spec.heads)
    qkv = L.op_conv(y, None, wq, bq, 3 * spec.cout, 1, coef=c2)
    a = L.op_attention(qkv, spec.heads)
    wp, bp = L.op_pack_conv(g("proj.weight"), g("proj.bias"))
    return L.op_conv(a, None, wp, bp, spec.cout, 1, res=y)


@pytest.mark.parametrize("tag", list(fx.BLOCK_CASES))
@pytest.mark.parametrize("n_emb", [1, 2])
def test_unet_block_golden(lib, golden, tag, n_emb):
    x, emb = fx.block_inputs(tag, n_emb)
    y = hip_block(lib, fx.block_params(tag), fx.block_spec(tag), x, emb)
    close(y, golden("blocks.npz")[f"{tag}_n{n_emb}_y"], what=f"block {tag}")


# ------------------------------------------------------------------ whole network / preconditioning
def make_plan(L, cfg):
    return L.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks,
                  cfg.attn_resolutions, cfg.resolution)


@pytest.fixture(scope="module")
def net_P(lib):
    plan = make_plan(lib, fx.CFG_P)
    P = orc.make_params(fx.CFG_P, 7)
    packed = plan.pack({k: dev(v) for k, v in P.items()})
    return plan, packed, P


def test_unet_forward_golden(lib, golden, net_P):
    plan, packed, P = net_P
    g = golden("unet_P.npz")
    x, cond = fx.randn("unet_P/x", 4, 2, 32, 32), fx.randn("unet_P/cond", 4, 2, 32, 32)
    ws = lib.Workspace()
    for tag, labels in fx.UNET_LABELS.items():
        F = plan.forward(packed, dev(x), dev(labels), cond=dev(cond), ws=ws)
        close(F, g[f"F_{tag}"], what=f"unet {tag}")
    close(plan.forward(packed, dev(x), dev(torch.tensor([0.3])), cond=None, ws=ws), g["F_nocond"], what="unet nocond")
    # training-mode workspace (nothing freed) must give the same numbers
    F = plan.forward(packed, dev(x), dev(fx.UNET_LABELS["nB"]), cond=dev(cond), ws=ws, training=True)
    close(F, g["F_nB"], what="unet nB training layout")


def test_unet_forward_wide_golden(lib, golden):
    plan = make_plan(lib, fx.CFG_W)
    packed = plan.pack({k: dev(v) for k, v in orc.make_params(fx.CFG_W, 11).items()})
    F = plan.forward(packed, dev(fx.randn("unet_W/x", 2, 2, 16, 16)), dev(fx.UNET_W_LABELS),
                     cond=dev(fx.randn("unet_W/cond", 2, 2, 16, 16)))
    close(F, golden("unet_W.npz")["F"], what="unet wide")


def test_model_precond_golden(lib, golden, net_P):
    plan, packed, P = net_P
    g = golden("unet_P.npz")
    x, cond = fx.randn("unet_P/x", 4, 2, 32, 32), fx.randn("unet_P/cond", 4, 2, 32, 32)
    for i, s in enumerate(fx.PRECOND_SIGMAS):
        D = plan.denoise(packed, dev(x * (1 + s)), dev(torch.tensor([s])), cond=dev(cond))
        close(D, g[f"D_sigma{i}"], what=f"precond sigma={s}")
    D, F = plan.denoise(packed, dev(x), dev(fx.PRECOND_SIGMA_B), cond=dev(cond), want_F=True)
    close(D, g["D_sigmaB"], what="precond sigma[B]")
    c_skip, c_out, c_in, c_noise = orc.precond_coeffs(fx.PRECOND_SIGMA_B.reshape(-1, 1, 1, 1))
    close(F, orc.unet_forward(P, fx.CFG_P, c_in * x, c_noise.flatten(), cond), what="F_x")


def test_fused_groupnorm_statistics_with_large_means(lib):
    """activations whose mean dwarfs their spread (|mean| / std ~ 50 at the conv_in output, ~20-40
    after two residual blocks) through the FUSED statistics path (producer epilogue records -> consumer).  The records
    are (sum, M2 about the tile mean) merged in fp64, so rstd keeps full precision; E[x^2] - E[x]^2 on fp32 partial
    sums loses 1e-3 here.  Both consumers are covered: in-kernel rows (inference) and the table kernel (training layout)."""
    plan = make_plan(lib, fx.CFG_P)
    P = {k: v.clone() for k, v in orc.make_params(fx.CFG_P, 7).items()}
    P["enc.128x128_conv.bias"] = 30.0 + 0.05 * fx.randn("t/bigmean/b0", 64)
    P["enc.128x128_block0.conv1.bias"] = -20.0 + 0.05 * fx.randn("t/bigmean/b1", 64)
    P["enc.64x64_block0.conv1.bias"] = 25.0 + 0.05 * fx.randn("t/bigmean/b2", 64)
    P["dec.64x64_up.conv0.bias"] = 40.0 + 0.05 * fx.randn("t/bigmean/b3", 64)
    packed = plan.pack({k: dev(v) for k, v in P.items()})
    x, cond = fx.randn("t/bigmean/x", 3, 2, 40, 24), fx.randn("t/bigmean/c", 3, 2, 40, 24)     # ragged tiles too
    lab = torch.tensor([0.4])
    with torch.no_grad():
        ref = orc.unet_forward(P, fx.CFG_P, x, lab, cond)
        t0 = orc.conv2d(torch.cat([cond, x], 1), P["enc.128x128_conv.weight"], P["enc.128x128_conv.bias"])
    ratio = float(t0.mean().abs() / t0.reshape(3, 16, -1).std(dim=-1).mean())
    assert ratio > 40, ratio
    close(plan.forward(packed, dev(x), dev(lab), cond=dev(cond)), ref, what=f"large-mean fused GN (|mean|/std = {ratio:.0f})")
    close(plan.forward(packed, dev(x), dev(lab), cond=dev(cond), training=True), ref, what="large-mean GN, training layout")


def test_embedding_kernel_golden(lib, golden):
    """A1 at op level: PositionalEmbedding (golden pe_y) -> mapping MLP -> the blocks' affine rows, one launch (K6)."""
    ch, rows = 64, 2 * (64 + 128 + 64)
    lab = fx.PE_LABELS
    w0, b0 = fx.param("t/emb", "map_layer0.weight", (ch, ch)), fx.param("t/emb", "map_layer0.bias", (ch,))
    w1, b1 = fx.param("t/emb", "map_layer1.weight", (ch, ch)), fx.param("t/emb", "map_layer1.bias", (ch,))
    wa, ba = fx.param("t/emb", "affine.weight", (rows, ch)), fx.param("t/emb", "affine.bias", (rows,))
    pe = torch.as_tensor(golden("ops.npz")["pe_y"])                      # the reference's PositionalEmbedding output
    e_ref = torch.nn.functional.silu(orc.linear(torch.nn.functional.silu(orc.linear(pe, w0, b0)), w1, b1))
    emb, film = lib.op_embedding(dev(lab), dev(w0), dev(b0), dev(w1), dev(b1), dev(wa), dev(ba))
    close(emb, e_ref, what="emb (mapping MLP on the golden positional embedding)")
    close(film, orc.linear(e_ref, wa, ba), what="film rows")
    # the positional embedding alone: identity layers expose it (silu is monotone; compare through its inverse-free form)
    eye, zero = torch.eye(ch), torch.zeros(ch)
    emb_id, _ = lib.op_embedding(dev(lab), dev(eye), dev(zero), dev(eye), dev(zero), dev(wa), dev(ba))
    silu = torch.nn.functional.silu
    close(emb_id, silu(silu(pe)), what="positional embedding through identity layers")
    # n = 1 (sampling) takes the split-row grid
    emb1, film1 = lib.op_embedding(dev(lab[2:3]), dev(w0), dev(b0), dev(w1), dev(b1), dev(wa), dev(ba))
    close(film1, orc.linear(e_ref[2:3], wa, ba), what="film rows n=1")


def test_unet_rectangular_and_odd_batch(lib, net_P):
    # T=64 x X=32 variant of BASELINE config 2 (net only needs H, W divisible by 4), batch 3
    plan, packed, P = net_P
    x, cond = fx.randn("t/rect/x", 3, 2, 64, 32), fx.randn("t/rect/c", 3, 2, 64, 32)
    lab = torch.tensor([0.2, -0.4, 0.9])
    with torch.no_grad():
        ref = orc.unet_forward(P, fx.CFG_P, x, lab, cond)
    close(plan.forward(packed, dev(x), dev(lab), cond=dev(cond)), ref, what="rect")


def test_shape_errors_are_rejected_on_host(lib, net_P):
    plan, packed, P = net_P
    with pytest.raises(RuntimeError, match="multiples of 4"):
        plan.workspace_bytes(2, 30, 32)
    with pytest.raises(RuntimeError, match="n_noise"):
        plan.forward(packed, dev(torch.zeros(4, 2, 32, 32)), dev(torch.zeros(3)))
    small = lib.Workspace()
    small.buf = torch.empty(1024, dtype=torch.uint8, device="cuda")
    small.get = lambda n, d: small.buf
    with pytest.raises(RuntimeError, match="workspace too small"):
        plan.forward(packed, dev(torch.zeros(4, 2, 32, 32)), dev(torch.zeros(1)), ws=small)


# ------------------------------------------------------------------ Heun sampler
@pytest.mark.parametrize("tag", list(fx.SAMPLER_CASES))
def test_sample_edm_golden(lib, golden, net_P, tag):
    plan, packed, P = net_P
    g = golden("sampler_P.npz")
    cond, m, init, steps = fx.sampler_inputs(tag)
    churn = fx.SAMPLER_CASES[tag][0]
    sd = lib.sampler_desc(orc.SamplerParams(S_churn=churn))
    step_noise = dev(torch.stack(steps)) if churn > 0 else None
    xs = plan.sample(packed, sd, dev(cond), dev(m), dev(init), step_noise, return_last=False)
    assert xs.dtype == torch.float64 and tuple(xs.shape) == (4, 19, 32, 32, 2)
    close(xs[:, -1:], g[f"{tag}_xs_last"], rtol=1e-4, atol=1e-5, what=f"sampler {tag} last")
    close(xs[:, ::6], g[f"{tag}_xs_traj"], rtol=1e-4, atol=1e-5, what=f"sampler {tag} trajectory")
    last = plan.sample(packed, sd, dev(cond), dev(m), dev(init), step_noise, return_last=True)
    assert tuple(last.shape) == (4, 1, 32, 32, 2) and torch.equal(last[:, 0], xs[:, -1])
    obs = (m == 0).permute(0, 2, 3, 1)
    assert torch.equal(last[:, 0].cpu()[obs], cond.permute(0, 2, 3, 1).double()[obs]), "observed entries must be kept exactly"


def test_sampler_requires_step_noise_when_churning(lib, net_P):
    plan, packed, P = net_P
    cond, m, init, steps = fx.sampler_inputs("churn_u")
    sd = lib.sampler_desc(orc.SamplerParams(S_churn=15.0))
    with pytest.raises(RuntimeError, match="step_noise"):
        plan.sample(packed, sd, dev(cond), dev(m), dev(init), None)


# ------------------------------------------------------------------ training-side elementwise kernels
def test_noise_inputs_loss_sqnorm_adam(lib, golden):
    g = golden("training_P.npz")
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    xc, cond_in, mc = fx.training_nchw(h, u, mask, cond_noise)
    x_noise, sigma = lib.edm_noise_inputs(dev(xc), dev(mc), dev(noise), dev(rnd_normal.flatten()))
    sig_ref = (rnd_normal * orc.P_STD + orc.P_MEAN).exp()
    close(sigma, sig_ref.flatten(), what="sigma")
    close(x_noise, xc + mc * noise * sig_ref, what="x_noise")
    D = fx.randn("t/loss/D", *xc.shape)
    loss, dD = lib.edm_loss(dev(D), dev(xc), dev(mc), dev(sig_ref.flatten()))
    Dg = D.clone().requires_grad_(True)
    ref = (orc.loss_weight(sig_ref) * (Dg * mc - xc * mc) ** 2).sum(dim=(1, 2, 3)).mean()
    ref.backward()
    close(loss, ref.detach().reshape(1), what="loss")
    close(dD, Dg.grad, what="dD")
    # sqnorm + Adam/EMA on the golden grads
    names = fx.TRAIN_GRAD_NAMES
    P = orc.make_params(fx.CFG_P, int(g["seed"]))
    flat_g = torch.cat([torch.as_tensor(g[f"grad::{n}"]).flatten() for n in names])
    flat_p = torch.cat([P[n].flatten() for n in names])
    sq = lib.sqnorm(dev(flat_g))
    assert abs(float(sq) - float((flat_g.double() ** 2).sum())) <= 1e-6 * float(sq)   # fp32 per-thread partials
    # feed the TRUE total norm so the clip factor equals the reference's
    sq_true = torch.tensor([float(g["clip_total_norm"]) ** 2], dtype=torch.float64).cuda()
    p, m, v, e = dev(flat_p), torch.zeros_like(dev(flat_p)), torch.zeros_like(dev(flat_p)), dev(flat_p).clone()
    lib.adam_ema_step(p, dev(flat_g), m, v, e, step=1, sqnorm_t=sq_true)
    close(p, torch.cat([torch.as_tensor(g[f"adam::{n}"]).flatten() for n in names]), rtol=1e-5, atol=1e-7, what="adam")
    close(e, torch.cat([torch.as_tensor(g[f"ema::{n}"]).flatten() for n in names]), rtol=1e-5, atol=1e-7, what="ema")
