"""fp32 error of the Winograd-domain weight gradient F(3x3, 2x2) against the direct form (CPU experiment).

    dW[co][ci][a][b] = sum_{n, y, x} dY[n][co][y][x] * X'[n][ci][y + a - 1][x + b - 1]

Direct form (csrc/wgrad_mfma.hip): per split a sequential fp32 sum over its pixels, the splits added in fp64.
Winograd form (csrc/wgrad_wino.hip): per 2x2 output tile  Yd = G dY G^T (4x4), Xd = B^T X' B (4x4) in fp32, per position and split a
sequential fp32 sum over the split's tiles, the splits added in fp64, then  dW = A^T M A  in fp64.
Both are compared with an fp64 evaluation on (i) unit-variance noise and (ii) a smooth field with a large mean in X' (the
SiLU output of a GroupNorm has mean ~0.2-0.3; a large mean is where the transformed sums could cancel).
The bar gradients are held to: |err| <= 1e-4 |g| + 1e-5 max|g|."""
import numpy as np
import torch

torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)     # 4x4 (input, as F(2x2,3x3))
G = torch.tensor([[1, 0], [.5, .5], [.5, -.5], [0, 1]], dtype=torch.float64)                              # 4x2 (dY)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, 0], [0, 1, 1, 1]], dtype=torch.float64)                       # 3x4 (output)
# B^T's last row is (0, 1, 0, -1); the matching A^T last column for F(3, 2) with that sign convention:
AT[:, 3] = torch.tensor([0, 0, -1], dtype=torch.float64)


def check_identity():
    d = torch.randn(4, dtype=torch.float64); g = torch.randn(2, dtype=torch.float64)
    y = AT @ ((G @ g) * (BT @ d))
    ref = torch.stack([g[0] * d[k] + g[1] * d[k + 1] for k in range(3)])
    assert torch.allclose(y, ref, atol=1e-12), (y, ref)


def seq_sum32(p, dim):
    """sequential fp32 sum along `dim` (what an MFMA accumulator does)"""
    return torch.cumsum(p.float(), dim=dim, dtype=torch.float32).select(dim, -1)


def study(name, x, dy, split_tiles):
    """x [N, Ci, H, W], dy [N, Co, H, W] float64"""
    N, Ci, H, W = x.shape
    Co = dy.shape[1]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    # fp64 reference
    ref = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64)
    for a in range(3):
        for b in range(3):
            ref[:, :, a, b] = torch.einsum("noyx,ncyx->oc", dy, xp[:, :, a:a + H, b:b + W])
    # tiles: [T, C, 4, 4] and [T, C, 2, 2], T = N * H/2 * W/2 in (n, ty, tx) order
    xt = xp.unfold(2, 4, 2).unfold(3, 4, 2).permute(0, 2, 3, 1, 4, 5).reshape(-1, Ci, 4, 4)
    yt = dy.unfold(2, 2, 2).unfold(3, 2, 2).permute(0, 2, 3, 1, 4, 5).reshape(-1, Co, 2, 2)
    T = xt.shape[0]
    ns = T // split_tiles
    # direct fp32: per split, sequential over its pixels (tile by tile, 4 pixels each)
    d32 = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64)
    x32, y32 = xt.float(), yt.float()
    for a in range(3):
        for b in range(3):
            # products per pixel: [ns, split_tiles * 4, Co, Ci]
            xs = x32[:, :, a:a + 2, b:b + 2].reshape(ns, split_tiles, Ci, 4).permute(0, 1, 3, 2).reshape(ns, -1, Ci)
            ys = y32.reshape(ns, split_tiles, Co, 4).permute(0, 1, 3, 2).reshape(ns, -1, Co)
            prod = ys[:, :, :, None] * xs[:, :, None, :]
            d32[:, :, a, b] = seq_sum32(prod, 1).double().sum(0)
    # Winograd fp32
    Xd = torch.einsum("ij,tcjk,lk->tcil", BT.float(), x32, BT.float())        # [T, Ci, 4, 4]
    Yd = torch.einsum("ij,tcjk,lk->tcil", G.float(), y32, G.float())          # [T, Co, 4, 4]
    M = torch.zeros(Co, Ci, 4, 4, dtype=torch.float64)
    for i in range(4):
        for l in range(4):
            prod = Yd[:, :, i, l].reshape(ns, split_tiles, Co)[:, :, :, None] * Xd[:, :, i, l].reshape(ns, split_tiles, Ci)[:, :, None, :]
            M[:, :, i, l] = seq_sum32(prod, 1).double().sum(0)
    w32 = torch.einsum("ij,ocjk,lk->ocil", AT, M, AT)
    # exactness of the algorithm itself in fp64
    Xd64 = torch.einsum("ij,tcjk,lk->tcil", BT, xt, BT)
    Yd64 = torch.einsum("ij,tcjk,lk->tcil", G, yt, G)
    w64 = torch.einsum("ij,ocjk,lk->ocil", AT, torch.einsum("toil,tcil->ocil", Yd64, Xd64), AT)
    assert torch.allclose(w64, ref, rtol=1e-10, atol=1e-9 * float(ref.abs().max()))
    gmax = float(ref.abs().max())
    bar = 1e-4 * ref.abs() + 1e-5 * gmax
    print(f"{name}: T = {T} tiles, {ns} splits of {split_tiles}; max|g| {gmax:.3f}")
    for tag, w in (("direct fp32", d32), ("Winograd F(3x3,2x2) fp32", w32)):
        e = (w - ref).abs()
        print(f"   {tag:26s} max|err| {float(e.max()):.3e}  rms {float((e ** 2).mean().sqrt()):.3e}  worst err / bar {float((e / bar).max()):.3f}")


if __name__ == "__main__":
    check_identity()
    N, C, H, W = 4, 12, 128, 128
    x = torch.nn.functional.silu(torch.randn(N, C, H, W, dtype=torch.float64))
    dy = torch.randn(N, C, H, W, dtype=torch.float64) * 1e-3
    study("SiLU(noise) x noise", x, dy, 2048)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H, dtype=torch.float64), torch.linspace(0, 1, W, dtype=torch.float64), indexing="ij")
    ph = torch.rand(N, C, 1, 1, dtype=torch.float64) * 6.28
    xs = 5.0 + torch.sin(6.28 * (2 * xx + yy) + ph) + 0.05 * torch.randn(N, C, H, W, dtype=torch.float64)
    dys = (torch.cos(6.28 * (xx - 3 * yy) + ph.flip(1)) + 0.3) * 1e-3 + 1e-4 * torch.randn(N, C, H, W, dtype=torch.float64)
    study("smooth field, mean 5, correlated gradient", xs, dys, 2048)
    study("smooth field, mean 5, noise gradient", xs, dy, 2048)
    study("SiLU(noise) x noise, long splits", x, dy, 8192)
