"""fp32 error of Winograd F(4x4, 3x3) against F(2x2, 3x3) and the direct form (CPU experiment for DESIGN.md section 8):
max / rms error vs an fp64 convolution on a 128 -> 128 layer with unit-variance input and U(-1,1)/sqrt(fan_in) weights.
The Winograd-domain products are accumulated over cin in fp32 like the MFMA does (sequential fp32 sum of fp32 products)."""
import numpy as np, torch
torch.manual_seed(0)
Cin = Cout = 128
H = W = 32
x = torch.randn(1, Cin, H, W, dtype=torch.float64)
w = (torch.rand(Cout, Cin, 3, 3, dtype=torch.float64) * 2 - 1) / (Cin * 9) ** 0.5
ref = torch.nn.functional.conv2d(x, w, padding=1)
direct32 = torch.nn.functional.conv2d(x.float(), w.float(), padding=1).double()

def mats(m):
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
    else:   # Lavin & Gray F(4x4, 3x3), points 0, +-1, +-2, inf
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                       [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], float)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)
    return [torch.tensor(a, dtype=torch.float32) for a in (BT, G, AT)]

def wino(m):
    BT, G, AT = mats(m)
    a = m + 2
    xp = torch.nn.functional.pad(x.float(), (1, 1, 1, 1))
    U = torch.einsum("ij,ocjk,lk->ocil", G, w.float(), G)                       # [o, c, a, a] fp32
    out = torch.zeros(1, Cout, H, W, dtype=torch.float32)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = xp[0, :, ty:ty + a, tx:tx + a]
            V = torch.einsum("ij,cjk,lk->cil", BT, d, BT)                       # [c, a, a]
            M = torch.zeros(Cout, a, a, dtype=torch.float32)
            for c in range(Cin):                                                # sequential fp32 accumulation
                M += U[:, c] * V[c]
            out[0, :, ty:ty + m, tx:tx + m] = torch.einsum("ij,ojk,lk->oil", AT, M, AT)
    return out.double()

scale = float(ref.abs().max())
for name, y in (("direct fp32", direct32), ("F(2x2,3x3)", wino(2)), ("F(4x4,3x3)", wino(4))):
    e = (y - ref).abs()
    print(f"{name:12s} max|err| {float(e.max()):.3e}  rms {float((e ** 2).mean().sqrt()):.3e}  (max|y| {scale:.2f}; max relative to max|y| {float(e.max()) / scale:.2e})")
