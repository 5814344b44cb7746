"""Per-kernel timing of one training step (event-pair profiler of libmcedm_hip)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcedm_amd  # noqa
from mcedm_amd import lib
from mcedm_amd.train import views_like
from oracle import mcedm_oracle as orc

wl = sys.argv[1] if len(sys.argv) > 1 else "s128"
cfg = orc.UNetConfig(ch=128, ch_mult=(1, 1, 1, 1), attn_resolutions=(16,)) if wl == "s128" else orc.UNetConfig()
B, H, W = (32, 128, 128) if wl in ("s128", "ref128") else (64, 32, 32)
dev = torch.device("cuda")
plan = lib.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks, cfg.attn_resolutions, cfg.resolution)
params = {k: v.to(dev) for k, v in orc.make_params(cfg, 7).items()}
packed = plan.pack(params)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 2, H, W, generator=g).to(dev); nz = torch.randn(B, 2, H, W, generator=g).to(dev); rn = torch.randn(B, generator=g).to(dev)
mask = torch.zeros(B, 2, H, W, device=dev); mask[:, 1] = 1
cond = x * (1 - mask) + torch.randn(B, 2, H, W, generator=g).to(dev) * mask
grads = [torch.zeros_like(params[n]) for n in plan.param_names]
ws = lib.Workspace()
def step():
    xn, sg = lib.edm_noise_inputs(x, mask, nz, rn)
    D = plan.denoise(packed, xn, sg, cond=cond, ws=ws, training=True)
    loss, dD = lib.edm_loss(D, x, mask, sg)
    plan.denoise_backward(packed, params, xn, sg, cond, dD, grads, ws)
step(); torch.cuda.synchronize()
lib.prof_enable(True); step(); torch.cuda.synchronize(); lib.prof_enable(False)
rows = sorted(lib.prof_report(), key=lambda r: -r["total_ms"])
tot = sum(r["total_ms"] for r in rows)
print(f"profiled kernel time {tot:.2f} ms")
for r in rows[:16]:
    print(f"{r['name'][:62]:62s} n={r['launches']:4d} {r['total_ms']:8.3f} ms  {r['flops']/r['total_ms']/1e9:7.1f} TF/s {r['bytes']/r['total_ms']/1e6:8.1f} GB/s")
