import os, sys, torch, traceback
sys.path.insert(0, "/root/repo")
import mcedm_amd
from mcedm_amd import lib
from oracle import mcedm_oracle as orc
CASES = [
    (dict(ch=64, ch_mult=(1, 2), attn_resolutions=(16,)), 1, 64, 32),
    (dict(ch=64, ch_mult=(2, 2, 1), attn_resolutions=()), 3, 32, 32),
    (dict(ch=128, ch_mult=(1, 1), attn_resolutions=(32, 16)), 5, 32, 32),
    (dict(ch=64, ch_mult=(1, 2, 2, 2), attn_resolutions=(4,), num_res_blocks=1), 2, 32, 32),
    (dict(ch=64, ch_mult=(1, 3), attn_resolutions=()), 2, 16, 48),
    (dict(ch=192, ch_mult=(1,), attn_resolutions=(32,)), 2, 32, 32),
]
bad = 0
for kw, B, H, W in CASES:
    try:
        cfg = orc.UNetConfig(**kw)
        plan = lib.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks, cfg.attn_resolutions, cfg.resolution)
        P = orc.make_params(cfg, 11)
        packed = plan.pack({k: v.cuda() for k, v in P.items()})
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, 2, H, W, generator=g) * 2; cond = torch.randn(B, 2, H, W, generator=g)
        sig = torch.rand(B, generator=g) * 5 + 0.1
        D = plan.denoise(packed, x.cuda(), sig.cuda(), cond=cond.cuda())
        with torch.no_grad():
            ref = orc.model_precond(P, cfg, x, sig, cond)
        err = (D.cpu() - ref).abs().max().item(); sc = ref.abs().max().item()
        ok = err <= 1e-5 + 1e-4 * sc
        bad += not ok
        print(("ok  " if ok else "FAIL"), kw, (B, H, W), f"max err {err:.2e} (scale {sc:.2f})", flush=True)
    except Exception as e:
        bad += 1
        print("ERR ", kw, (B, H, W), repr(e)[:300], flush=True)
sys.exit(1 if bad else 0)
