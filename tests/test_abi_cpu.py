"""CPU-only checks of the C ABI (no compute calls): libmcedm_hip.so loads without a GPU, exports exactly the symbols
include/mcedm_hip.h declares, and its host-side entry points (plan construction, parameter table, sizes, the sigma
schedule, argument validation) behave as documented."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

import mcedm_amd  # noqa: F401
from mcedm_amd import lib as L
from oracle import mcedm_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcedm_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcedm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = declared_symbols()
    assert len(names) >= 36
    for n in names:
        getattr(lib, n)                                  # AttributeError == header / library drift
    nm = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\bT (mcedm_[a-z0-9_]+)$", nm, flags=re.M)))
    assert exported == names, (set(exported) ^ set(names))
    assert sorted(set(L.EXPORTS + L.OP_EXPORTS)) == names, set(L.EXPORTS + L.OP_EXPORTS) ^ set(names)
    assert lib.mcedm_version() == 4
    # the header's constants the binding mirrors
    hdr = open(HEADER).read()
    consts = {k: int(v) for k, v in re.findall(r"#define (MCEDM_[A-Z0-9_]+) (\d+)\b", hdr)}
    assert consts["MCEDM_ABI_VERSION"] == L.ABI_VERSION == 4
    assert consts["MCEDM_GN_SYNC_WORDS"] == L.GN_SYNC_WORDS
    for name, idx in L.VARIANTS.items():
        assert consts["MCEDM_VARIANT_" + name.upper()] == idx, name


def test_plan_parameter_table_matches_state_dict_order():
    for cfg in (orc.UNetConfig(), orc.UNetConfig(ch=128, ch_mult=(1, 1, 1, 1), attn_resolutions=(16,))):
        plan = L.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks,
                      cfg.attn_resolutions, cfg.resolution)
        ref = orc.param_shapes(cfg)
        assert plan.param_names == [n for n, _ in ref]
        assert plan.param_shapes == [tuple(s) for _, s in ref]
        assert plan.packed_bytes > 4 * sum(int(torch.tensor(s).prod()) for _, s in ref)
        assert plan.workspace_bytes(2, 32, 32) < plan.workspace_bytes(2, 32, 32, training=True)
        assert plan.sampler_workspace_bytes(2, 32, 32) > plan.workspace_bytes(2, 32, 32)


def test_dx_cond_plans_list_the_head_parameters_in_state_dict_order():
    """dx_cond (adm_blocks.py:233-280): cat_dx widens conv_in; dx_enc / combine_enc are registered before self.enc."""
    import dataclasses
    for mode, code in (("cat", L.DX_CAT), ("enc", L.DX_ENC)):
        cfg = dataclasses.replace(orc.UNetConfig(in_channels=1, cond_channels=1, out_ch=1), dx_channels=1, dx_mode=mode)
        plan = L.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks,
                      cfg.attn_resolutions, cfg.resolution, dx_channels=1, dx_mode=code)
        ref = orc.param_shapes(cfg)
        assert plan.param_names == [n for n, _ in ref] and plan.param_shapes == [tuple(s) for _, s in ref]
    assert plan.param_names[4:10] == ["dx_enc.0.weight", "dx_enc.0.bias", "dx_enc.2.weight", "dx_enc.2.bias",
                                      "combine_enc.weight", "combine_enc.bias"]
    with pytest.raises(RuntimeError, match="dx_channels"):
        L.Plan(1, 1, 1, 64, (1, 1, 1), 1, (32,), 128, dx_channels=1, dx_mode=L.DX_NONE)


def test_t_steps_host_helper_matches_oracle():
    sd = L.sampler_desc(orc.SamplerParams(timesteps=18))
    t = torch.tensor(L.edm_t_steps(sd), dtype=torch.float64)
    torch.testing.assert_close(t, orc.edm_t_steps(18, 0.002, 80.0, 7.0), rtol=1e-14, atol=0)
    assert t[-1] == 0 and t[0] == 80.0


@pytest.mark.parametrize("ch,mult,attn", [(96, (1, 1), ()), (32, (1, 5), ()), (64, (1, 1, 1), (64,)), (160, (1,), ())])
def test_attention_width_not_a_multiple_of_64_is_rejected(ch, mult, attn):
    """head_dim = C / (C // 64) in the reference (adm_blocks.py:135,175); only head_dim 64 is built, so a
    width like 96 or 160 at an attention block must be refused, not computed with 64-wide heads."""
    width_bad = any((ch * m) % 64 and ch * m >= 64 for lv, m in enumerate(mult) if lv == len(mult) - 1 or (128 >> lv) in attn)
    if width_bad:
        with pytest.raises(RuntimeError, match="channels_per_head"):
            L.Plan(2, 2, 2, ch, mult, 1, attn, 128)
    else:
        L.Plan(2, 2, 2, ch, mult, 1, attn, 128)


def test_host_side_argument_validation():
    with pytest.raises(RuntimeError, match="multiple of 8"):
        L.Plan(2, 2, 2, 12, (1,), 1, (), 128)
    plan = L.Plan(2, 2, 2, 64, (1, 1, 1), 1, (32,), 128)
    with pytest.raises(RuntimeError, match="multiples of 4"):
        plan.workspace_bytes(1, 30, 32)
    sd = L.sampler_desc(orc.SamplerParams(timesteps=1))
    with pytest.raises(RuntimeError, match="timesteps"):
        L.edm_t_steps(sd)
    with pytest.raises(RuntimeError, match="CPU tensor"):
        plan.forward(torch.zeros(8), torch.zeros(1, 2, 32, 32), torch.zeros(1))


def test_dx_entry_points_validate_before_any_launch():
    """The dx_cond entry points reject a dx tensor / a dx-conditioned sampler on a plan built without dx_cond, and a wrong
    dx shape, on the host (no GPU in this container: the checks run before anything is enqueued)."""
    import ctypes as C
    lib = L.load()
    plain = L.Plan(1, 1, 1, 64, (1, 1, 1), 1, (32,), 128)
    x = torch.zeros(1, 1, 32, 32)
    with pytest.raises(RuntimeError, match="without dx_cond"):
        plain.denoise(torch.zeros(8), x, torch.ones(1), cond=x, dx=x)
    dxp = L.Plan(1, 1, 1, 64, (1, 1, 1), 1, (32,), 128, dx_channels=1, dx_mode=L.DX_ENC)
    with pytest.raises(RuntimeError, match=r"dx must be fp32 \[B, 1, H, W\]"):
        dxp.denoise(torch.zeros(8), x, torch.ones(1), cond=x, dx=torch.zeros(1, 2, 32, 32))
    sd = L.sampler_desc(orc.SamplerParams(timesteps=18))
    gd = L.GuidanceDesc(1, 0.002, 0.03125, 0.0, 0.0, 1.0, 0.0, 1.0, 5.0)
    one = C.c_void_p(16)          # never dereferenced: the plan check comes first
    rc = lib.mcedm_heun_sample_dxcond(plain._h, one, C.byref(sd), C.byref(gd), None, one, one, None, one, 1, one, 1 << 30, 1, 32, 32, None)
    assert rc == -1 and b"dx_cond plan" in lib.mcedm_last_error()
    assert dxp.workspace_bytes(2, 32, 32) > plain.workspace_bytes(2, 32, 32)      # the head's tensors are in the layout


def test_repaint_schedule_and_ddpm_plan_host_side():
    """f1 host pieces: the DDPM parameter table in Model.state_dict() order and the rounded sigma schedule
    (round_sigma of models/ddim.py:949-957 restated as an exact nearest-neighbour search) against the oracle's torch.cdist."""
    from oracle import ddpm_oracle as dorc
    from oracle import fixtures as fx
    cfg = dorc.DdpmConfig()                      # configs/model/ddim_res32.yaml: resolution 128, attention at 32
    plan = L.DdpmPlan(cfg.in_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks, cfg.attn_resolutions, cfg.resolution)
    ref = dorc.param_shapes(cfg)
    assert plan.param_names == [n for n, _ in ref] and plan.param_shapes == [tuple(s) for _, s in ref]
    assert plan.repaint_workspace_bytes(2) > plan.workspace_bytes(2) > 0
    betas = dorc.betas_of(cfg)
    steps, aext = dorc.edm_steps_of(betas), dorc.alphas_ext_of(betas)
    for N, churn in ((18, 0.0), (50, 15.0), (5, 0.0)):
        sp = dorc.RepaintParams(timesteps=N, S_churn=churn)
        rd, keep = L.repaint_desc(sp, steps, aext, 1, 1)
        t = torch.tensor(L.repaint_schedule(rd), dtype=torch.float64)
        smin, smax = max(sp.sigma_min, float(steps[-1])), min(sp.sigma_max, float(steps[0]))
        idx = torch.arange(N, dtype=torch.float64)
        ts = (smax ** (1 / sp.rho) + idx / (N - 1) * (smin ** (1 / sp.rho) - smax ** (1 / sp.rho))) ** sp.rho
        want = torch.cat([dorc.round_sigma(steps, ts), torch.zeros(1, dtype=torch.float64)])
        assert torch.equal(t, want), (N, (t - want).abs().max())
    with pytest.raises(RuntimeError, match="only 64 is built"):
        L.DdpmPlan(2, 2, 64, (1, 2), 1, (), 32)
    with pytest.raises(RuntimeError, match="multiple of 32"):
        L.DdpmPlan(2, 2, 48, (1,), 1, (), 32)


def test_ddim_timestep_sequence_equals_numpy_linspace():
    """the quad sequence is [int(s) for s in np.linspace(0, sqrt(0.8 n), N) ** 2] (models/ddim.py:826-828).
    numpy computes arange(N) * (stop / (N - 1)) and pins the last sample to stop; hi * i / (N - 1) truncates to a different
    timestep for 85 of ~1200 (n, N) pairs (n = 1000, N = 100: 799 instead of 800).  Host helper vs numpy on a sweep."""
    import numpy as np
    bad = []
    for n in (100, 250, 500, 1000, 2000, 4000):
        for N in list(range(1, 260)) + [n]:
            if N > n:
                continue
            ref = [int(s) for s in list(np.linspace(0, np.sqrt(n * 0.8), N) ** 2)]
            got = L.ddim_timesteps(n, N, "quad")
            if got != ref:
                bad.append((n, N))
            assert L.ddim_timesteps(n, N, "uniform") == list(range(0, n, n // N))
    assert not bad, bad[:10]
    assert L.ddim_timesteps(1000, 100, "quad")[-1] == 800 and L.ddim_timesteps(1000, 16, "quad")[1:4] == [3, 14, 31]
    with pytest.raises(RuntimeError, match="bad schedule"):
        L.ddim_timesteps(10, 11, "quad")


def test_bench_roofline_helpers():
    """bench.py's host-side arithmetic (no GPU): both roofline fractions of SURVEY.md 8(d) per forward and per state, and the
    HBM-bound kernel rows."""
    import bench
    f = bench.fractions("s128", 76.0, 12.0, 32)
    gf, mb = bench.ALGORITHMIC["s128"]
    assert abs(f["forward_hbm_frac"] - mb * 1e6 * 32 / 12.0e-3 / 8e12) < 1e-12
    assert abs(f["forward_fp32_frac_algorithmic"] - gf * 1e9 * 32 / 12.0e-3 / 157.3e12) < 1e-12
    assert abs(f["per_state"]["gflop"] - 35 * gf) < 1e-9 and abs(f["per_state"]["hbm_frac"] - 76.0 * 35 * mb * 1e6 / 8e12) < 1e-12
    assert bench.fractions("repaint128", 1.0, 1.0, 1) == {}
    prof = [{"name": "gn_bwd_kernel", "launches": 45, "total_ms": 5.5, "flops": 2e10, "bytes": 1.2e10},
            {"name": "conv_wino_kernel<WinoCfg<4>, false>", "launches": 53, "total_ms": 18.0, "flops": 4e12, "bytes": 1e10}]
    rows = bench.hbm_bound_rows(prof)
    assert [r["name"] for r in rows] == ["gn_bwd_kernel"] and abs(rows[0]["hbm_frac"] - round(1.2e10 / 5.5e-3 / 1e9, 1) / 8000.0) < 1e-3
    ro = bench.roofline_of(prof)
    assert ro["kernel"].startswith("conv_wino") and abs(ro["executed_over_algorithmic"] - 4 / 9) < 1e-12 and ro["frac"] < ro["direct_equivalent_frac"]


def test_plan_variants_are_per_plan_and_validated():
    """mcedm_unet_plan_set_variant (ABI 4): a field of the plan, validated on the host.  With the Winograd kernels switched off a
    decoder block's 1x1 skip projection is folded into conv1 again, so that plan's workspace layout differs from its sibling's --
    visible without a GPU."""
    cfg = orc.UNetConfig(ch=128, ch_mult=(1, 1, 1, 1), attn_resolutions=(16,))
    mk = lambda: L.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks,
                        cfg.attn_resolutions, cfg.resolution)
    pa, pb = mk(), mk()
    base = pb.workspace_bytes(2, 64, 64)
    assert pa.workspace_bytes(2, 64, 64) == base
    pa.set_variant("conv_wino", 0)
    assert pa.workspace_bytes(2, 64, 64) != base and pb.workspace_bytes(2, 64, 64) == base
    pa.set_variant("conv_wino", -1)
    assert pa.workspace_bytes(2, 64, 64) == base
    lib = L.load()
    lib.mcedm_unet_plan_set_variant.argtypes = [C.c_void_p, C.c_int, C.c_int]
    assert lib.mcedm_unet_plan_set_variant(pa._h, 99, 1) != 0 and b"unknown switch" in lib.mcedm_last_error()
    assert lib.mcedm_unet_plan_set_variant(pa._h, 0, 7) != 0
