"""The 3 x bf16 split convolution kernels (dpft_amd/csrc/conv_x3.hip): fp32 tensors in, fp32 results out, every fp32 product
formed from six exact bf16 term products on the bf16 matrix cores.  What has to hold for the mode to count as fp32
(VERDICT r4 #1d): its error against fp64 is not above the fp32 MFMA path's on any convolution of the bench step it takes, the
planes it is fed add up to the fp32 value bit for bit, and ragged / strided / split-K forms agree with fp64 like the fp32 path."""
import glob
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    return float((a - ref).norm() / (ref.norm() + 1e-300))


def _problem(B, H, W, C, K, k, s, seed=0):
    g = torch.Generator().manual_seed(1000 + seed + B + H * 3 + W * 5 + C + K + k + s)
    x = torch.randn(B, H, W, C, generator=g)
    w = (torch.randn(K, k, k, C, generator=g) / (C * k * k) ** 0.5)
    pad = k // 2
    xa = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yref = F.conv2d(xa, w.double().permute(0, 3, 1, 2), stride=s, padding=pad)
    dy = torch.randn(yref.shape, generator=g, dtype=torch.float64)
    (yref * dy).sum().backward()
    return x, w, pad, yref.detach().permute(0, 2, 3, 1), dy.permute(0, 2, 3, 1).contiguous().float(), xa.grad.permute(0, 2, 3, 1)


def test_split_planes_add_up_to_the_fp32_value_bit_for_bit():
    from dpft_amd.hip import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1 << 16, generator=g)
    # every binade the step's tensors live in, signs, exact bf16 values, zeros, values one ulp off a bf16 value
    x = torch.cat([x * 1e-20, x * 1e-6, x, x * 3e4, x * 1e20, torch.zeros(64), -torch.zeros(64),
                   torch.tensor([1.0, -1.0, 0.5, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.0, 1.00390625, 3.0 * 2.0 ** -126])
                   .repeat(8)]).contiguous()
    x = x[: x.numel() // 4 * 4].to(DEV)
    p = ops.split_planes(x)
    assert p.dtype == torch.bfloat16 and tuple(p.shape) == (3, x.numel())
    s = p[0].double() + p[1].double() + p[2].double()
    assert torch.equal(s, x.double()), float((s - x.double()).abs().max())
    assert torch.equal(p[0], x.bfloat16())                                     # plane 0 = the RNE bf16 of the value
    assert float((p[1].double().abs() - 2.0 ** -8 * x.double().abs()).max()) <= 0.0     # each plane at most half an ulp of the one before
    assert float((p[2].double().abs() - 2.0 ** -16 * x.double().abs()).max()) <= 0.0


def _big_multitap_rows():
    """3x3 rows of the committed conv table that the split rule takes (forward and stride-1 data gradient, >= 2 GFLOP)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_conv_table_fp32.txt")))
    rows = set()
    with open(files[-1]) as f:
        next(f)
        for ln in f:
            p = ln.split()
            if len(p) < 8:
                continue
            B, H, W, C, K, k, s = (int(v) for v in p[1:8])
            if k > 1 and C % 64 == 0 and K % 64 == 0 and 2.0 * B * (H // s) * (W // s) * K * k * k * C >= 2e9:
                rows.add((B, H, W, C, K, k, s))
    return sorted(rows)


@pytest.mark.parametrize("shape", _big_multitap_rows(), ids=lambda s: "x".join(map(str, s)))
def test_split_mode_error_not_above_the_fp32_mfma_paths(shape):
    """Same problem, same operands (BatchNorm + ReLU prologue, tile statistics), split on / off: relative L2 error vs fp64 of the
    split kernels <= the fp32 MFMA kernels' (in fact ~3x below: one rounding per 16 reduction indices instead of one per 2)."""
    from dpft_amd.hip import ops
    B, H, W, C, K, k, s = shape
    x, w, pad, yref, dy, dxref = _problem(B, H, W, C, K, k, s)
    g = torch.Generator().manual_seed(3)
    bn = torch.stack((torch.randn(C, generator=g) * 0.5, torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3, torch.ones(C)))
    xd = ((x.double() - bn[0].double()) * bn[1].double() + bn[2].double()).clamp_min(0)
    w64 = w.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    yp = F.conv2d(xd.permute(0, 3, 1, 2), w64, stride=s, padding=pad)
    yp.backward(dy.double().permute(0, 3, 1, 2))
    yref_pro, dwref_pro = yp.detach().permute(0, 2, 3, 1), w64.grad.permute(0, 2, 3, 1)
    xg, wg, dyg, bng = x.to(DEV), w.to(DEV), dy.to(DEV), bn.to(DEV)
    wt = ops.weight_transpose(wg)
    err = {}
    try:
        for split in (False, True):
            ops.conv_set_split(split)
            cv = ops.conv_problem(B, H, W, C, K, k, k, s, pad)
            y, stats = ops.conv_fwd(cv, xg, wg, pro=(bng, True), want_stats=True)
            y0, _ = ops.conv_fwd(cv, xg, wg)
            e = {"fwd+prologue": _rel(y, yref_pro), "fwd": _rel(y0, yref)}
            ones, zeros = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
            bnp = ops.bn_finalize(stats, cv.tile_rows, cv.M, ones, zeros, 1e-5, 0.1, zeros.clone(), ones.clone())
            yr = yref_pro.reshape(-1, K)
            assert float((bnp[0].double().cpu() - yr.mean(0)).abs().max()) < 1e-5 * float(yr.abs().max())
            if s == 1:
                e["dgrad"] = _rel(ops.conv_dgrad(cv, dyg, wt), dxref)
            e["wgrad+prologue"] = _rel(ops.conv_wgrad(cv, xg, dyg, pro=(bng, True)), dwref_pro)      # (wgrad_x3_kernel where K, C >= 128)
            err[split] = e
    finally:
        ops.conv_set_split(True)
    print(shape, {k_: (f"{err[False][k_]:.2e}", f"{err[True][k_]:.2e}") for k_ in err[True]})
    for k_ in err[True]:
        # (the weight gradient sums over pixel splits in both modes: partial slabs added in fp32 -- the split products cannot be
        # better than that final sum, hence the looser factor there)
        lim = 1.25 if k_.startswith("wgrad") else 1.02
        assert err[True][k_] < 2e-6 and err[True][k_] <= err[False][k_] * lim + 1e-9, (k_, err[False][k_], err[True][k_])


@pytest.mark.parametrize("shape", [(3, 33, 57, 256, 256, 3, 1), (4, 32, 57, 256, 256, 3, 1), (2, 64, 114, 128, 128, 3, 1), (4, 16, 29, 512, 512, 3, 1),
                                   (4, 32, 57, 256, 1024, 1, 1), (3, 31, 57, 1024, 256, 1, 1), (4, 64, 114, 256, 256, 3, 2), (2, 9, 11, 64, 64, 3, 1),
                                   (4, 128, 228, 64, 64, 3, 1)])
@pytest.mark.parametrize("tile", [None, "128,128,2", "128,64,1", "64,64,3"])
def test_planes_operands_forward_and_data_gradient_vs_fp64(shape, tile, monkeypatch):
    """dpft_conv_desc::a_planes / w_planes: the GEMM reads both operands as three bf16 planes (igemm_x3p_kernel) -- ragged row
    counts, all three tiles, split-K with the in-launch fix-up, stride-2 parity classes, padding from the buffer range check."""
    from dpft_amd.hip import ops
    B, H, W, C, K, k, s = shape
    if tile:
        monkeypatch.setenv("DPFT_FORCE_TILE", tile)
    ops._conv_cache.clear()      # (workspace sizes follow the forced split count)
    x, w, pad, yref, dy, dxref = _problem(B, H, W, C, K, k, s, seed=7)
    xg, wg, dyg = x.to(DEV), w.to(DEV), dy.to(DEV)
    wt = ops.weight_transpose(wg)
    cv = ops.conv_problem(B, H, W, C, K, k, k, s, pad)
    xp, wp, dyp, wtp = (ops.split_planes(t) for t in (xg, wg, dyg, wt))
    y, stats = ops.conv_fwd(cv, xg, wg, want_stats=True, planes=(xp, wp))
    dx = ops.conv_dgrad(cv, dyg, wt, planes=(dyp, wtp))
    base = torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(2)).to(DEV)
    acc = ops.conv_dgrad(cv, dyg, wt, out=base.clone(), accumulate=True, planes=(dyp, wtp))
    e_y, e_dx, e_acc = _rel(y, yref), _rel(dx, dxref), _rel(acc - base, dxref)
    print(shape, tile, f"fwd {e_y:.2e} dgrad {e_dx:.2e}")
    assert e_y < 1e-6 and e_dx < 1e-6 and e_acc < 1e-5, (e_y, e_dx, e_acc)
    ones, zeros = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    bnp = ops.bn_finalize(stats, cv.tile_rows, cv.M, ones, zeros, 1e-5, 0.1, zeros.clone(), ones.clone())
    yr = yref.reshape(-1, K)
    assert float((bnp[0].double().cpu() - yr.mean(0)).abs().max()) < 1e-5 * float(yr.abs().max())
    # the planes path is deterministic: a second launch gives the same bits
    y2, _ = ops.conv_fwd(cv, xg, wg, want_stats=True, planes=(xp, wp))
    assert torch.equal(y, y2)


@pytest.mark.gpu
def test_eight_wave_split_kernel_passes_the_same_cases():
    """conv_x3w.hip (the split GEMM on eight waves per 128 x 128 tile, off by default: DESIGN.md section 3): every case of this file
    again in a process that has it switched on."""
    import os, subprocess, sys
    if os.environ.get("DPFT_X3W") == "1":
        pytest.skip("already the eight-wave run")
    env = dict(os.environ, DPFT_X3W="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "not eight_wave"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
