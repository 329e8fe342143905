"""Every distinct convolution problem of the bench step (kradar.json, batch 4: the rows of the committed per-shape conv
table profiles/r0N_conv_table_fp32.txt, written by `DPFT_CONV_TABLE=... python bench.py`) -- forward, data gradient and
weight gradient through the C-ABI vs fp64 F.conv2d, in the operand forms the ResNet plan / FPN use them with (BN + ReLU
operand prologue, BN tile statistics, bias, accumulating data gradient).  Pins the dispatch table (tile shape, split-K,
parity-class / K-split / thin-input variants are chosen by problem size) at the sizes the headline number is measured on."""
import glob
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table_rows():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_conv_table_fp32.txt")))
    assert files, "no committed conv table under profiles/"
    kinds = {}
    with open(files[-1]) as f:
        next(f)
        for ln in f:
            p = ln.split()
            if len(p) < 8:
                continue
            key = tuple(int(v) for v in p[1:8])            # B H W C K k s
            kinds.setdefault(key, set()).add(p[0])
    # biggest problems last so that a failure on a small one shows up early
    return sorted(kinds.items(), key=lambda kv: kv[0][0] * kv[0][1] * kv[0][2] * kv[0][3] * kv[0][4] * kv[0][5] ** 2)


ROWS = _table_rows()


def _close(a, ref, what, rtol=1e-4, atol_scale=2e-5):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    atol = atol_scale * max(float(ref.abs().max()), 1e-6)
    torch.testing.assert_close(a, ref, rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")
    return float((a - ref).norm() / (ref.norm() + 1e-30))


@pytest.mark.parametrize("shape,kinds", ROWS, ids=["x".join(str(v) for v in k) for k, _ in ROWS])
def test_bench_step_conv_problem_vs_fp64(shape, kinds):
    from dpft_amd.hip import ops
    B, H, W, C, K, k, s = shape
    pad = k // 2
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    g = torch.Generator().manual_seed(sum(v * (i + 3) for i, v in enumerate(shape)) % 9973)
    x = torch.randn(B, H, W, C, generator=g)
    w = (torch.randn(K, C, k, k, generator=g) / (C * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    body = C % 32 == 0 and K % 32 == 0                     # a ResNet-body conv: BN+ReLU prologue, BN statistics epilogue
    bias = torch.randn(K, generator=g) if K == 16 else None     # FPN convs carry a bias
    pro = None
    if body:
        pro = (torch.stack((torch.randn(C, generator=g) * 0.5, torch.rand(C, generator=g) + 0.5,
                            torch.randn(C, generator=g) * 0.3, torch.ones(C))), True)
    xd = x.double()
    if pro is not None:
        xd = ((xd - pro[0][0].double()) * pro[0][1].double() + pro[0][2].double()).clamp_min(0)
    xa = xd.permute(0, 3, 1, 2).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    yref = F.conv2d(xa, wd, None if bias is None else bias.double(), stride=s, padding=pad)
    dy = torch.randn(yref.shape, generator=g, dtype=torch.float64)
    (yref * dy).sum().backward()

    cv = ops.conv_problem(B, H, W, C, K, k, k, s, pad)
    assert (cv.OH, cv.OW) == tuple(yref.shape[2:])
    xg, wg = x.to(DEV), w.to(DEV).permute(0, 2, 3, 1)
    prod = None if pro is None else (pro[0].to(DEV), True)
    errs = {}
    y, stats = ops.conv_fwd(cv, xg, wg, bias=None if bias is None else bias.to(DEV), pro=prod, want_stats=body)
    errs["fwd"] = _close(y.permute(0, 3, 1, 2), yref, "fwd")
    if stats is not None:
        ones, zeros = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
        bnp = ops.bn_finalize(stats, cv.tile_rows, cv.M, ones, zeros, 1e-5, 0.1, zeros.clone(), ones.clone())
        yr = yref.detach().permute(0, 2, 3, 1).reshape(-1, K)
        _close(bnp[0], yr.mean(0), "bn mean", atol_scale=1e-4)
        _close(bnp[3], 1 / torch.sqrt(yr.var(0, unbiased=False) + 1e-5), "bn invstd")
    dyg = dy.permute(0, 2, 3, 1).contiguous().float().to(DEV)
    dw = ops.conv_wgrad(cv, xg, dyg, pro=prod)
    errs["wgrad"] = _close(dw.permute(0, 3, 1, 2), wd.grad, "wgrad")
    if "dgrad" in kinds or C >= 16:
        wt = ops.weight_transpose(wg)
        dx = ops.conv_dgrad(cv, dyg, wt)
        errs["dgrad"] = _close(dx.permute(0, 3, 1, 2), xa.grad, "dgrad")
        base = torch.randn(B, H, W, C, generator=g).to(DEV)            # accumulating form (downsample branch)
        acc = ops.conv_dgrad(cv, dyg, wt, out=base.clone(), accumulate=True)
        _close((acc - base).permute(0, 3, 1, 2), xa.grad, "dgrad accumulate", atol_scale=1e-4)
    print(shape, sorted(kinds), {k_: f"{v:.1e}" for k_, v in errs.items()})
    assert all(v < 2e-6 for v in errs.values()), errs            # fp32-roundoff class (the fp32 MFMA is an exact fmaf chain)


def _body_dgrad_rows():
    """dgrad rows of the table that are ResNet-body convs (the plan runs them with a BatchNorm-backward reduction in the
    epilogue): C % 64 == 0 and K % 64 == 0."""
    rows = [(k, kinds) for k, kinds in ROWS if "dgrad" in kinds and k[3] % 64 == 0 and k[4] % 64 == 0]
    # not in the bench step: row counts that are NOT multiples of the tile height (batch 3), so that the kernels that request
    # their epilogue operands with the first tile (EpiPrefetch, round 4) also run with partly filled last tiles
    rows += [((3, 32, 57, 1024, 256, 1, 1), {"dgrad"}), ((3, 33, 57, 256, 256, 3, 1), {"dgrad"}), ((3, 33, 57, 256, 1024, 1, 1), {"dgrad"})]
    return rows


@pytest.mark.parametrize("shape,kinds", _body_dgrad_rows(), ids=["x".join(str(v) for v in k) for k, _ in _body_dgrad_rows()])
def test_bench_step_dgrad_with_bn_reduce_epilogue_vs_fp64(shape, kinds):
    """the fp32 data gradient + BatchNorm-backward-reduce epilogue (and, for stride-1 1x1 convs, the
    identity-branch ReLU backward folded into the same epilogue) at the bench step's shapes -- the form
    dpft_resnet_backward_stage runs -- vs fp64: dx, and sums = (sum d, sum d * xhat) of d = dx under the ReLU mask, for both
    mask sources (byte mask of a block output / bn(y) > 0).  Where the launch cannot carry the reduction (split-K) the
    flag says so and sums stays untouched."""
    from dpft_amd.hip import ops
    B, H, W, C, K, k, s = shape
    pad = k // 2
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    g = torch.Generator().manual_seed(sum(v * (i + 5) for i, v in enumerate(shape)) % 9973)
    w = (torch.randn(K, C, k, k, generator=g) / (C * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    cv = ops.conv_problem(B, H, W, C, K, k, k, s, pad)
    dy = torch.randn(B, cv.OH, cv.OW, K, generator=g)
    dx_ref = torch.nn.grad.conv2d_input((B, C, H, W), w.double(), dy.double().permute(0, 3, 1, 2), stride=s, padding=pad) \
        .permute(0, 2, 3, 1).contiguous()
    bn_y = torch.randn(B, H, W, C, generator=g) * 1.5 + 0.3
    mean, invstd = torch.randn(C, generator=g) * 0.4, torch.rand(C, generator=g) + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    block = torch.stack((mean, gamma * invstd, beta, invstd)).contiguous()
    xhat = (bn_y.double() - mean.double()) * invstd.double()
    wt = ops.weight_transpose(w.to(DEV).permute(0, 2, 3, 1))
    cases = [("self-mask", None, None)]
    mask_bits = torch.rand(B, H, W, C, generator=g) > 0.4
    packed = (mask_bits.view(B, H, W, C // 4, 4).to(torch.uint8) * torch.tensor([1, 2, 4, 8], dtype=torch.uint8)).sum(-1) \
        .to(torch.uint8).contiguous()
    cases.append(("byte mask", packed, None))
    if s == 1 and k == 1:                                   # conv1 of a bottleneck: identity branch in the same epilogue
        res_src = torch.randn(B, H, W, C, generator=g)
        block_out = torch.where(torch.rand(B, H, W, C, generator=g) > 0.5, torch.rand(B, H, W, C, generator=g) + 0.1,
                                torch.zeros(B, H, W, C))
        rm = ((block_out > 0).view(B, H, W, C // 4, 4).to(torch.uint8) * torch.tensor([1, 2, 4, 8], dtype=torch.uint8)).sum(-1) \
            .to(torch.uint8).contiguous()
        cases.append(("byte mask + identity branch", packed, (res_src, block_out, rm)))
    for name, m8, res in cases:
        ref = dx_ref if res is None else dx_ref + torch.where(res[1] > 0, res[0].double(), torch.zeros((), dtype=torch.float64))
        mask = mask_bits if m8 is not None else ((xhat * gamma.double() + beta.double()) > 0)
        # self-mask elements whose bn(y) is within fp32 round-off of zero may flip: leave them out of the reference sums
        sure = torch.ones_like(mask) if m8 is not None else ((xhat * gamma.double() + beta.double()).abs() > 1e-5)
        d = torch.where(mask, ref, torch.zeros((), dtype=torch.float64))
        s_ref = torch.stack((d.sum((0, 1, 2)), (d * xhat).sum((0, 1, 2))))
        slack = torch.stack(((ref.abs() * ~sure).sum((0, 1, 2)), (ref.abs() * xhat.abs() * ~sure).sum((0, 1, 2))))
        sums = torch.zeros(2, C, device=DEV)
        rg = None if res is None else (res[0].to(DEV), res[1].to(DEV), res[2].to(DEV))
        dx, applied = ops.conv_dgrad_bn_reduce(cv, dy.to(DEV), wt, bn_y.to(DEV), block.to(DEV), sums,
                                               bn_mask8=None if m8 is None else m8.to(DEV), residual=rg)
        e = _close(dx, ref, f"dgrad [{name}]")
        assert e < 2e-6, (name, e)
        if applied:
            got = sums.double().cpu()
            scale = torch.stack((d.abs().sum((0, 1, 2)), (d * xhat).abs().sum((0, 1, 2))))      # cancellation-aware bound
            err = ((got - s_ref).abs() - slack).clamp_min(0) / scale.clamp_min(1e-30)
            assert float(err.max()) < 2e-6, (name, float(err.max()))
        else:
            assert float(sums.abs().max()) == 0.0, name
        print(shape, name, "applied" if applied else "not carried (split-K / classes)", f"dx {e:.1e}")
