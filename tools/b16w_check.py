"""Parity + timing of the large-tile bf16 conv kernels (conv_b16w.hip, act16 = 2) against fp64 on the same bf16-valued operands.
Every form the mixed-precision plan launches: forward with BatchNorm tile statistics, plain / accumulating data gradient, the
data gradient that carries the next BatchNorm's backward reduction (self mask / byte mask) and the residual form of a
bottleneck's conv1.  Run with DPFT_B16W_MIN256 / DPFT_B16W_MIN128 lowered so that small problems take the big tiles, and with
DPFT_B16W=0 for the four-wave kernels on the same inputs.
   python tools/b16w_check.py [check|time] [B,H,W,C,K,k,s ...]          prints one line per (shape, form); exit 1 on a miss"""
import ctypes as C, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
from dpft_amd.hip.lib import lib, make_desc, ptr, stream

DEV = torch.device("cuda", 0)
CHECK = ["2,24,40,64,128,3,1", "2,24,40,128,256,1,1", "1,33,29,64,128,3,2", "2,32,57,256,256,3,1", "2,32,57,1024,256,1,1",
         "2,32,57,256,1024,1,1", "3,17,23,128,512,1,2", "1,9,7,192,384,3,1", "1,5,3,64,128,1,1", "4,64,114,128,128,3,1"]
TIME = ["8,32,57,256,256,3,1", "8,32,57,256,1024,1,1", "8,32,57,1024,256,1,1", "8,64,114,128,128,3,1", "8,64,114,128,512,1,1",
        "8,64,114,512,128,1,1", "8,128,228,64,256,1,1", "8,128,228,256,64,1,1", "8,16,29,512,512,3,1", "8,16,29,512,2048,1,1",
        "8,16,29,2048,512,1,1", "4,32,57,256,256,3,1", "4,32,57,256,1024,1,1", "4,32,57,1024,256,1,1"]


def pack_mask(bits):
    B, H, W, Cc = bits.shape
    return (bits.view(B, H, W, Cc // 4, 4).to(torch.uint8) * torch.tensor([1, 2, 4, 8], dtype=torch.uint8)).sum(-1).to(torch.uint8).contiguous()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(spec, check):
    B, H, W, Cc, K, k, s = map(int, spec.split(","))
    pad = k // 2
    d = make_desc(B, H, W, Cc, K, k, k, s, pad)
    d.act16 = 2
    g = torch.Generator().manual_seed(sum(v * (i + 3) for i, v in enumerate((B, H, W, Cc, K, k, s))) % 9973)
    x = torch.randn(B, H, W, Cc, generator=g).bfloat16()
    w = (torch.randn(K, k, k, Cc, generator=g) / (Cc * k * k) ** 0.5).bfloat16()
    dy = torch.randn(B, d.OH, d.OW, K, generator=g).bfloat16()
    wt = w.permute(3, 1, 2, 0).contiguous()
    xd, wd, dyd, wtd = x.to(DEV), w.to(DEV), dy.to(DEV), wt.to(DEV)
    tr = C.c_int32(0)
    tiles = int(lib.dpft_conv2d_stats_tiles(C.byref(d), C.byref(tr)))
    ws = torch.zeros(max(int(lib.dpft_conv2d_workspace_bytes(C.byref(d))), 16) + (1 << 20), dtype=torch.uint8, device=DEV)
    y = torch.empty(B, d.OH, d.OW, K, dtype=torch.bfloat16, device=DEV)
    stats = torch.zeros(tiles, 2, K, dtype=torch.float32, device=DEV)
    M, Mi = B * d.OH * d.OW, B * H * W
    fl = 2.0 * M * K * k * k * Cc
    bad = []

    def fwd():
        lib.call("dpft_conv2d_nhwc_fwd_f32", C.byref(d), ptr(xd), ptr(wd), None, None, 0, ptr(y), ptr(stats), ptr(ws), stream())
    dx = torch.full((B, H, W, Cc), 7.0, dtype=torch.bfloat16, device=DEV)

    def dgrad(acc=0):
        lib.call("dpft_conv2d_nhwc_dgrad_f32", C.byref(d), ptr(dyd), ptr(wtd), ptr(dx), acc, ptr(ws), stream())
    # fused forms: BatchNorm of the layer whose dout this data gradient writes (shape of dx)
    bn_y = (torch.randn(B, H, W, Cc, generator=g) * 1.5 + 0.3).bfloat16()
    mean, invstd = torch.randn(Cc, generator=g) * 0.4, torch.rand(Cc, generator=g) + 0.5
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    block = torch.stack((mean, gamma * invstd, beta, invstd)).contiguous().to(DEV)
    mask_bits = torch.rand(B, H, W, Cc, generator=g) > 0.4
    m8 = pack_mask(mask_bits).to(DEV)
    res_src = torch.randn(B, H, W, Cc, generator=g).bfloat16()
    block_out = torch.where(torch.rand(B, H, W, Cc, generator=g) > 0.5, torch.rand(B, H, W, Cc, generator=g) + 0.1, torch.zeros(B, H, W, Cc)).bfloat16()
    rm8 = pack_mask(block_out > 0).to(DEV)
    bn_yd, res_d, bo_d = bn_y.to(DEV), res_src.to(DEV), block_out.to(DEV)
    sums = torch.zeros(2, Cc, device=DEV)
    applied = C.c_int32(0)

    def fused(mask8, resid):
        lib.call("dpft_conv2d_nhwc_dgrad_bn_reduce_f32", C.byref(d), ptr(dyd), ptr(wtd), ptr(dx), 0,
                 ptr(res_d) if resid else None, ptr(bo_d) if resid else None, ptr(rm8) if resid else None, ptr(bn_yd), ptr(block),
                 ptr(mask8), int(mask8 is None), ptr(sums), C.addressof(applied), ptr(ws), stream())
    ops.conv_set_compute("bf16")
    try:
        if not check:
            t_f, t_d = timed(fwd), timed(dgrad)
            sums.zero_()
            t_m = timed(lambda: fused(m8, False)) if s == 1 else float("nan")
            t_r = timed(lambda: fused(m8, True)) if (s == 1 and k == 1) else float("nan")
            print(f"{spec:26s} rows/tile {tr.value:3d}  fwd+stats {t_f:7.1f} us {fl / t_f / 1e6:7.1f} TF | dgrad {t_d:7.1f} us {fl / t_d / 1e6:7.1f} TF | "
                  f"dgrad+bnr {t_m:7.1f} us | residual form {t_r:7.1f} us", flush=True)
            return bad
        rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
        elem = lambda a, b: float(((a.double().cpu() - b).abs() - 2.0 ** -8 * b.abs()).max() / b.abs().max())
        w64 = w.double().permute(0, 3, 1, 2)
        y_ref = F.conv2d(x.double().permute(0, 3, 1, 2), w64, stride=s, padding=pad).permute(0, 2, 3, 1)
        xin = torch.zeros(B, Cc, H, W, dtype=torch.float64, requires_grad=True)
        F.conv2d(xin, w64, stride=s, padding=pad).backward(dy.double().permute(0, 3, 1, 2))
        dx_ref = xin.grad.permute(0, 2, 3, 1)
        # forward + statistics (twice: the split-K tickets must be left clean)
        for rep in range(2):
            y.fill_(5.0)
            fwd()
            torch.cuda.synchronize()
            e1, e2 = rel(y, y_ref), elem(y, y_ref)
            cnt = torch.full((tiles,), float(tr.value), dtype=torch.float64)
            cnt[-1] = M - tr.value * (tiles - 1)
            st = stats.double().cpu()
            mu = (st[:, 0] * cnt[:, None]).sum(0) / M
            m2 = st[:, 1].sum(0) + (cnt[:, None] * (st[:, 0] - mu) ** 2).sum(0)
            yr = y_ref.reshape(-1, K)
            e3 = float((mu - yr.mean(0)).abs().max() / yr.abs().max())
            e4 = float((m2 / M - yr.var(0, unbiased=False)).abs().max() / yr.var(0).max())
            ok = e1 < 2.5e-3 and e2 < 1e-4 and e3 < 1e-5 + 1e-6 and e4 < 1e-4
            print(f"{spec:26s} rows/tile {tr.value:3d} fwd#{rep}   rel {e1:.1e} elem {e2:.1e} mean {e3:.1e} var {e4:.1e} {'ok' if ok else 'MISS'}", flush=True)
            if not ok:
                bad.append((spec, "fwd"))
        # plain and accumulating data gradient
        dx.fill_(7.0)
        dgrad(0)
        torch.cuda.synchronize()
        e1, e2 = rel(dx, dx_ref), elem(dx, dx_ref)
        ok = e1 < 2.5e-3 and e2 < 1e-4
        print(f"{spec:26s} dgrad        rel {e1:.1e} elem {e2:.1e} {'ok' if ok else 'MISS'}", flush=True)
        if not ok:
            bad.append((spec, "dgrad"))
        base = torch.randn(B, H, W, Cc, generator=g).bfloat16()
        dx.copy_(base.to(DEV))
        dgrad(1)
        torch.cuda.synchronize()
        ref = dx_ref + base.double()
        e1, e2 = rel(dx, ref), elem(dx, ref)
        ok = e1 < 2.5e-3 and e2 < 1e-4
        print(f"{spec:26s} dgrad +=     rel {e1:.1e} elem {e2:.1e} {'ok' if ok else 'MISS'}", flush=True)
        if not ok:
            bad.append((spec, "dgrad accumulate"))
        # fused reductions
        xhat = (bn_y.double() - mean.double()) * invstd.double()
        cases = [("self mask", None, False), ("byte mask", m8, False)]
        if s == 1 and k == 1:
            cases.append(("residual", m8, True))
        for name, mk, resid in cases:
            sums.zero_()
            dx.fill_(7.0)
            fused(mk, resid)
            torch.cuda.synchronize()
            ref = dx_ref if not resid else dx_ref + torch.where(block_out.double() > 0, res_src.double(), torch.zeros((), dtype=torch.float64))
            e1, e2 = rel(dx, ref), elem(dx, ref)
            ok = e1 < 2.5e-3 and e2 < 1e-4
            msg = ""
            if applied.value:
                # the reduction reads the ROUNDED stored value; reference sums from the kernel's own dx isolate the reduction
                got_dx = dx.double().cpu()
                bnv = xhat * gamma.double() + beta.double()
                mask = mask_bits if mk is not None else (bnv > 0)
                sure = torch.ones_like(mask) if mk is not None else (bnv.abs() > 1e-5)
                dd = torch.where(mask, got_dx, torch.zeros((), dtype=torch.float64))
                s_ref = torch.stack((dd.sum((0, 1, 2)), (dd * xhat).sum((0, 1, 2))))
                slack = torch.stack(((got_dx.abs() * ~sure).sum((0, 1, 2)), (got_dx.abs() * xhat.abs() * ~sure).sum((0, 1, 2))))
                scale = torch.stack((dd.abs().sum((0, 1, 2)), (dd * xhat).abs().sum((0, 1, 2))))
                err = float((((sums.double().cpu() - s_ref).abs() - slack).clamp_min(0) / scale.clamp_min(1e-30)).max())
                ok = ok and err < 2e-6
                msg = f"sums {err:.1e}"
            else:
                ok = ok and float(sums.abs().max()) == 0.0
                msg = "not carried"
            print(f"{spec:26s} dgrad {name:9s} rel {e1:.1e} elem {e2:.1e} {msg} {'ok' if ok else 'MISS'}", flush=True)
            if not ok:
                bad.append((spec, name))
    finally:
        ops.conv_set_compute("fp32")
    return bad


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("check", "time") else "check"
    specs = [a for a in sys.argv[1:] if "," in a] or (CHECK if mode == "check" else TIME)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    bad = []
    for sp in specs:
        bad += run(sp, mode == "check")
    if bad:
        print("MISSES:", bad)
        sys.exit(1)
