"""Accuracy (vs fp64) and time of the conv forward / data gradient in the three compute modes."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
torch.manual_seed(0)
for (B, H, W, C, K, k, s_) in [(4, 32, 57, 256, 256, 3, 1), (4, 32, 57, 256, 1024, 1, 1), (4, 32, 57, 1024, 256, 1, 1), (2, 64, 114, 128, 128, 3, 1)]:
    pad = k // 2
    x = torch.randn(B, H, W, C) * 3 + 1
    w = torch.randn(K, C, k, k) / (C * k * k) ** 0.5
    yref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, stride=s_, padding=pad)
    dy = torch.randn(yref.shape, dtype=torch.float64)
    dxref = torch.autograd.grad(F.conv2d(x.double().permute(0, 3, 1, 2).requires_grad_(True), w.double(), None, stride=s_, padding=pad), [], [], allow_unused=True) if False else None
    xa = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    (F.conv2d(xa, w64, None, stride=s_, padding=pad) * dy).sum().backward()
    wgref = w64.grad
    cv = ops.conv_problem(B, H, W, C, K, k, k, s_, pad)
    wd = w.permute(0, 2, 3, 1).contiguous().cuda(); wt = ops.weight_transpose(wd)
    xd = x.cuda(); dyd = dy.permute(0, 2, 3, 1).contiguous().float().cuda()
    line = f"{(B,H,W,C,K,k,s_)}:"
    for mode in ("fp32", "bf16", "bf16x3"):
        ops.conv_set_compute(mode)
        def run():
            y, _ = ops.conv_fwd(cv, xd, wd); dx = ops.conv_dgrad(cv, dyd, wt); dw = ops.conv_wgrad(cv, xd, dyd); return y, dx, dw
        y, dx, dw = run()
        ey = float((y.double().cpu().permute(0, 3, 1, 2) - yref).norm() / yref.norm())
        ed = float((dx.double().cpu().permute(0, 3, 1, 2) - xa.grad).norm() / xa.grad.norm())
        ew = float((dw.double().cpu().permute(0, 3, 1, 2) - wgref).norm() / wgref.norm())
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        line += f"  {mode}: err fwd {ey:.1e} dgrad {ed:.1e} wgrad {ew:.1e}, fwd+dgrad+wgrad {e0.elapsed_time(e1)/20*1e3:.0f} us |"
    ops.conv_set_compute("fp32")
    print(line, flush=True)
