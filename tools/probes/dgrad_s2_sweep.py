"""Stride-2 data gradients at small odd / even map sizes vs fp64 (which path of the dispatch is off?)."""
import os, sys, itertools
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dpft_amd.hip import ops
g = torch.Generator().manual_seed(0)
shapes = [(2, 32, 11, 128, 128, 3, 2), (2, 32, 11, 256, 512, 1, 2), (2, 32, 12, 128, 128, 3, 2), (2, 31, 11, 128, 128, 3, 2),
          (2, 16, 6, 256, 256, 3, 2), (2, 8, 3, 512, 512, 3, 2), (2, 16, 6, 512, 1024, 1, 2), (2, 8, 3, 1024, 2048, 1, 2),
          (2, 24, 40, 128, 128, 3, 2), (2, 24, 40, 256, 512, 1, 2), (2, 12, 20, 256, 256, 3, 2), (2, 6, 10, 512, 512, 3, 2),
          (2, 10, 27, 128, 128, 3, 2), (2, 5, 14, 256, 256, 3, 2), (2, 3, 7, 512, 512, 3, 2), (4, 64, 27, 128, 128, 3, 2),
          (2, 32, 11, 64, 64, 3, 1), (2, 32, 11, 128, 128, 3, 1)]
for B, H, W, C, K, k, s in shapes:
    pad = k // 2
    x = torch.randn(B, H, W, C, generator=g)
    w = (torch.randn(K, C, k, k, generator=g) / (C * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    xa = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yref = F.conv2d(xa, w.double(), None, stride=s, padding=pad)
    dy = torch.randn(yref.shape, generator=g, dtype=torch.float64)
    (yref * dy).sum().backward()
    cv = ops.conv_problem(B, H, W, C, K, k, k, s, pad)
    wg = w.cuda().permute(0, 2, 3, 1)
    dyg = dy.permute(0, 2, 3, 1).contiguous().float().cuda()
    wt = ops.weight_transpose(wg)
    dx = ops.conv_dgrad(cv, dyg, wt).permute(0, 3, 1, 2).double().cpu()
    err = float((dx - xa.grad).norm() / xa.grad.norm())
    d = (dx - xa.grad).abs().amax(dim=(0, 1))                        # worst error per input pixel (H, W)
    bad = (d > 1e-4 * float(xa.grad.abs().max())).nonzero().tolist()
    print(f"B{B} {H}x{W} C{C} K{K} k{k} s{s}: rel-L2 {err:.2e}  bad pixels {len(bad)} {bad[:12]}")
