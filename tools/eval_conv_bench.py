"""Inference conv + BatchNorm (+ residual) (+ ReLU) launches of the camera body, each alone: us and TF per shape.
   python tools/eval_conv_bench.py [B]      (DPFT_FORCE_TILE=bm,bn,splits for tile A/Bs)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
# (H, W, C, K, k, stride, residual)
SHAPES = [(128, 228, 64, 64, 1, 1, 0), (128, 228, 64, 64, 3, 1, 0), (128, 228, 64, 256, 1, 1, 1), (128, 228, 256, 64, 1, 1, 0),
          (64, 114, 512, 128, 1, 1, 0), (64, 114, 128, 128, 3, 1, 0), (64, 114, 128, 512, 1, 1, 1),
          (32, 57, 1024, 256, 1, 1, 0), (32, 57, 256, 256, 3, 1, 0), (32, 57, 256, 1024, 1, 1, 1),
          (16, 29, 2048, 512, 1, 1, 0), (16, 29, 512, 512, 3, 1, 0), (16, 29, 512, 2048, 1, 1, 1)]
g = torch.Generator().manual_seed(0)
tot = 0.0
for H, W, C, K, k, s, res in SHAPES:
    cv = ops.conv_problem(B, H, W, C, K, k, k, s, k // 2)
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    w = (torch.randn(K, k, k, C, generator=g) / (C * k * k) ** 0.5).to(dev)
    bn = torch.stack((torch.zeros(K), torch.ones(K), torch.zeros(K), torch.ones(K))).contiguous().to(dev)
    r = torch.randn(B, cv.OH, cv.OW, K, generator=g).to(dev) if res else None
    f = lambda: ops.conv_fwd_bnact(cv, x, w, bn, True, r)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * B * cv.OH * cv.OW * K * k * k * C
    print(f"B={B} {H}x{W} {C}->{K} k{k} res={res}: {us:7.1f} us {fl / us / 1e6:6.1f} TF", flush=True)
