"""3x3 convs of the camera body on the split kernels: forward (no prologue / BatchNorm + ReLU prologue + statistics) and data gradient,
each alone.  DPFT_X3W=0 / 1 in separate processes.   python tools/x3w_bench.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def timed(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for H, W, C in ((128, 228, 64), (64, 114, 128), (32, 57, 256), (16, 29, 512)):
    cv = ops.conv_problem(B, H, W, C, C, 3, 3, 1, 1)
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    w = (torch.randn(C, 3, 3, C, generator=g) / (C * 9) ** 0.5).to(dev)
    wt = ops.weight_transpose(w)
    dy = torch.randn(B, H, W, C, generator=g).to(dev)
    blk = torch.stack((torch.zeros(C), torch.ones(C), torch.zeros(C), torch.ones(C))).contiguous().to(dev)
    y = torch.empty(B, H, W, C, device=dev)
    dx = torch.empty(B, H, W, C, device=dev)
    fl = 2.0 * B * H * W * C * C * 9
    t0 = timed(lambda: ops.conv_fwd(cv, x, w, out=y))
    t1 = timed(lambda: ops.conv_fwd(cv, x, w, pro=(blk, True), want_stats=True, out=y))
    t2 = timed(lambda: ops.conv_dgrad(cv, dy, wt, out=dx))
    print(f"B={B} {H}x{W} {C}->{C} 3x3: fwd {t0:6.1f} us {fl / t0 / 1e6:6.1f} TF | fwd+prologue+stats {t1:6.1f} us {fl / t1 / 1e6:6.1f} TF | dgrad {t2:6.1f} us {fl / t2 / 1e6:6.1f} TF", flush=True)
