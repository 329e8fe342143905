"""Isolated timing of the BN backward kernels over the ResNet-101 (512x910, B=4) activation shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip.lib import lib, ptr, stream
dev = torch.device("cuda", 0)
shapes = [(465920, 64, 1), (116736, 64, 6), (116736, 256, 4), (116736, 128, 1), (29184, 128, 7), (29184, 512, 5),
          (29184, 256, 1), (7296, 256, 45), (7296, 1024, 24), (7296, 512, 1), (1856, 512, 5), (1856, 2048, 4)]
tot_r = tot_a = 0.0
print("M K calls | reduce us GB/s | apply us GB/s")
for M, K, calls in shapes:
    y = torch.randn(M, K, device=dev); d = torch.randn(M, K, device=dev); dy = torch.empty_like(y)
    bnp = torch.rand(4, K, device=dev) + 0.5; gamma = torch.rand(K, device=dev) + 0.5
    sums = torch.empty(2, K, device=dev); dgb = torch.empty(2, K, device=dev)
    def red():
        lib.call("dpft_bn_bwd_reduce_f32", ptr(y), ptr(d), None, ptr(bnp), ptr(bnp), ptr(sums), M, K, stream())
    def app():
        lib.call("dpft_bn_bwd_apply_f32", ptr(y), ptr(d), None, ptr(bnp), ptr(bnp), ptr(gamma), ptr(sums), ptr(dy),
                 ptr(dgb[0]), ptr(dgb[1]), M, K, stream())
    res = []
    for fn, passes in ((red, 2), (app, 3)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append((us, passes * M * K * 4 / us / 1e3))
    tot_r += res[0][0] * calls; tot_a += res[1][0] * calls
    print(f"{M:7d} {K:5d} {calls:3d} | {res[0][0]:7.1f} {res[0][1]:6.0f} | {res[1][0]:7.1f} {res[1][1]:6.0f}")
print(f"camera totals per step: reduce {tot_r/1e3:.2f} ms, apply {tot_a/1e3:.2f} ms")
