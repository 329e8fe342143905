"""Stem pool backward (bn_relu_maxpool_bwd): time on the camera's stem output, tiled vs gather form (DPFT_POOL_BWD_TILED=0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dpft_amd.hip import ops
B = int(os.environ.get("BATCH", "4"))
torch.manual_seed(0)
y = torch.randn(B, 256, 455, 64, device="cuda")
bnp = torch.stack((torch.zeros(64), torch.rand(64) + 0.5, torch.randn(64) * 0.5, torch.ones(64))).cuda()
dout = torch.randn(B, 128, 228, 64, device="cuda")
dz = ops.bn_relu_maxpool_bwd(y, bnp, dout)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    dz = ops.bn_relu_maxpool_bwd(y, bnp, dout)
e1.record(); torch.cuda.synchronize()
print(f"pool bwd B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us  checksum {float(dz.double().sum()):.6f} {float(dz.double().abs().sum()):.6f}")
