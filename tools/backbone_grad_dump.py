"""One training forward + backward of a ResNet encoder body through its launch plan, gradients written to a file.
The target of A/B tests over process-wide switches (DPFT_BN_FUSE, DPFT_ACT16 ...): run twice, compare the files.
  python tools/backbone_grad_dump.py OUT.pt [resnet50] [B,H,W] [fp32|bf16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.hip import ops
from dpft_amd.models.backbones.resnet import Backbone, BackboneBase

out = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
B, H, W = map(int, (sys.argv[3] if len(sys.argv) > 3 else "2,96,160").split(","))
mode = sys.argv[4] if len(sys.argv) > 4 else "fp32"
dev = torch.device("cuda", 0)
BackboneBase.ACT16_MIN_PIXELS = 0          # bf16 storage for this (small) map too
ops.conv_set_compute(mode)
torch.manual_seed(5)
m = Backbone(name, in_channels=3, multi_scale=4).to(dev).train()
if os.environ.get("TAME", "0") == "1":
    # At random init the body is chaotic in its input (a 2^-9 perturbation grows ~5x per stage: bf16 and fp32 runs then share
    # nothing).  Small residual-branch gains (torchvision's zero_init_residual, softened) make the blocks near-identity maps:
    # perturbations stay at their size and a comparison ACROSS arithmetic modes means something.
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if n_.endswith("bn3.weight"):
                p_.fill_(0.25)
g = torch.Generator().manual_seed(6)
x = torch.randn(B, H, W, 3, generator=g).to(dev)
feats = m(x)
douts = {k: torch.randn(v.shape, generator=g).to(dev) for k, v in feats.items()}
if os.environ.get("TAME", "0") == "1":
    # gradient = the output itself: zero where a ReLU mask may flip between arithmetic modes (a random gradient there turns a
    # flipped fraction f of the mask into a relative error of sqrt(f))
    loss = sum(0.5 * (feats[k].float() ** 2).sum() for k in feats)
else:
    loss = sum((feats[k] * douts[k]).sum() for k in feats)
loss.backward()
torch.cuda.synchronize()
res = {"loss": float(loss), "feats": {k: v.detach().float().cpu() for k, v in feats.items()},
       "grads": {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None}}
torch.save(res, out)
print("loss", float(loss), "params with grad", len(res["grads"]))
