"""Train-mode ResNet body forward + backward through the launch plan, saved to a file: outputs, parameter gradients, running
statistics.  tests/test_gpu_kernels.py runs it with DPFT_BN_SUMS = 0 / 1 (BatchNorm statistics as per-tile tables + a finalize
launch per layer, or as fixed-point column sums that the consumers read themselves, csrc/common.h: BnSumsRef) and compares the
files; the size is chosen so that every kernel with a sums form takes part (the streaming 1x1 kernel needs >= 16 384 rows).
   python tools/bn_sums_check.py OUT.pt [ResNet50|ResNet101] [B,H,W]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.models.backbones import build_backbone

out_path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "ResNet50"
B, H, W = map(int, (sys.argv[3] if len(sys.argv) > 3 else "2,512,256").split(","))
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
torch.manual_seed(3)
bb = build_backbone(name, dict(name=name, weights="", in_channels=3, multi_scale=4, norm_layer="BatchNorm2d"))
with torch.no_grad():
    for m in bb.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
x = (torch.rand(B, H, W, 3, generator=g) * 255).to(dev)
res = {"x": x.cpu(), "sd": {k: v.detach().cpu().clone() for k, v in bb.state_dict().items()}, "name": name}
bb = bb.to(dev).train()
for rep in range(2):      # twice: the second pass must not see anything of the first (accumulators are cleared per forward)
    bb.zero_grad(set_to_none=True)
    outs = bb(x)
    cots = {k: torch.randn(v.shape, generator=torch.Generator().manual_seed(7 + int(k))).to(dev) for k, v in outs.items()}
    sum((outs[k] * cots[k]).sum() for k in outs).backward()
    torch.cuda.synchronize()
    res[rep] = {"out": {k: v.detach().cpu() for k, v in outs.items()},
                "grad": {n: p.grad.detach().cpu() for n, p in bb.named_parameters()},
                "buf": {n: b.detach().cpu().clone() for n, b in bb.named_buffers()}}
torch.save(res, out_path)
print("saved", out_path, {k: tuple(v.shape) for k, v in res[0]["out"].items()})
