import sys, torch
sys.path.insert(0, '.')
from dpft_amd.models.backbones import build_backbone
from oracle import dprt_oracle as O
name, cin = 'ResNet50', 6
g = torch.Generator().manual_seed(1); torch.manual_seed(1)
bb = build_backbone(name, dict(name=name, weights='', in_channels=cin, multi_scale=4, norm_layer='BatchNorm2d'))
sd = {'bb.' + k: (v.detach().double() if v.is_floating_point() else v) for k, v in bb.state_dict().items()}
x = torch.rand(2, 128, 96, cin, generator=g) * 255
bb = bb.cuda().train()
outs = bb(x.cuda())
sd_ref = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
ref = O.backbone(x.double(), sd_ref, 'bb', name, train=True, multi_scale=4)
only = sys.argv[1] if len(sys.argv) > 1 else None
cots = {k: torch.randn(ref[k].shape, generator=g, dtype=torch.float64) for k in ref}
if only:
    for k in cots:
        if k != only: cots[k].zero_()
sum((ref[k] * cots[k]).sum() for k in ref).backward()
sum((outs[k] * cots[k].float().cuda()).sum() for k in outs).backward()
for n, p in bb.named_parameters():
    if not any(t in n for t in ('layer3.4', 'layer3.5', 'layer4.0', 'layer2.3', 'layer3.0')): continue
    gref = sd_ref['bb.' + n].grad
    d = (p.grad.double().cpu() - gref).abs()
    den = float(gref.abs().max()) + 1e-12
    flat = d.flatten()
    print(f"{n:40s} err={float(flat.max())/den:.3e}  nbad(>1e-2)={(flat/den > 1e-2).sum().item()}/{flat.numel()}  gmax={den:.3e}")
