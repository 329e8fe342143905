"""Per-parameter gradient error (HIP vs fp64 oracle, CPU fp32 as yardstick) of the small frozen-BatchNorm case of
tests/test_gpu_model.py::test_frozen_batchnorm_backward_matches_oracle, in network order: where does an error start?"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_model as T
from dpft_amd.synthetic import make_batch
from oracle import dprt_oracle as O

train = os.environ.get("TRAIN", "0") == "1"
cfg = T.small_config(dropout=0.0)
g = torch.Generator().manual_seed(int(os.environ.get("SEED", "12")))
model = T._build(cfg, g)
sd64 = T.state_dict_f64(model)
def leafs(dtype):
    return {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
                else (v.to(dtype) if v.is_floating_point() else v)) for k, v in sd64.items()}
sd_ref, sd32 = leafs(torch.float64), leafs(torch.float32)
batch = make_batch(cfg["model"]["inputs"], 2, seed=7, shapes=T.SHAPES)
b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
ref = O.dprt_forward(sd_ref, cfg, b64, train=train)
ref32 = O.dprt_forward(sd32, cfg, batch, train=train)
model = model.to("cuda").train(train)
out = model({k: v.to("cuda") for k, v in batch.items()})
cots = {k: torch.randn(ref[k].shape, generator=g, dtype=torch.float64) for k in ref}
sum((ref[k] * cots[k]).sum() for k in ref).backward()
sum((ref32[k] * cots[k].float()).sum() for k in ref32).backward()
sum((out[k] * cots[k].float().to("cuda")).sum() for k in out).backward()
want = os.environ.get("VIEW", "radar_bev")
for n, p in model.named_parameters():
    if sd_ref[n].grad is None or want not in n:
        continue
    e, e32 = T.rel_l2(p.grad, sd_ref[n].grad), T.rel_l2(sd32[n].grad, sd_ref[n].grad)
    print(f"{n:60s} hip {e:.2e} fp32 {e32:.2e} |g| {float(sd_ref[n].grad.norm()):.2e}")
