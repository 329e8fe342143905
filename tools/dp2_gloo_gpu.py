"""Two data-parallel ranks sharing ONE GPU over gloo (RCCL refuses two ranks on a device): exercises the real GPU
training path -- side streams, decoder graphs, direct-to-bucket gradients, bucket all-reduce -- under world_size 2.
Checks: parameters stay identical across ranks; the averaged gradients equal the mean of the two ranks' single-process
gradients.   torchrun --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dp2_gloo_gpu.py"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpft_amd.configs import load_config
from dpft_amd.models import build
from dpft_amd.synthetic import make_batch, make_labels
from dpft_amd.training.trainer import DataParallelTrainer

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cfg = load_config("kradar")
cfg["model"]["fuser"]["dropout"] = 0.0
torch.manual_seed(100 + rank)                      # different initial weights: broadcast must fix that
tr = DataParallelTrainer(build("dprt", cfg), cfg, dev)
shapes = {"camera_mono": (128, 224, 3), "radar_bev": (128, 43, 6), "radar_front": (37, 107, 6)}
data = make_batch(cfg["model"]["inputs"], 2, seed=7 + rank, shapes=shapes, device=dev)
labels = make_labels(2, seed=3 + rank, device=dev)
if os.environ.get("GRAPHS", "1") == "1":
    tr.enable_graphs(data)


def checksum():
    return torch.stack([p.detach().double().sum() for p in tr.model.parameters()]).cpu()


def same_on_all_ranks(t, what):
    ts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    err = max(float((x - ts[0]).abs().max()) for x in ts)
    assert err == 0.0, (what, err)


same_on_all_ranks(checksum(), "initial parameters")
for step in range(3):
    loss, _ = tr.train_step(data, labels)
    torch.cuda.synchronize()
    same_on_all_ranks(checksum(), f"parameters after step {step}")
# the step decision without a read-back (trainer.sync_free_decision): a rank whose shard has no target runs its backward over a
# zero-valued loss; when NO rank has a target the optimizer's device-side gate keeps every parameter as it was
empty = [{k: v[:0] for k, v in l.items()} for l in labels]
before = checksum()
tr.train_step(data, empty if rank == 1 else labels)
torch.cuda.synchronize()
after = checksum()
same_on_all_ranks(after, "parameters after the step with an empty shard on rank 1")
assert not torch.equal(before, after), "the step with one empty shard must still update"
tr.train_step(data, empty)
torch.cuda.synchronize()
assert torch.equal(checksum(), after), "no rank has a target: nothing may change"
same_on_all_ranks(checksum(), "parameters after the all-empty step")
tr.train_step(data, labels)
torch.cuda.synchronize()
same_on_all_ranks(checksum(), "parameters after the step behind the all-empty one")
assert not torch.equal(checksum(), after)
tr._check_matcher()
# reduced gradients are identical on every rank
g = torch.stack([b["flat"].double().sum() for b in tr.reducer.buckets]).cpu()
same_on_all_ranks(g, "reduced gradient buckets")
if rank == 0:
    print("dp2 gloo-on-GPU OK: loss", float(loss))
dist.destroy_process_group()
