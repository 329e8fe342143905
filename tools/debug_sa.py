import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_kernels import _sa_layers
from dpft_amd.models.fusers import train_fused as tf
dev = torch.device("cuda", 0)
B, Q, V = 4, 400, 3
layers = _sa_layers(V, 0.0, dev)
torch.manual_seed(1)
x = (torch.randn(B, Q, 16, device=dev) * 0.7).requires_grad_(True)
pos = (torch.randn(Q, 16, device=dev) * 0.5).requires_grad_(True)
ref = torch.stack([ml.forward_self_attn(x, pos.unsqueeze(0).expand(B, -1, -1)) for ml in layers])
seed = torch.zeros(1, dtype=torch.int64, device=dev)
out = tf.self_attn_blocks(layers, x, pos, seed, 3, 0.0)
err = (out - ref).abs().amax(-1)   # V,B,Q
print(err.amax(-1))
bad = (err > 1e-3).nonzero()
print(bad[:20], len(bad))
