import torch, sys
sys.path.insert(0, "/root/repo")
from dpft_amd.hip import ops
from dpft_amd.hip.lib import lib, stream
torch.manual_seed(0)
dev = torch.device("cuda")
# memops
a = torch.randn(1000003 * 4 // 4, device=dev); b = torch.empty_like(a); c = torch.randn(77, device=dev); d = torch.ones(77, device=dev)
e = torch.randn(13, device=dev)[1:]      # unaligned (4-byte) source
f = torch.empty(12, device=dev)
z = torch.ones(5000, device=dev)
i64 = torch.arange(6, device=dev).view(2, 3); j64 = torch.empty_like(i64)
ops.memops([(b, a), (d, c), (f, e), (z, None), (j64, i64)])
torch.cuda.synchronize()
assert torch.equal(a, b) and torch.equal(c, d) and torch.equal(f, e) and float(z.abs().sum()) == 0 and torch.equal(i64, j64)
pairs = [(torch.empty(100 + i, device=dev), torch.randn(100 + i, device=dev)) for i in range(40)]
ops.memops(pairs); torch.cuda.synchronize()
assert all(torch.equal(x, y) for x, y in pairs)
# sum_leading
x = torch.randn(3, 4, 400, 16, device=dev); y = torch.randn(12, 400, 16, device=dev)
o = ops.sum_leading([x], (4, 400, 16)); assert torch.allclose(o, x.sum(0), atol=1e-5)
o = ops.sum_leading([x, y], (400, 16)); ref = x.sum((0, 1)) + y.sum(0); assert torch.allclose(o, ref, atol=1e-4), (o - ref).abs().max()
g = torch.randn(4, 5, 14, 16, device=dev); r = torch.randn(32, 4, 5, 14, 16, device=dev); ref = g + r.sum(0)
ops.sum_leading([r], g.shape, out=g, accumulate=True); assert torch.allclose(g, ref, atol=1e-4)
o1 = ops.sum_leading([x, y], (400, 16)); assert torch.equal(o, o1)
# seed
st = torch.tensor([12345], dtype=torch.int64, device=dev); sn = torch.empty_like(st)
lib.call("dpft_seed_advance", st.data_ptr(), sn.data_ptr(), 7, stream()); torch.cuda.synchronize()
assert int(sn) == 12345 and int(st) == 12352
print("new ops OK")
