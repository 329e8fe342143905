"""The matcher's assignment step on the device (dpft_lsap_batch_dev_f32, dpft_amd/csrc/lsap.hip) against
scipy.optimize.linear_sum_assignment -- the call the reference makes per sample on a .cpu() copy of the cost matrix
(src/dprt/training/loss.py:305 through assigner.py:134-150) -- and against the host restatement of the same algorithm
(dpft_lsap_batch_f32, tests/test_host.py): same pairs, same order, bit for bit (integer work)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _device_assign(cost: np.ndarray, counts, status=None):
    from dpft_amd.hip.lib import lib, stream
    B, N, Mmax = cost.shape
    c = torch.from_numpy(cost).to(DEV)
    cnt = torch.tensor(list(counts), dtype=torch.int32, device=DEV)
    match = torch.full((B, Mmax, 2), 77, dtype=torch.int32, device=DEV)
    nm = torch.full((B,), 77, dtype=torch.int32, device=DEV)
    st = status if status is not None else torch.zeros(1, dtype=torch.int32, device=DEV)
    lib.call("dpft_lsap_batch_dev_f32", c.data_ptr(), cnt.data_ptr(), match.data_ptr(), nm.data_ptr(), st.data_ptr(), B, N, Mmax,
             stream())
    torch.cuda.synchronize()
    return match.cpu().numpy(), nm.cpu().numpy(), int(st[0])


def _host_assign(cost: np.ndarray, counts):
    from dpft_amd.hip.lib import lib
    B, N, Mmax = cost.shape
    match = np.empty((B, Mmax, 2), np.int32)
    nm = np.empty(B, np.int32)
    cnt = np.asarray(counts, np.int32)
    assert lib.dpft_lsap_batch_f32(cost.ctypes.data, B, N, Mmax, cnt.ctypes.data, match.ctypes.data, nm.ctypes.data) == 0
    return match, nm


def _check_against_scipy(cost, counts, match, nm, tag):
    from scipy.optimize import linear_sum_assignment
    for b, m in enumerate(counts):
        m = int(m)
        if m == 0:
            assert nm[b] == 0 and (match[b] == -1).all(), tag
            continue
        i, j = linear_sum_assignment(cost[b, :, :m])
        k = len(i)
        assert nm[b] == k and (match[b, :k, 0] == i).all() and (match[b, :k, 1] == j).all() and (match[b, k:] == -1).all(), (tag, b)


def test_device_assignments_are_scipys_pairs_in_scipys_order():
    """600 random rectangular problems in both orientations (more queries than targets and the reverse), continuous costs,
    half-integer costs (many ties), three-valued costs (almost only ties), empty samples, -1 padding."""
    rng = np.random.default_rng(0)
    for trial in range(600):
        B, N, Mmax = int(rng.integers(1, 6)), int(rng.integers(1, 90)), int(rng.integers(1, 14))
        if trial % 50 == 7:
            N, Mmax = int(rng.integers(1, 12)), int(rng.integers(12, 80))          # wide: the queries are the rows
        cost = rng.standard_normal((B, N, Mmax)).astype(np.float32)
        if trial % 4 == 1:
            cost = np.round(cost * 2) / 2
        elif trial % 4 == 2:
            cost = np.abs(cost) * 1e3
        elif trial % 4 == 3:
            cost[:] = rng.integers(0, 3, cost.shape)
        counts = rng.integers(0, Mmax + 1, B)
        match, nm, st = _device_assign(cost, counts)
        assert st == 0
        _check_against_scipy(cost, counts, match, nm, trial)
        hm, hn = _host_assign(cost, counts)
        assert (hm == match).all() and (hn == nm).all(), trial


def test_device_assignments_at_the_training_shape_and_beyond():
    """kradar.json's shape (400 queries, up to 30 targets a frame, batch 4 and 8), a matrix that needs more than 64 KiB of
    LDS, one too large to stage in LDS at all, and a square one."""
    rng = np.random.default_rng(1)
    for B, N, Mmax in ((4, 400, 7), (8, 400, 30), (2, 1200, 20), (2, 1600, 40), (1, 64, 64)):
        cost = rng.standard_normal((B, N, Mmax)).astype(np.float32)
        counts = [Mmax] + [int(c) for c in rng.integers(0, Mmax + 1, B - 1)]
        match, nm, st = _device_assign(cost, counts)
        assert st == 0
        _check_against_scipy(cost, counts, match, nm, (B, N, Mmax))
    # ties at the training shape: costs on a coarse grid
    cost = (rng.integers(0, 4, (4, 400, 12)) * 0.25).astype(np.float32)
    counts = [12, 5, 0, 1]
    match, nm, st = _device_assign(cost, counts)
    _check_against_scipy(cost, counts, match, nm, "ties")


def test_non_finite_costs_are_reported_through_the_status_word():
    """scipy raises ValueError for NaN / inf entries; the kernel leaves 1 + b in the status word (device memory or page-locked
    host memory) and gives the sample no pairs; entries beyond the sample's count are not looked at."""
    rng = np.random.default_rng(2)
    cost = rng.standard_normal((4, 400, 7)).astype(np.float32)
    counts = [7, 3, 0, 1]
    cost[3, 7, 6] = np.inf                                   # beyond counts[3]
    match, nm, st = _device_assign(cost, counts)
    assert st == 0
    _check_against_scipy(cost, counts, match, nm, "inf beyond the count")
    cost[1, 5, 2] = np.nan
    match, nm, st = _device_assign(cost, counts)
    assert st == 2 and nm[1] == 0 and (match[1] == -1).all()
    good = cost.copy()
    good[1] = 0
    _check_against_scipy(good, [7, 0, 0, 1], match, nm, "the other samples")
    cost[1, 5, 2] = 0.0
    match, nm, st = _device_assign(cost, [7, 3, 9, 1])       # more targets than the packed width: refused, no pairs
    assert st == 0x20000 + 2 and nm[2] == 0 and (match[2] == -1).all() and nm[0] == 7
    pinned = torch.zeros(1, dtype=torch.int32).pin_memory()
    cost[1, 5, 2] = -np.inf
    _device_assign(cost, counts, status=pinned)
    assert int(pinned[0]) == 2


def test_loss_with_device_assignments_equals_the_host_path_and_defers_the_error():
    """Loss.forward_fused with assign_on_device: loss terms and gradients bit-equal to the read-back path's; a NaN in the
    outputs (-> in the cost matrix) raises scipy's ValueError at check_assignment_status instead of at the call."""
    from dpft_amd.configs import load_config
    from dpft_amd.synthetic import make_labels
    from dpft_amd.training.loss import build_loss
    cfg = load_config("kradar")
    torch.manual_seed(3)
    B, N, ncls = 4, 400, 4
    labels = make_labels(B, seed=5, device=DEV)
    ncls = labels[0]["gt_class"].shape[-1]

    def outputs():
        g = torch.Generator(device="cpu").manual_seed(11)
        mk = lambda *s: torch.randn(*s, generator=g).to(DEV).requires_grad_(True)
        return {"class": mk(B, N, ncls), "center": mk(B, N, 3), "size": mk(B, N, 3), "angle": mk(B, N, 2)}

    res = []
    for on_dev in (False, True):
        fn = build_loss(cfg["train"])
        fn.assign_on_device = on_dev
        out = outputs()
        total, losses = fn(out, labels)
        total.backward()
        fn.check_assignment_status()
        res.append((total.detach().clone(), {k: v.detach().clone() for k, v in losses.items()},
                    {k: v.grad.clone() for k, v in out.items()}))
    (t0, l0, g0), (t1, l1, g1) = res
    assert torch.equal(t0, t1) and t0 > 0
    for k in l0:
        assert torch.equal(l0[k], l1[k]), k
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    fn = build_loss(cfg["train"])
    fn.assign_on_device = True
    out = outputs()
    with torch.no_grad():
        out["center"][2, 17, 1] = float("nan")
    fn(out, labels)                                           # no error here: nothing is read back
    with pytest.raises(ValueError, match="sample 2"):
        fn.check_assignment_status()
    fn.check_assignment_status()                              # reported once
