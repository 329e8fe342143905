"""K-Radar exporter and evaluation loop on the GPU: the HIP selection kernel through the
C-ABI vs the oracle (bit-exact masks) and vs the file trees the reference's own exporter wrote (tests/golden)."""
import os

import numpy as np
import pytest
import torch

from test_oracle_golden import export_cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


def read_tree(root):
    tree = {}
    for d, _, files in os.walk(root):
        for f in files:
            full = os.path.join(d, f)
            tree[os.path.relpath(full, root).replace(os.sep, "/")] = open(full).read()
    return tree


def to_dev(d):
    return {k: v.to(DEV) for k, v in d.items()}


def test_exporter_writes_the_reference_file_tree(golden, tmp_path):
    from dpft_amd.evaluation.exporters.kradar import KRadarExporter
    for ci, (case, calls) in enumerate(export_cases(golden)):
        dst = tmp_path / f"c{ci}"
        exporter = KRadarExporter(categories=case["categories"])
        for out, tgts, step in calls:
            exporter(to_dev(out), [to_dev(t) for t in tgts], step, str(dst))
        tree = read_tree(str(dst))
        assert sorted(tree) == sorted(case["tree"])
        for path, text in case["tree"].items():
            assert tree[path] == text, path


@pytest.mark.parametrize("B,N,ncls,seed", [(1, 1, 2, 0), (3, 400, 2, 1), (2, 1000, 8, 2), (4, 257, 3, 3)])
def test_selection_mask_and_rows_bit_exact(B, N, ncls, seed):
    """mask bits == cls & conf & fov of the oracle for every threshold, survivors in candidate order with the
    oracle's columns (yaw: device atan2f vs torch CPU atan2, 1e-6)."""
    from dpft_amd.evaluation.exporters.kradar import KRadarExporter
    from oracle import export_oracle as EO
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn(B, N, ncls, generator=g) * 0.7
    cls[:, ::7] = (cls[:, ::7] * 10).round() / 10                       # values that hit thresholds exactly
    center = torch.stack((-5 + torch.rand(B, N, generator=g) * 85, -8 + torch.rand(B, N, generator=g) * 16,
                          -3 + torch.rand(B, N, generator=g) * 10), -1)
    center[:, ::11, 0] = 72.0
    center[:, 1::11, 1] = -6.4
    size = 1 + torch.rand(B, N, 3, generator=g) * 4
    yaw = (torch.rand(B, N, generator=g) * 2 - 1) * 3.14
    angle = torch.stack((torch.sin(yaw), torch.cos(yaw)), -1)
    thrs = [0.0, 0.3, 0.5, 0.7, 0.9, 1.5, -1.0, 0.1]
    rows, counts, mask = KRadarExporter.select(cls.to(DEV), center.to(DEV), size.to(DEV), angle.to(DEV), thrs)
    rows, counts, mask = rows.cpu().numpy(), counts.cpu().numpy(), mask.cpu().numpy()
    for b in range(B):
        for t, thr in enumerate(thrs):
            ref_mask, _, _ = EO.selection_mask(cls[b], center[b], angle[b], thr)
            assert np.array_equal((mask[b] >> t) & 1, ref_mask.numpy().astype(np.uint8)), (b, thr)
            ref = EO.construct_objects({"class": cls[b], "center": center[b], "size": size[b], "angle": angle[b]}, thr)
            assert counts[b, t] == ref.shape[0]
            got = rows[b, t, :counts[b, t]].astype(np.float64)
            assert np.array_equal(got[:, :7], ref[:, [0, 8, 9, 10, 11, 12, 13]])
            np.testing.assert_allclose(got[:, 7], ref[:, 14], rtol=0, atol=1e-6)


def test_exporter_argument_errors(tmp_path):
    from dpft_amd.evaluation.exporters.kradar import KRadarExporter
    from dpft_amd.hip.lib import HipLibraryError
    with pytest.raises(ValueError):
        KRadarExporter(categories={"Sedan": 0})
    with pytest.raises(TypeError):
        KRadarExporter(time_zone=[("day", 0), ("night", 1)])
    z = torch.zeros(1, 4, 2)
    with pytest.raises(HipLibraryError):                                 # no CPU path
        KRadarExporter.select(z, torch.zeros(1, 4, 3), torch.zeros(1, 4, 3), z, [0.5])
    with pytest.raises(ValueError):
        KRadarExporter.select(z.to(DEV), torch.zeros(1, 4, 3).to(DEV), torch.zeros(1, 4, 3).to(DEV), z.to(DEV), [0.1] * 9)
