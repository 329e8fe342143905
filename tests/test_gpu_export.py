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


def test_evaluator_end_to_end(tmp_path):
    """dprt.evaluate's loop on a saved checkpoint: metrics + export of every batch, latency protocol, complexity."""
    import copy
    from dpft_amd.configs import load_config
    from dpft_amd.evaluation import build_evaluator
    from dpft_amd.models import build
    from dpft_amd.synthetic import make_batch, make_labels
    from oracle import export_oracle as EO
    from oracle import metric_oracle as MO
    cfg = copy.deepcopy(load_config("kradar"))
    cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"
    shapes = {"camera_mono": (96, 160, 3), "radar_bev": (128, 43, 6), "radar_front": (37, 107, 6)}
    torch.manual_seed(3)
    model = build("dprt", cfg)
    with torch.no_grad():                                               # some confident foreground predictions
        model.fuser.heads[-1].layers["class_head"][-1].weight.mul_(8.0)
    ckpt = tmp_path / "20240101-000000_checkpoint_0007.pt"
    torch.save(model, str(ckpt))
    loader = []
    for i in range(2):
        labels = make_labels(2, seed=20 + i)
        for b, lab in enumerate(labels):
            lab["description"] = torch.tensor([(i + b) % 9, b % 2, (2 * i + b) % 7])
        loader.append((make_batch(cfg["model"]["inputs"], 2, seed=30 + i, shapes=shapes), labels))
    ev = build_evaluator(cfg)
    ev.evaluate(str(ckpt), loader, str(tmp_path / "out"))
    dst = tmp_path / "out" / "20240101-000000"
    scalars = [__import__("json").loads(l) for l in open(dst / "scalars.jsonl")]
    tags = {s["tag"] for s in scalars}
    assert {"test/mAP", "test/mGIoU", "test/Inference_time_mean_ms", "test/Inference_time_std_ms", "test/FLOPS",
            "test/MACS", "test/Parameters"} <= tags
    assert all(s["step"] == 7 for s in scalars)
    by = {s["tag"]: s["value"] for s in scalars}
    assert by["test/FLOPS"] > 1e9 and by["test/MACS"] * 2 == by["test/FLOPS"]
    assert by["test/Parameters"] == sum(p.numel() for p in model.parameters())
    # the exported tree and the epoch metrics equal the oracle's on the model's own outputs
    model = model.to(DEV).eval()
    tree, acc = {}, {"mAP": 0.0, "mGIoU": 0.0}
    for i, (data, labels) in enumerate(loader):
        with torch.no_grad():
            out = {k: v.cpu() for k, v in model(to_dev(data)).items()}
        for path, text in EO.export_tree(out, labels, i * len(labels), categories=cfg["data"].get("categories")).items():
            tree[path] = tree.get(path, "") + text
        for k, v in MO.metric_forward(out, labels).items():
            acc[k] += float(v) / len(loader)
    got = {p: t for p, t in read_tree(str(dst)).items() if p.startswith("exports/")}
    assert sorted(got) == sorted(tree)
    for path, text in tree.items():
        assert got[path] == text, path
    for k in acc:
        assert abs(by[f"test/{k}"] - acc[k]) < 1e-5, (k, by[f"test/{k}"], acc[k])
