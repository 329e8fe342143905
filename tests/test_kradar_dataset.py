"""The K-Radar folder reader (dpft_amd/data/kradar.py) against the reference's own KRadarDataset on the same files.

A small pre-processed tree is written to a temp dir (random JPEG frames, radar maps in dB, calibration matrices, label
rows incl. boxes outside the field of view); the reference class is imported through oracle/ref_import (its
``torchvision.io.read_image`` -- torchvision is absent -- bound to the same Pillow decode) and every item is compared
tensor for tensor, key order included.  The device-transform mode (raw uint8 frame, unscaled maps) is checked against the
host mode + the arithmetic of the two transforms."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_import

PIL = pytest.importorskip("PIL")


def _write_tree(root, n_seq=2, n_samples=3, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    for s in range(n_seq):
        for i in range(n_samples):
            d = os.path.join(root, "train", f"{s + 1}", f"{i:05d}")
            os.makedirs(d)
            frame = rng.integers(0, 256, size=(72, 128, 3), dtype=np.uint8)
            Image.fromarray(frame).save(os.path.join(d, "mono.jpg"), quality=95)
            np.save(os.path.join(d, "mono_info.npy"), rng.normal(size=(4, 4)))
            np.save(os.path.join(d, "ra.npy"), 60.0 + 180.0 * rng.random(size=(32, 107, 6)))      # dB, partly outside [100, 200]
            np.save(os.path.join(d, "ra_info.npy"), rng.normal(size=(4, 4)))
            np.save(os.path.join(d, "ea.npy"), 60.0 + 180.0 * rng.random(size=(37, 107, 6)))
            np.save(os.path.join(d, "ea_info.npy"), rng.normal(size=(4, 4)))
            n = int(rng.integers(0, 6))
            boxes = np.concatenate([rng.uniform(-20, 90, size=(n, 1)), rng.uniform(-40, 40, size=(n, 1)),
                                    rng.uniform(-4, 6, size=(n, 1)), rng.uniform(-3.1, 3.1, size=(n, 1)),
                                    rng.uniform(0.5, 5, size=(n, 3)), rng.integers(-1, 1, size=(n, 1)).astype(np.float64),
                                    np.arange(n, dtype=np.float64)[:, None]], axis=1)
            np.save(os.path.join(d, "labels.npy"), boxes)
            np.save(os.path.join(d, "description.npy"), np.array([s, i, 0, 1, 2, 3], dtype=np.float64))
    return root


FOV = {"x": [0.0, 72.0], "y": [-6.4, 6.4], "z": [-2.0, 6.0], "azimuth": [-50, 50]}
KW = dict(camera="M", radar="BF", num_classes=2, scale=True, fov=FOV, dtype="float32")


def test_folder_reader_walks_the_tree_and_filters_labels(tmp_path):
    from dpft_amd.data.kradar import KRadarFolderDataset, MAX_POWER, MIN_POWER
    root = _write_tree(str(tmp_path))
    raw = KRadarFolderDataset(root, device_transforms=True, **KW)
    host = KRadarFolderDataset(root, device_transforms=False, **KW)
    assert len(raw) == len(host) == 6
    np.random.seed(0)
    for i in range(len(raw)):
        (a, la), (b, lb) = raw[i], host[i]
        assert list(a.keys()) == list(b.keys()) == [
            "camera_mono", "radar_bev", "radar_front", "label_to_camera_mono_t", "label_to_radar_bev_t", "label_to_radar_front_t",
            "label_to_camera_mono_p", "label_to_radar_bev_p", "label_to_radar_front_p", "camera_mono_shape", "radar_bev_shape",
            "radar_front_shape"]
        assert a["camera_mono"].dtype == torch.uint8 and a["camera_mono"].shape == (72, 128, 3)
        assert torch.equal(a["camera_mono"].float(), b["camera_mono"])
        for k in ("radar_bev", "radar_front"):      # the transform the device kernel applies to the raw map
            assert torch.equal(torch.clip((a[k] - MIN_POWER) / (MAX_POWER - MIN_POWER) * 255.0, 0, 255), b[k])
            assert float(b[k].min()) >= 0.0 and float(b[k].max()) <= 255.0
        assert a["camera_mono_shape"].tolist() == [72, 128, 3] and a["radar_front_shape"].tolist() == [37, 107, 6]
        assert not a["label_to_camera_mono_t"].any() and a["label_to_radar_bev_p"].shape == (3, 4)
        c = la["gt_center"]
        az = torch.rad2deg(torch.atan2(c[:, 1], c[:, 0]))
        assert bool(((c[:, 0] > 0) & (c[:, 0] < 72) & (c[:, 1].abs() < 6.4) & (c[:, 2] > -2) & (c[:, 2] < 6) & (az.abs() < 50)).all())
        assert la["gt_class"].shape == (c.shape[0], 2) and la["gt_angle"].shape == (c.shape[0], 2)
        for k in la:
            assert torch.equal(la[k], lb[k])
    # modality dropout: one draw per sample, never both
    np.random.seed(1)
    drop = KRadarFolderDataset(root, device_transforms=True, camera_dropout=0.5, radar_dropout=0.5, **KW)
    kinds = set()
    for i in range(len(drop)):
        s, _ = drop[i]
        cam0, rad0 = not s["camera_mono"].any(), not (s["radar_bev"].any() or s["radar_front"].any())
        assert cam0 != rad0
        kinds.add(cam0)
    assert kinds == {True, False}


def test_folder_reader_collates_like_the_reference_loader(tmp_path):
    from dpft_amd.data.kradar import KRadarFolderDataset
    from dpft_amd.data.loader import listed_collating
    root = _write_tree(str(tmp_path))
    ds = KRadarFolderDataset(root, device_transforms=True, **KW)
    batch, labels = listed_collating([ds[i] for i in range(4)])
    assert list(batch.keys())[0] == "camera_mono" and batch["camera_mono"].shape == (4, 72, 128, 3)
    assert batch["label_to_radar_front_p"].shape == (4, 3, 4) and len(labels) == 4


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference is not present")
def test_folder_reader_equals_the_reference_dataset(tmp_path):
    """Item for item against src/dprt/datasets/kradar/dataset.py on the same files (no resize: torchvision's is absent)."""
    from dpft_amd.data.kradar import KRadarFolderDataset, read_image_hwc
    ref_import.install()
    import dprt.datasets.kradar.dataset as ref_ds
    ref_ds.read_image = lambda path: read_image_hwc(path).movedim(-1, 0)      # torchvision.io.read_image: (C, H, W) uint8
    root = _write_tree(str(tmp_path), seed=3)
    ref = ref_ds.KRadarDataset(root, **KW)
    ours = KRadarFolderDataset(root, device_transforms=False, **KW)
    assert len(ref) == len(ours) == 6
    fov_t = {k: torch.tensor(v) for k, v in FOV.items()}
    ref.fov = fov_t
    # upstream walks the sequences in os.listdir order (file-system dependent), this repo in sorted order (every rank must see
    # the same index -> sample map): pair the items by their sample folder, and check that both hold the same set
    where = {os.path.dirname(f["description"]): j for j, f in enumerate(ours.samples)}
    assert sorted(where) == sorted(os.path.dirname(f["description"]) for f in ref.dataset_paths)
    for i in range(len(ref)):
        np.random.seed(i)
        item_r, label_r = ref[i]
        np.random.seed(i)
        item_o, label_o = ours[where[os.path.dirname(ref.dataset_paths[i]["description"])]]
        assert list(item_r.keys()) == list(item_o.keys())
        for k in item_r:
            assert item_r[k].dtype == item_o[k].dtype and torch.equal(item_r[k], item_o[k]), k
        assert list(label_r.keys()) == list(label_o.keys())
        for k in label_r:
            assert torch.equal(label_r[k], label_o[k]), k
    # the same with a modality dropped, the same random draw
    ref2 = ref_ds.KRadarDataset(root, camera_dropout=0.4, radar_dropout=0.4, **KW)
    ref2.fov = fov_t
    ours2 = KRadarFolderDataset(root, device_transforms=False, camera_dropout=0.4, radar_dropout=0.4, **KW)
    for i in range(len(ref2)):
        np.random.seed(100 + i)
        item_r, _ = ref2[i]
        np.random.seed(100 + i)
        item_o, _ = ours2[where[os.path.dirname(ref2.dataset_paths[i]["description"])]]
        for k in item_r:
            assert torch.equal(item_r[k], item_o[k]), k
