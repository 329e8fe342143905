"""Oracle vs golden vectors produced by the imported reference (oracle/gen_golden.py)."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import dprt_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUSER_CFG = dict(i_iter=4, m_views=3, n_heads=[8, 8, 8], n_points=[4, 4, 4], activation="Mish")
VIEWS = ("camera_mono", "radar_bev", "radar_front")


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-4, atol_scale=1e-5):
    b = T(b) if not isinstance(b, torch.Tensor) else b
    atol = atol_scale * max(float(b.abs().max()), 1e-6)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def test_embedding(golden):
    g = golden("embedding.npz")
    for k in "01234":
        out = O.sinusoidal_embedding(T(g["in" + k]), num_feats=16, normalize=True)
        close(out, g["out" + k], rtol=1e-6, atol_scale=1e-6)


def test_querent(golden):
    g = golden("querent.npz")
    q = O.querent(2, [20, 20, 1], [4, -50, 0], [72, 50, 0])
    close(q, g["center"], rtol=1e-6, atol_scale=1e-7)
    assert abs(float(q[0, 0, 0]) - 2.5712) < 1e-3 and abs(float(q[0, 0, 1]) + 3.0642) < 1e-3


def test_reference_points(golden):
    g = golden("refpoints.npz")
    for v in range(3):
        ref = O.reference_points(T(g["centers"]), T(g[f"t{v}"]), T(g[f"p{v}"]), T(g[f"shape{v}"]))
        close(ref, g[f"ref{v}"], rtol=1e-5, atol_scale=1e-6)


def _fuser_inputs(g):
    sd = {k[3:]: T(v) for k, v in g.items() if k.startswith("sd/")}
    sd = {"fuser." + k: v for k, v in sd.items()}
    views = [[T(g[f"view/{n}/{l}"]) for l in range(5)] for n in VIEWS]
    proj = [(T(g[f"t{v}"]), T(g[f"p{v}"])) for v in range(3)]
    shp = [T(g[f"shape{v}"]) for v in range(3)]
    return sd, views, proj, shp


def test_fuser_forward(golden):
    g = golden("fuser_small.npz")
    sd, views, proj, shp = _fuser_inputs(g)
    c0 = T(g["center0"])
    B = c0.shape[0]
    query = sd["fuser.query"].unsqueeze(0).repeat(B, 1, 1)
    qpos = sd["fuser.query_embedding.weight"].unsqueeze(0).repeat(B, 1, 1)
    refs = [O.reference_points(c0, t, p, s) for (t, p), s in zip(proj, shp)]
    ml = O.mlfusion(query, views[0], refs[0], qpos, sd,
                    "fuser.mpfusion.fusion0.ml_fusion_layers.ms_deform_attn0", 8, 4, "Mish")
    close(ml, g["ml00_out"])
    mp = O.mpfusion(query, views, refs, qpos, sd, "fuser.mpfusion.fusion0", [8] * 3, [4] * 3, "Mish")
    close(mp, g["mp0_out"])
    out = O.impfusion(views, shp, proj, c0, sd, "fuser", FUSER_CFG)
    assert list(out.keys()) == ["center", "size", "angle", "class"]
    for k in out:
        close(out[k], g[f"out/{k}"])
    # bit-exact index outputs
    assert torch.equal(out["class"].argmax(-1), T(g["out/class"]).argmax(-1))


def test_fuser_grads(golden):
    g = golden("fuser_small.npz")
    gg = golden("fuser_grads.npz")
    sd, views, proj, shp = _fuser_inputs(g)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    views = [[l.clone().requires_grad_(True) for l in lv] for lv in views]
    out = O.impfusion(views, shp, proj, T(g["center0"]), sd, "fuser", FUSER_CFG)
    loss = sum((out[k] * T(gg[f"cot/{k}"])).sum() for k in out)
    close(loss.detach(), gg["loss"], rtol=1e-4)
    loss.backward()
    for k, v in gg.items():
        if k.startswith("grad/"):
            close(sd["fuser." + k[5:]].grad, v, rtol=2e-3, atol_scale=2e-4)
    for vi, n in enumerate(VIEWS):
        for l in range(5):
            close(views[vi][l].grad, gg[f"gview/{n}/{l}"], rtol=2e-3, atol_scale=2e-4)


def test_head(golden):
    g = golden("head.npz")
    sd = {"h." + k[3:]: T(v) for k, v in g.items() if k.startswith("sd/")}
    out = O.detection_head(T(g["x"]), T(g["ref"]), sd, "h")
    for k in out:
        close(out[k], g[f"out/{k}"], rtol=1e-5)


def test_loss_pieces(golden):
    g = golden("loss.npz")
    close(O.focal_loss(T(g["focal_in"]), T(g["focal_tgt"])), g["focal_out"], rtol=1e-6)
    pred = {k: T(g[f"pred/{k}"])[0] for k in ("class", "center", "size", "angle")}
    tgt = {k: T(g[f"tgt/{k}"])[0] for k in ("gt_class", "gt_center", "gt_size", "gt_angle")}
    losses = O.set_criterion(pred, tgt, T(g["i"])[0], T(g["j"])[0])
    for k, v in losses.items():
        close(v, g[f"loss/{k}"], rtol=1e-5)
    corners = O.box_corners(tgt["gt_center"], tgt["gt_size"], T(g["yaw"])[0])
    close(corners, T(g["corners"])[0], rtol=1e-5)


def test_msda_core_vs_scalar_restatement():
    """grid_sample core == scalar restatement of the upstream kernel body, fp64."""
    gen = torch.Generator().manual_seed(0)
    shapes = [(5, 7), (3, 4), (1, 2)]
    lsi = [0, 35, 47]
    N, M, D, Lq, L, P = 2, 2, 2, 3, 3, 2
    value = torch.randn(N, 49, M, D, generator=gen, dtype=torch.float64)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=gen, dtype=torch.float64) * 1.4 - 0.2
    attn = torch.rand(N, Lq, M, L, P, generator=gen, dtype=torch.float64)
    a = O.msda_core(value, shapes, loc, attn)
    b = O.msda_core_scalar(value, shapes, lsi, loc, attn)
    torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)


def _hidden_stubs():
    """transformers probes torchvision / deepspeed with find_spec and chokes on the reference-import stubs another test
    may have installed; hide them while it is imported and used (order-independent)."""
    from oracle import ref_import
    return ref_import.stubs_hidden()


def test_msda_core_vs_transformers():
    """Independent second opinion (this container only): transformers' pure-PyTorch MSDA."""
    with _hidden_stubs():
        try:
            from transformers.models.deformable_detr.modeling_deformable_detr import (
                MultiScaleDeformableAttention as HFMSDA)
        except ImportError:
            pytest.skip("transformers MSDA not importable")
        gen = torch.Generator().manual_seed(1)
        shapes = [(6, 9), (3, 5)]
        N, M, D, Lq, L, P = 2, 4, 2, 5, 2, 3
        S = sum(h * w for h, w in shapes)
        value = torch.randn(N, S, M, D, generator=gen)
        loc = torch.rand(N, Lq, M, L, P, 2, generator=gen)
        attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=gen), -1).view(N, Lq, M, L, P)
        try:
            ref = HFMSDA().forward(value, torch.tensor(shapes), shapes, torch.tensor([0, 54]), loc, attn, 64)
        except TypeError as e:  # API drift
            pytest.skip(f"transformers MSDA signature differs: {e}")
    torch.testing.assert_close(O.msda_core(value, shapes, loc, attn), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,train", [("resnet50", False), ("resnet50", True), ("resnet101", False)])
def test_resnet_body_vs_transformers(name, train):
    """Independent second opinion for the torchvision ResNet restatement (third party, parity unpinned by the reference;
    SURVEY 4 / VERDICT r1 weak #3): transformers' ResNetModel (bottleneck, v1.5 = stride on the 3x3 conv,
    ``downsample_in_bottleneck=False``) with the SAME weights mapped name by name; all four stage outputs, eval-mode
    running statistics and train-mode batch statistics."""
    with _hidden_stubs():
        try:
            from transformers import ResNetConfig, ResNetModel
        except ImportError:
            pytest.skip("transformers ResNet not importable")
        depths = O.RESNET_DEPTHS[name]
        hf = ResNetModel(ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048],
                                      depths=list(depths), layer_type="bottleneck", hidden_act="relu",
                                      downsample_in_bottleneck=False))
        g = torch.Generator().manual_seed(11)
        hsd = hf.state_dict()
        for k, v in hsd.items():                                   # non-trivial affine + running statistics
            if k.endswith("running_var") or k.endswith("normalization.weight"):
                v.copy_(torch.rand(v.shape, generator=g) * 0.5 + 0.75)
            elif k.endswith("running_mean") or k.endswith("normalization.bias"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
        sd = {}

        def bn(dst, src):
            for f in ("weight", "bias", "running_mean", "running_var"):
                sd[f"{dst}.{f}"] = hsd[f"{src}.normalization.{f}"]

        sd["body.conv1.weight"] = hsd["embedder.embedder.convolution.weight"]
        bn("body.bn1", "embedder.embedder")
        for s, nb in enumerate(depths):
            for b in range(nb):
                src, dst = f"encoder.stages.{s}.layers.{b}", f"body.layer{s + 1}.{b}"
                for c in range(3):
                    sd[f"{dst}.conv{c + 1}.weight"] = hsd[f"{src}.layer.{c}.convolution.weight"]
                    bn(f"{dst}.bn{c + 1}", f"{src}.layer.{c}")
                if f"{src}.shortcut.convolution.weight" in hsd:
                    sd[f"{dst}.downsample.0.weight"] = hsd[f"{src}.shortcut.convolution.weight"]
                    bn(f"{dst}.downsample.1", f"{src}.shortcut")
        assert sum(v.numel() for k, v in sd.items()) == sum(v.numel() for k, v in hsd.items() if "num_batches" not in k)
        x = torch.rand(2, 3, 64, 96, generator=g) * 255.0          # raw 0..255 inputs like the dataset's
        hf.train(train)
        with torch.no_grad():
            ref = hf(x, output_hidden_states=True).hidden_states[1:]   # [0] = embedder output (after the max-pool)
            ours = O.resnet_body(x, sd, "body", depths, train=train)
    assert len(ref) == 4
    for k, r in zip("1234", ref):
        assert ours[k].shape == r.shape
        torch.testing.assert_close(ours[k], r, rtol=1e-4, atol=1e-4 * float(r.abs().max()))


def test_giou_yaw_basic():
    c = torch.tensor([[0.0, 0, 0], [10, 0, 0], [0.5, 0, 0]])
    s = torch.tensor([[2.0, 2, 2]] * 3)
    a = torch.zeros(3)
    g = O.giou3d_yaw(c, s, a, c, s, a)
    assert abs(float(g[0, 0]) - 1.0) < 1e-9
    assert float(g[0, 1]) == -1.0                       # disjoint => reference quirk: exactly -1
    inter = 1.5 * 2 * 2; iou = inter / (16 - inter); evol = 2.5 * 2 * 2; uni = inter / iou
    assert abs(float(g[0, 2]) - (iou - (evol - uni) / evol)) < 1e-9
    # rotation by 90 degrees of a cube changes nothing
    g2 = O.giou3d_yaw(c[:1], s[:1], torch.tensor([1.5707963267948966]), c[:1], s[:1], a[:1])
    assert abs(float(g2[0, 0]) - 1.0) < 1e-6
    # degenerate (zero size) prediction => -1 like the reference's validity mask
    g3 = O.giou3d_yaw(c[:1], torch.tensor([[0.0, 2, 2]]), a[:1], c[:1], s[:1], a[:1])
    assert float(g3[0, 0]) == -1.0


def test_metric_oracle_matches_reference_golden(golden):
    """oracle/metric_oracle.py vs the reference's mAP3D / mGIoU3D / Metric outputs (tests/golden/metric.npz)."""
    from oracle import metric_oracle as MO
    g = golden("metric.npz")
    ci = 0
    while f"c{ci}_class" in g:
        B = int(g[f"c{ci}_B"])
        out = {k: torch.from_numpy(g[f"c{ci}_{k}"]) for k in ("center", "size", "angle", "class")}
        tgts = [{k: torch.from_numpy(g[f"c{ci}_t{b}_{k}"]) for k in ("gt_center", "gt_size", "gt_angle", "gt_class")}
                for b in range(B)]
        res = MO.metric_forward(out, tgts)
        for k in ("mAP", "mGIoU"):
            ref = float(g[f"c{ci}_{k}"])
            assert abs(float(res[k]) - ref) <= 1e-5 * max(1.0, abs(ref)), (ci, k, float(res[k]), ref)
        ci += 1
    assert ci == 4


def synthetic_tesseract(E, A, seed):
    rs = np.random.RandomState(seed)
    return (10.0 ** (rs.rand(64, 256, E, A) * 12.0 + 4.0)).astype(np.float32)


def test_radar_projection_oracle_matches_reference_golden(golden):
    """oracle/radar_oracle.py vs KRadarProcessor.get_radar_data outputs (tests/golden/radar_projection.npz)."""
    from oracle import radar_oracle as RO
    g = golden("radar_projection.npz")
    for ci in range(2):
        E, A = [int(v) for v in g[f"c{ci}_shape"]]
        ra, ea = RO.radar_projection(synthetic_tesseract(E, A, int(g[f"c{ci}_seed"])), g["doppler_raster"])
        np.testing.assert_allclose(ra, g[f"c{ci}_ra"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ea, g[f"c{ci}_ea"], rtol=1e-6, atol=1e-6)
        assert ra.shape == (256, A, 6) and ea.shape == (E, A, 6)


def export_cases(golden):
    """(case dict from export.json, [(outputs, targets, step), ...]) for every exporter golden case."""
    import json
    import os
    g = golden("export.npz")
    with open(os.path.join(os.path.dirname(__file__), "golden", "export.json")) as f:
        trees = json.load(f)
    for name, case in sorted(trees.items()):
        calls = []
        for si, step in enumerate(case["steps"]):
            out = {k: torch.from_numpy(g[f"{name}_s{si}_{k}"]) for k in ("class", "center", "size", "angle")}
            tgts = [{k: torch.from_numpy(g[f"{name}_s{si}_t{b}_{k}"])
                     for k in ("gt_center", "gt_size", "gt_angle", "gt_class", "description")} for b in range(case["B"])]
            calls.append((out, tgts, step))
        yield case, calls


def test_export_oracle_matches_reference_golden(golden):
    """oracle/export_oracle.py vs the file trees the reference's KRadarExporter wrote (tests/golden/export.json)."""
    from oracle import export_oracle as EO
    n = 0
    for case, calls in export_cases(golden):
        tree = {}
        for out, tgts, step in calls:
            for path, text in EO.export_tree(out, tgts, step, categories=case["categories"]).items():
                tree[path] = tree.get(path, "") + text
        assert sorted(tree) == sorted(case["tree"]), set(tree) ^ set(case["tree"])
        for path, text in case["tree"].items():
            assert tree[path] == text, path
        assert any("dummy" in t for t in case["tree"].values())          # the placeholder path is exercised
        n += 1
    assert n == 2


KRADAR_LOSS_WEIGHTS = {"total_class": 1.0, "object_class": 0.0, "center": 1.0, "size": 1.0, "angle": 1.0}


def _assign_case(g, ci):
    B = int(g[f"c{ci}_B"])
    out = {k: T(g[f"c{ci}_{k}"]) for k in ("class", "center", "size", "angle")}
    tgts = [{k: T(g[f"c{ci}_t{b}_{k}"]) for k in ("gt_center", "gt_size", "gt_angle", "gt_class")} for b in range(B)]
    return B, out, tgts


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_hungarian_and_loss_forward_match_reference_golden(golden, ci):
    """the oracle's cost matrix, assignment, weighted batch losses, total and input gradients vs the
    reference's own HungarianAnassigner.forward (assigner.py:58-143) and Loss.forward (loss.py:486-564) -- empty
    targets, a degenerate box and more targets than queries included."""
    g = golden("assign.npz")
    B, out, tgts = _assign_case(g, ci)
    for b, tgt in enumerate(tgts):
        if tgt["gt_center"].shape[0] == 0:
            assert f"c{ci}_b{b}_i" not in g
            continue
        i, j, C = O.hungarian({k: v[b] for k, v in out.items()}, tgt, KRADAR_LOSS_WEIGHTS)
        close(C, g[f"c{ci}_b{b}_cost"], rtol=1e-5, atol_scale=1e-6)
        assert torch.equal(i, T(g[f"c{ci}_b{b}_i"])) and torch.equal(j, T(g[f"c{ci}_b{b}_j"]))     # bit-exact indices
    leaf = {k: v.clone().requires_grad_(True) for k, v in out.items()}
    total, batch_losses = O.loss_forward(leaf, tgts, KRADAR_LOSS_WEIGHTS)
    close(total, g[f"c{ci}_total"], rtol=1e-5, atol_scale=1e-6)
    for k, v in batch_losses.items():
        close(v, g[f"c{ci}_loss_{k}"], rtol=1e-5, atol_scale=1e-6)
    total.backward()
    for k in out:
        close(leaf[k].grad, g[f"c{ci}_grad_{k}"], rtol=1e-5, atol_scale=1e-6)


def test_msda_module_initial_parameters_equal_the_reference_seeded_init():
    """tests/golden/msda_init.npz (oracle/gen_golden.py: gen_msda_init): the reference's MSDeformAttn built under
    torch.manual_seed(1234).  The product module written in this repo's own form (VERDICT r4 housekeeping) must draw the same
    random numbers in the same order and produce the same star pattern: every tensor bit for bit."""
    import os
    import numpy as np
    import torch
    from dpft_amd.models.layers.ms_deform_attn import MSDeformAttn
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msda_init.npz"))
    for tag, geom in (("fuser", (16, 5, 8, 4)), ("wide", (64, 3, 4, 2))):
        torch.manual_seed(1234)
        mod = MSDeformAttn(*geom)
        sd = mod.state_dict()
        keys = [k[len(tag) + 1:] for k in gold.files if k.startswith(tag + ".") and not k.startswith(tag + ".rand.")
                and k[len(tag) + 1:] in sd]
        assert sorted(keys) == sorted(sd.keys()), (keys, list(sd))
        for k in keys:
            assert torch.equal(sd[k], torch.from_numpy(gold[f"{tag}.{k}"])), (tag, k)
