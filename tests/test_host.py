"""CPU-only checks: C-ABI surface, host-side mirror of the reference interface, configs."""
import copy
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def test_cabi_exports_every_declared_symbol():
    from dpft_amd.hip.lib import LIB_PATH, SIGNATURES, lib
    header = open(os.path.join(ROOT, "include", "dpft_hip.h")).read()
    declared = set(re.findall(r"\b(dpft_[a-z0-9_]+)\s*\(", header))
    declared -= {"dpft_stream_t"}
    assert declared, "no declarations parsed"
    assert os.path.exists(LIB_PATH), "libdpft_hip.so not built (run __graft_entry__.build())"
    dll = lib.load()
    for name in sorted(declared):
        assert hasattr(dll, name), f"{name} declared in include/dpft_hip.h but not exported"
        assert name in SIGNATURES, f"{name} has no ctypes signature"
    assert set(SIGNATURES) == declared
    assert dll.dpft_version() >= 100


def test_cabi_argument_errors_are_reported():
    """Argument validation happens before any launch, so it is testable without a GPU."""
    import ctypes as C
    from dpft_amd.hip.lib import ConvDesc, lib
    bad = ConvDesc(1, 8, 8, 64, 64, 3, 3, 1, 1, 5, 5)       # OH/OW inconsistent
    assert lib.dpft_conv2d_workspace_bytes(C.byref(bad)) == -1
    assert b"inconsistent" in lib.dpft_last_error()
    rc = lib.dpft_xattn_fwd_f32(None, None, None, None, None, None, None, None, None, 1, 1, 4, 4, 4, None)
    assert rc != 0 and b"M=8" in lib.dpft_last_error()
    # compute-mode switch: 0 fp32 (default) | 1 bf16 operands | 2 three-term bf16 split; anything else is rejected
    assert lib.dpft_conv_get_compute() == 0
    assert lib.dpft_conv_set_compute(3) == -1 and b"conv_set_compute" in lib.dpft_last_error()
    assert lib.dpft_conv_set_compute(1) == 0 and lib.dpft_conv_get_compute() == 1
    assert lib.dpft_conv_set_compute(0) == 0 and lib.dpft_conv_get_compute() == 0


def test_every_compute_entry_rejects_null_arguments():
    """Error behaviour of the boundary: every compute entry validates its arguments before touching the device and
    reports DPFT_ERR_ARG + a message through dpft_last_error() (no GPU needed; run in a child process because a missing
    check would be a crash)."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cabi_null_probe.py")], capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    got = json.loads(res.stdout.strip().splitlines()[-1])
    assert len(got) >= 44, sorted(got)
    for name, (rc, err) in got.items():
        assert rc == -1, (name, rc, err)                       # DPFT_ERR_ARG
        assert err and ":" in err, (name, err)                 # "<entry>: what was wrong"


def test_state_dict_layout_matches_reference_naming():
    from dpft_amd.configs import load_config
    from dpft_amd.models import build
    model = build("dprt", load_config("kradar"))
    sd = model.state_dict()
    n_params = sum(p.numel() for p in model.parameters())
    assert n_params == 89_892_084 + 2_208                       # SURVEY App. A (+ the un-cloned template head)
    assert sd["backbones.radar_bev.adjustment_layer.weight"].shape == (3, 6, 1, 1)
    assert "backbones.camera_mono.adjustment_layer.weight" not in sd
    assert sd["backbones.camera_mono.body.conv1.weight"].shape == (64, 3, 7, 7)
    assert sd["backbones.camera_mono.body.layer3.22.conv3.weight"].shape == (1024, 256, 1, 1)
    assert sd["backbones.camera_mono.body.layer2.0.downsample.0.weight"].shape == (512, 256, 1, 1)
    assert sd["backbones.camera_mono.body.layer2.0.downsample.1.running_var"].shape == (512,)
    assert sd["backbones.radar_front.body.layer3.5.bn2.num_batches_tracked"].dtype == torch.int64
    assert "backbones.radar_front.body.layer3.6.conv1.weight" not in sd          # ResNet-50 depth
    assert sd["necks.camera_mono.fpn.inner_blocks.0.0.weight"].shape == (16, 3, 1, 1)
    assert sd["necks.radar_bev.fpn.inner_blocks.4.0.bias"].shape == (16,)
    assert sd["necks.radar_bev.fpn.layer_blocks.2.0.weight"].shape == (16, 16, 3, 3)
    assert sd["fuser.mpfusion.fusion3.ml_fusion_layers.ms_deform_attn2.ms_deform_attn.sampling_offsets.weight"].shape == (320, 16)
    assert sd["fuser.mpfusion.fusion0.reduction_layer.weight"].shape == (16, 48)
    assert sd["fuser.heads.3.layers.class_head.6.weight"].shape == (2, 16)
    assert sd["head.layers.center_head.0.weight"].shape == (16, 16)
    bn = sum(1 for k in sd if k.endswith("running_mean"))
    assert bn == 104 + 53 + 53
    # conv weights live in the [Cout][kh][kw][Cin] physical layout the HIP kernels read
    w = model.backbones["camera_mono"].body.layer1[0].conv2.weight
    assert w.permute(0, 2, 3, 1).is_contiguous()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_fuser_state_dict_identical_to_reference():
    from oracle import ref_import
    ref_import.install()
    from dprt.models.fusers import build_fuser as ref_build_fuser
    from dprt.models.heads import build_head as ref_build_head
    from dpft_amd.configs import load_config
    from dpft_amd.models.fusers import build_fuser
    from dpft_amd.models.heads import build_head
    cfg = load_config("kradar")
    comp, m = cfg["computing"], cfg["model"]
    ours = build_fuser(m["fuser"]["name"], dict(comp | m["fuser"]), head=build_head(m["head"]["name"], dict(comp | m["head"])))
    ref = ref_build_fuser(m["fuser"]["name"], dict(comp | m["fuser"]),
                          head=ref_build_head(m["head"]["name"], dict(comp | m["head"])))
    a, b = ours.state_dict(), ref.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert a[k].shape == b[k].shape, k
    # deterministic initialisations agree (ring-pattern sampling offsets, zero attention weights)
    k = "mpfusion.fusion0.ml_fusion_layers.ms_deform_attn0.ms_deform_attn.sampling_offsets.bias"
    assert torch.equal(a[k], b[k])
    ref.load_state_dict(a)        # our checkpoint loads into the reference module unchanged


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("name", ["kradar", "kradar_camera_mono", "kradar_radar_bev", "kradar_radar_front", "kradar_radar"])
def test_builtin_configs_equal_reference_jsons(name):
    from dpft_amd.configs import load_config
    ours = load_config(name)
    ref = json.load(open(os.path.join(REF, "config", name + ".json")))
    for bb in ref["model"]["backbones"].values():
        bb["weights"] = ""
    for section in ("computing", "train", "model", "evaluate"):
        assert ours[section] == ref[section], section
    path = os.path.join(REF, "config", name + ".json")
    with pytest.warns(UserWarning, match="RANDOM initialisation"):            # explicit opt-in, never silent
        loaded = load_config(path, offline=True)
    assert loaded["model"] == ours["model"]
    assert load_config(path)["model"] == json.load(open(path))["model"]      # default: the JSON unchanged
    mapped = load_config(path, weight_files={"IMAGENET1K_V2": "/w/imagenet.pt"})
    assert all(bb["weights"] == "/w/imagenet.pt" for bb in mapped["model"]["backbones"].values())


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU / HIP library (no oracle, no eager fallback)."""
    from dpft_amd.configs import load_config
    from dpft_amd.models import build
    from dpft_amd.synthetic import make_batch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = load_config("kradar_radar_front")
    model = build("dprt", cfg)
    batch = make_batch(cfg["model"]["inputs"], 1)
    with pytest.raises(RuntimeError, match="no CPU path|CUDA"):
        model(batch)
    import dpft_amd.models.dprt as d
    src = open(d.__file__).read() + open(os.path.join(ROOT, "dpft_amd", "hip", "ops.py")).read()
    assert "oracle" not in src


def test_oracle_camera_mono_cpu_plumbing():
    """BASELINE config[0]: kradar_camera_mono, batch 1, CPU forward of the restatement (reduced frame)."""
    from dpft_amd.configs import load_config
    from dpft_amd.models import build
    from dpft_amd.synthetic import make_batch
    from oracle import dprt_oracle as O
    cfg = load_config("kradar_camera_mono")
    cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"
    torch.manual_seed(0)
    model = build("dprt", cfg)
    batch = make_batch(["camera_mono"], 1, shapes={"camera_mono": (72, 128, 3)})
    out = O.dprt_forward(model.state_dict(), cfg, batch, train=False)
    assert list(out) == ["center", "size", "angle", "class"]
    assert out["center"].shape == (1, 400, 3) and out["class"].shape == (1, 400, 2)
    assert all(torch.isfinite(v).all() for v in out.values())


def test_synthetic_batch_contract():
    from dpft_amd.synthetic import make_batch, make_labels
    b = make_batch(["camera_mono", "radar_bev", "radar_front"], 4)
    assert list(b)[0] == "camera_mono" and b["camera_mono"].shape == (4, 512, 910, 3)
    assert b["radar_bev"].shape == (4, 256, 107, 6) and b["radar_front"].shape == (4, 37, 107, 6)
    assert b["camera_mono_shape"].tolist()[0] == [720, 1280, 3] and b["camera_mono_shape"].dtype == torch.int64
    assert not b["label_to_camera_mono_t"].any() and b["label_to_radar_bev_p"].shape == (4, 3, 4)
    b2 = make_batch(["camera_mono"], 4)
    assert torch.equal(b["camera_mono"], b2["camera_mono"])
    lab = make_labels(4)
    assert len(lab) == 4 and all(1 <= l["gt_center"].shape[0] <= 8 for l in lab)


def _foreign_tree(state_dict, module_path, cls_name):
    """A module tree whose classes pickle as ``<module_path>.<cls_name>`` (a class path that will NOT be importable when
    the file is read) holding exactly ``state_dict`` -- the shape of what torch.save(model) writes for the third-party
    parts (torchvision ResNet / FPN) of a reference checkpoint."""
    import sys
    import types
    mod = sys.modules.get(module_path) or types.ModuleType(module_path)
    cls = getattr(mod, cls_name, None) or type(cls_name, (torch.nn.Module,), {"__module__": module_path})
    setattr(mod, cls_name, cls)
    sys.modules[module_path] = mod
    root = cls()
    for key, t in state_dict.items():
        node, parts = root, key.split(".")
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, cls())
            node = node._modules[part]
        if parts[-1] in ("running_mean", "running_var", "num_batches_tracked"):
            node.register_buffer(parts[-1], t.clone())
        else:
            node.register_parameter(parts[-1], torch.nn.Parameter(t.clone()))
    return root


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_load_reads_reference_whole_module_checkpoint(tmp_path):
    """f-3 remainder (VERDICT r1 missing #1): ``dpft_amd.models.load`` on a ``torch.save(model)`` file whose classes are
    the REFERENCE's (dprt.models.dprt.DPRT, IMPFusion, MLFusion, MSDeformAttn, LinearDetectionHead, the querent, the
    embeddings -- built here by the imported reference itself) and torchvision's (class paths that are not importable
    when the file is read): the rebuilt dpft_amd model has the same hyper-parameters, parameters and buffers."""
    import sys
    from collections import OrderedDict
    from oracle import ref_import
    ref_import.install()
    from dprt.models.dprt import DPRT as RefDPRT
    from dprt.models.embeddings import build_embedding
    from dprt.models.fusers import build_fuser
    from dprt.models.heads import build_head
    from dprt.models.queries import build_querent
    from dpft_amd.configs import load_config
    from dpft_amd.models import build, load
    cfg = load_config("kradar")
    cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"         # keep the CPU test light
    cfg["model"]["fuser"]["dropout"] = 0.05
    comp, m = cfg["computing"], cfg["model"]
    torch.manual_seed(12)
    ours = build("dprt", cfg)                                               # source of the third-party tensors
    with torch.no_grad():
        for p in ours.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    sd = ours.state_dict()
    head = build_head(m["head"]["name"], dict(comp | m["head"]))
    fuser = build_fuser(m["fuser"]["name"], dict(comp | m["fuser"]), head=head)
    ref = RefDPRT(inputs=m["inputs"], skiplinks=m["skiplinks"],
                  embeddings={v: build_embedding(e["name"], dict(comp | e)) for v, e in m["embeddings"].items()},
                  querent=build_querent(m["querent"]["name"], dict(comp | m["querent"])), fuser=fuser, head=head)
    ref.fuser.load_state_dict({k[len("fuser."):]: v for k, v in sd.items() if k.startswith("fuser.")})
    ref.head.load_state_dict({k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")})
    made = []
    try:
        for kind, path, cname in (("backbones", "torchvision.models.resnet", "ResNet"),
                                  ("necks", "torchvision.ops.feature_pyramid_network", "FeaturePyramidNetwork")):
            made.append(path)
            holder = torch.nn.ModuleDict()
            for v in m["inputs"]:
                pre = f"{kind}.{v}."
                holder[v] = _foreign_tree(OrderedDict((k[len(pre):], t) for k, t in sd.items() if k.startswith(pre)),
                                          path, cname)
            setattr(ref, kind, holder)
        ref.eval()
        ckpt = tmp_path / "20240101-120000_checkpoint_0042.pt"
        torch.save(ref, str(ckpt))
    finally:
        for path in made:
            if path not in ref_import.STUBS:
                sys.modules.pop(path, None)
    assert b"dprt.models.fusers.mpfusion" in ckpt.read_bytes() and b"torchvision.models.resnet" in ckpt.read_bytes()
    model, epoch, stamp = load(str(ckpt))
    assert (epoch, stamp) == (42, "20240101-120000")
    assert type(model).__module__.startswith("dpft_amd.") and not model.training
    got = model.state_dict()
    assert list(got.keys()) == list(sd.keys())
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    assert model.fuser.dropout == 0.05 and model.fuser.i_iter == 4 and model.inputs == m["inputs"]
    from dpft_amd.models.checkpoint import infer_config, read_foreign
    inferred = infer_config(read_foreign(str(ckpt)))["model"]
    def covers(got, want, path):          # every key of the config with its value; extra inferred keys are checked below
        if isinstance(want, dict):
            for k, v in want.items():
                assert k in got, (path, k)
                covers(got[k], v, path + "." + k)
        else:
            assert got == want, (path, got, want)
    for section in ("inputs", "skiplinks", "backbones", "necks", "embeddings", "querent", "fuser", "head"):
        covers(inferred[section], m[section], section)
    # hyper-parameters that are not tensors are READ from the pickled modules, here the reference's defaults
    e0 = inferred["embeddings"][m["inputs"][0]]
    assert (e0["temperature"], e0["eps"], e0["offset"]) == (10000, 1e-6, 0.0) and abs(e0["scale"] - 6.283185307179586) < 1e-12
    assert inferred["querent"]["distribution"] == ["linear"] * 3 and inferred["fuser"]["ffn_layer"] == "Linear"
    assert inferred["head"]["bias"] is False and inferred["head"]["dropout"] == 0.0
    # ... and a non-default value survives the round trip instead of being replaced by a default
    foreign = read_foreign(str(ckpt))
    for layer in foreign.embeddings[m["inputs"][0]].embedding_layers.values():
        layer.__dict__["temperature"] = 20
    assert infer_config(foreign)["model"]["embeddings"][m["inputs"][0]]["temperature"] == 20
    del foreign.head.__dict__["dropout"]
    with pytest.raises(ValueError, match="dropout"):
        infer_config(foreign)
    # this package's own whole-module checkpoints keep working
    own = tmp_path / "20240101-130000_checkpoint_0001.pt"
    torch.save(ours, str(own))
    again, e2, _ = load(str(own))
    assert e2 == 1 and type(again) is type(ours)


def test_unpickler_does_not_replace_missing_native_classes(tmp_path):
    """Only FOREIGN roots (dprt, torchvision, ...) become stand-ins; a missing class of this package or of torch raises."""
    import pickle
    from dpft_amd.models.checkpoint import ForeignModule, _Unpickler
    import io
    with pytest.raises((AttributeError, ImportError)):
        _Unpickler(io.BytesIO(b"")).find_class("dpft_amd.models.dprt", "NoSuchClass")
    with pytest.raises((AttributeError, ImportError)):
        _Unpickler(io.BytesIO(b"")).find_class("torch.nn", "NoSuchLayer")
    assert issubclass(_Unpickler(io.BytesIO(b"")).find_class("torchvision.models.resnet", "ResNet"), ForeignModule)


def test_whole_module_pickle_drops_process_local_state():
    """``torch.save(model)`` of a model that has run on the GPU (trainer.py:256-258): the HIP streams, the captured decoder
    graphs and the fused inference decoder parked in ``__dict__`` must not reach the pickle (a ``torch.Stream`` cannot be
    pickled at all), nor the trainer's reducer links or device-side table caches.  The GPU round trip is
    tests/test_gpu_trainer.py; this is the mechanism, on the CPU."""
    import io
    from dpft_amd.configs import load_config
    from dpft_amd.models import build
    from dpft_amd.training.trainer import DataParallelTrainer
    cfg = copy.deepcopy(load_config("kradar"))
    cfg["model"]["backbones"]["camera_mono"]["name"] = "ResNet50"
    model = build("dprt", cfg)
    tr = DataParallelTrainer(model, cfg, "cpu")
    assert model.necks["camera_mono"].grad_direct is tr.reducer
    model.__dict__["_view_streams"] = [torch.Stream(device="cpu")]          # what DPRT._place_streams leaves behind
    model.__dict__["_queues_found"] = 3
    model.__dict__["_graphed_fuser"] = lambda: None                          # unpicklable stand-ins
    model.fuser.__dict__["_fused_decoder"] = lambda: None
    emb = model.embeddings["camera_mono"].embedding_layers["embedding0"]
    emb._tables[(1, 1, "cpu")] = (torch.zeros(1), torch.zeros(1))
    model.querent._cache[("k",)] = torch.zeros(1)
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)
    for k in ("_view_streams", "_queues_found", "_graphed_fuser"):
        assert k not in again.__dict__ and k in model.__dict__          # dropped from the pickle, kept on the live model
    assert "_fused_decoder" not in again.fuser.__dict__
    assert again.necks["camera_mono"].grad_direct is None and again.backbones["radar_bev"].grad_direct is None
    assert again.embeddings["camera_mono"].embedding_layers["embedding0"]._tables == {} and again.querent._cache == {}
    for (k, a), (k2, b) in zip(model.state_dict().items(), again.state_dict().items()):
        assert k == k2 and torch.equal(a, b) and a.stride() == b.stride(), k


def test_scheduler_factory_matches_torch():
    """``build_scheduler`` (src/dprt/training/scheduler.py:30-36): plain, chained and sequential schedules."""
    from dpft_amd.training.scheduler import build_scheduler
    def lrs(spec, n=6):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1.0)
        spec = copy.deepcopy(spec)
        sch = build_scheduler(spec.pop("name"), **spec)(opt)
        out = []
        for _ in range(n):
            out.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        return out
    assert lrs({"name": "ConstantLR", "factor": 1.0}) == [1.0] * 6
    assert lrs({"name": "StepLR", "step_size": 2, "gamma": 0.1})[:5] == pytest.approx([1, 1, 0.1, 0.1, 0.01])
    seq = {"name": "SequentialLR", "milestones": [2],
           "schedulers": [{"name": "ConstantLR", "factor": 0.5, "total_iters": 2}, {"name": "ExponentialLR", "gamma": 0.5}]}
    assert lrs(seq)[:5] == pytest.approx([0.5, 0.5, 1.0, 0.5, 0.25])
    ch = {"name": "ChainedScheduler",
          "schedulers": [{"name": "ExponentialLR", "gamma": 0.5}, {"name": "ConstantLR", "factor": 0.1, "total_iters": 1}]}
    assert lrs(ch)[:3] == pytest.approx([0.1, 0.5, 0.25])
    assert seq["schedulers"][0]["name"] == "ConstantLR"      # the config is not consumed (the reference pops from it)


def test_c_assignment_solver_returns_scipys_pairs_in_scipys_order():
    """dpft_lsap_batch_f32 (host code of the matcher, dpft_amd/csrc/cabi.cpp) vs scipy.optimize.linear_sum_assignment -- the
    call the reference makes per sample (training/loss.py:305): same pairs, same order, on random rectangular problems in both
    orientations, with many and with heavy ties, empty samples, and the -1 padding; non-finite costs are refused."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    from dpft_amd.hip.lib import lib
    rng = np.random.default_rng(0)

    def run(cost, counts):
        B, N, Mmax = cost.shape
        match = np.empty((B, Mmax, 2), np.int32)
        nm = np.empty(B, np.int32)
        cnt = np.asarray(counts, np.int32)
        return lib.dpft_lsap_batch_f32(cost.ctypes.data, B, N, Mmax, cnt.ctypes.data, match.ctypes.data, nm.ctypes.data), match, nm

    for trial in range(1200):
        B, N, Mmax = int(rng.integers(1, 5)), int(rng.integers(1, 40)), int(rng.integers(1, 12))
        cost = rng.standard_normal((B, N, Mmax)).astype(np.float32)
        if trial % 4 == 1:
            cost = np.round(cost * 2) / 2
        elif trial % 4 == 2:
            cost = np.abs(cost) * 1e3
        elif trial % 4 == 3:
            cost[:] = rng.integers(0, 3, cost.shape)
        counts = rng.integers(0, Mmax + 1, B)
        rc, match, nm = run(cost, counts)
        assert rc == 0
        for b in range(B):
            m = int(counts[b])
            if m == 0:
                assert nm[b] == 0 and (match[b] == -1).all()
                continue
            i, j = linear_sum_assignment(cost[b, :, :m])
            k = len(i)
            assert nm[b] == k and (match[b, :k, 0] == i).all() and (match[b, :k, 1] == j).all() and (match[b, k:] == -1).all(), (trial, b)
    cost = rng.standard_normal((4, 400, 7)).astype(np.float32)          # the training step's shape
    rc, match, nm = run(cost, [7, 3, 0, 1])
    assert rc == 0 and list(nm) == [7, 3, 0, 1]
    for b, m in enumerate([7, 3, 0, 1]):
        if m:
            i, j = linear_sum_assignment(cost[b, :, :m])
            assert (match[b, :m, 0] == i).all() and (match[b, :m, 1] == j).all()
    cost[1, 5, 2] = np.nan
    assert run(cost, [7, 3, 0, 1])[0] != 0 and b"non-finite" in lib.dpft_last_error()
    cost[1, 5, 2] = 0.0
    cost[3, 7, 6] = np.inf                                               # beyond the sample's count: not looked at
    assert run(cost, [7, 3, 0, 1])[0] == 0


def test_integration_shim_loads_the_library_and_exposes_the_extension_api():
    """integration/MultiScaleDeformableAttention.py: the module name and the two callables the reference imports
    (ms_deform_attn.py:24,32,58), bound to the C-ABI symbols of include/dpft_hip.h (no compute without a GPU)."""
    import importlib.util
    import inspect
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("MultiScaleDeformableAttention",
                                                  os.path.join(root, "integration", "MultiScaleDeformableAttention.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert list(inspect.signature(m.ms_deform_attn_forward).parameters) == [
        "value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight", "im2col_step"]
    assert list(inspect.signature(m.ms_deform_attn_backward).parameters) == [
        "value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight", "grad_output", "im2col_step"]
    assert m._lib.dpft_msda_fwd_f32 and m._lib.dpft_msda_bwd_f32
    import torch
    with pytest.raises(RuntimeError):      # CPU tensors are refused, not computed on some fallback
        m.ms_deform_attn_forward(torch.zeros(1, 4, 2, 2), torch.tensor([[2, 2]]), torch.tensor([0]),
                                 torch.zeros(1, 3, 2, 1, 1, 2), torch.zeros(1, 3, 2, 1, 1), 64)
