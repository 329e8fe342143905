"""CPU-only checks: C-ABI surface, host-side mirror of the reference interface, configs."""
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
