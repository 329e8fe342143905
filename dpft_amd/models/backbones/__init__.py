from dpft_amd.models.backbones.resnet import build_resnet


def build_backbone(name: str, *args, **kwargs):
    """src/dprt/models/backbones/__init__.py:7-15 (only the ResNet family is on the hot path)."""
    if "resnet" in name.lower():
        return build_resnet(*args, **kwargs)
    raise ValueError(f"backbone {name!r} is outside the dpft_amd hot path (ResNet50/101/152 only)")
