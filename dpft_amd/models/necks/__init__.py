from dpft_amd.models.necks.fpn import build_fpn


def build_neck(name: str, *args, **kwargs):
    """src/dprt/models/necks/__init__.py:4-6"""
    if "fpn" in name.lower():
        return build_fpn(name, *args, **kwargs)
    raise ValueError(f"unknown neck {name!r}")
