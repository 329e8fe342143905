from dpft_amd.models.heads.detection import build_detection_head


def build_head(name: str, *args, **kwargs):
    """src/dprt/models/heads/__init__.py:4-6"""
    if "detection" in name.lower():
        return build_detection_head(name, *args, **kwargs)
    raise ValueError(f"unknown head {name!r}")
