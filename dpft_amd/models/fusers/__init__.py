from dpft_amd.models.fusers.mpfusion import build_mpfusion


def build_fuser(name: str, *args, **kwargs):
    """src/dprt/models/fusers/__init__.py:4-6"""
    if "impfusion" in name.lower():
        return build_mpfusion(*args, **kwargs)
    raise ValueError(f"unknown fuser {name!r}")
