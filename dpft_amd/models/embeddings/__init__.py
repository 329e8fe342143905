from torch import nn

from dpft_amd.models.embeddings.sinusoidal import build_sinusoidal_embedding


def build_embedding(name: str, *args, **kwargs) -> nn.Module:
    """src/dprt/models/embeddings/__init__.py:6-8"""
    if "sinusoidal" in name:
        return build_sinusoidal_embedding(*args, **kwargs)
    raise ValueError(f"unknown embedding {name!r}")
