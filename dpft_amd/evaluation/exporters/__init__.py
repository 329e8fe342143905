"""Exporters, mirror of ``src/dprt/evaluation/exporters/__init__.py`` (``build(name, config)``)."""
from dpft_amd.evaluation.exporters.kradar import KRadarExporter, build_kradar   # noqa: F401


def build(name: str, *args, **kwargs):
    if name == "kradar":
        return build_kradar(*args, **kwargs)
