from dpft_amd.models.queries.data_agnostic import build_data_agnostic_query


def build_querent(name: str, *args, **kwargs):
    """src/dprt/models/queries/__init__.py:5-9 (learnable queries are not used by any config)."""
    if "data_agnostic" in name.lower():
        return build_data_agnostic_query(name, *args, **kwargs)
    raise ValueError(f"querent {name!r} is outside the dpft_amd hot path")
