from dpft_amd.models.layers.ms_deform_attn import MSDeformAttn  # noqa: F401
