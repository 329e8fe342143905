from dpft_amd.training.loss import build_loss  # noqa: F401
