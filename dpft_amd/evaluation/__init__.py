"""Evaluation-side pieces that sit inside the reference's training loop."""
from dpft_amd.evaluation.metric import Metric, build_metric   # noqa: F401
