"""Evaluation-side pieces: the per-step metrics, the K-Radar exporter and the evaluation loop."""
from dpft_amd.evaluation.evaluator import CentralizedEvaluator, build_evaluator   # noqa: F401
from dpft_amd.evaluation.metric import Metric, build_metric   # noqa: F401
