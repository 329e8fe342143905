"""Evaluation-side pieces: the per-step metrics, and the K-Radar exporter."""
from dpft_amd.evaluation.metric import Metric, build_metric   # noqa: F401
from dpft_amd.evaluation.evaluator import DataParallelEvaluator, build_evaluator, merge_rank_exports   # noqa: F401
