"""`lib.train.data.processing_utils` of the reference, reduced to the functions the tracker calls:
`sample_target` (reference lib/train/data/processing_utils.py:159-243) and `grounding_resize` (:60-141), backed by the HIP
pre-processing kernels."""
from uvltrack_amd.preprocess import grounding_resize, sample_target, sample_target_fused  # noqa: F401
