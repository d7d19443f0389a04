"""`lib.train.data.processing_utils` of the reference, reduced to the function the per-frame path calls:
`sample_target` (reference lib/train/data/processing_utils.py:159-243), backed by the HIP pre-processing kernel."""
from uvltrack_amd.preprocess import sample_target, sample_target_fused  # noqa: F401
