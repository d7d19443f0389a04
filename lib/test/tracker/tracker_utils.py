"""`lib.test.tracker.tracker_utils` of the reference, reduced to the pre-processor the UVLTrack tracker uses
(`Preprocessor_wo_mask`, reference lib/test/tracker/tracker_utils.py:20-29; instantiated at lib/test/tracker/uvltrack.py:29)."""
from uvltrack_amd.preprocess import Preprocessor_wo_mask  # noqa: F401
