"""registry.HEADS['modality_adaptive_box_head'] -- reference lib/models/heads/__init__.py:4-13."""
from lib import registry
from uvltrack_amd.model import ModalityAdaptiveBoxHead, build_head  # noqa: F401


@registry.HEADS.register('modality_adaptive_box_head')
def build_modality_adaptive_box_head(cfg):
    return build_head(cfg)
