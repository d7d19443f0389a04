"""registry.BACKBONES['modality_unified_feature_extractor'] -- reference lib/models/backbones/__init__.py:4-7."""
from lib import registry
from uvltrack_amd.model import ModalityUnifiedFeatureExtractor, build_backbone  # noqa: F401


@registry.BACKBONES.register('modality_unified_feature_extractor')
def build_modality_unified_feature_extractor(cfg):
    return build_backbone(cfg)
