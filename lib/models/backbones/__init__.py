"""Backbone registry entry of the per-frame path.  Config key MODEL.BACKBONE.TYPE = 'modality_unified_feature_extractor'
resolves to the HIP-backed extractor (the reference registers its eager module under the same key,
lib/models/backbones/__init__.py:4-7)."""
from lib.registry import BACKBONES
from uvltrack_amd.model import ModalityUnifiedFeatureExtractor, build_backbone  # noqa: F401

BACKBONES.register("modality_unified_feature_extractor", build_backbone)
