"""Head registry entry of the per-frame path.  Config key MODEL.HEAD.TYPE = 'modality_adaptive_box_head' resolves to the
HIP-backed head (same key as the reference, lib/models/heads/__init__.py:4-13; every switch it reads from cfg.MODEL.HEAD is
read by uvltrack_amd.spec.spec_from_cfg)."""
from lib.registry import HEADS
from uvltrack_amd.model import ModalityAdaptiveBoxHead, build_head  # noqa: F401

HEADS.register("modality_adaptive_box_head", build_head)
