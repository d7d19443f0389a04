"""registry.MODELS['uvltrack'] / build_model(cfg) -- reference lib/models/uvltrack/uvltrack.py:47-57."""
from lib import registry
from lib.models.backbones import *  # noqa: F401,F403  (registers BACKBONES)
from lib.models.heads import *  # noqa: F401,F403      (registers HEADS)
from uvltrack_amd.model import UVLTrack, build_model as _build_model  # noqa: F401


@registry.MODELS.register('uvltrack')
def build_model(cfg):
    return _build_model(cfg)
