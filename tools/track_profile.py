"""cProfile of the tracker loop (tools/track_bench.py's single-sequence path): where the host time of a frame goes."""
import cProfile
import io
import os
import pstats
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lib.test.tracker.uvltrack import UVLTrack  # noqa: E402
from lib.test.utils import TrackerParams  # noqa: E402
from uvltrack_amd import weightgen as wg  # noqa: E402
from uvltrack_amd.model import ModalityAdaptiveBoxHead, ModalityUnifiedFeatureExtractor  # noqa: E402
from uvltrack_amd.model import UVLTrack as Net  # noqa: E402
from uvltrack_amd.spec import spec_b  # noqa: E402

ns = lambda **kw: types.SimpleNamespace(**kw)
spec = spec_b(128, 256)
net = Net(ModalityUnifiedFeatureExtractor(spec), ModalityAdaptiveBoxHead(spec), max_batch=1)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in wg.make_state_dict(spec, 0, include_unused=True).items()}, strict=True)
cfg = ns(TEST=ns(UPDATE_INTERVAL=20, THRESHOLD=0.0, MODE="BBOX"), TRAIN=ns(CONT_WEIGHT=1.0),
         MODEL=ns(BACKBONE=ns(LANGUAGE=ns(VOCAB_PATH="", BERT=ns(MAX_QUERY_LEN=spec.text_len)))))
p = TrackerParams()
p.cfg, p.template_factor, p.template_size, p.search_factor, p.search_size, p.grounding_size, p.debug = cfg, 2.0, spec.template_size, 4.0, 256, 256, 0
rng = np.random.default_rng(0)
frames = [rng.integers(0, 256, size=(720, 1280, 3), dtype=np.uint8) for _ in range(8)]
trk = UVLTrack(p, "synthetic", network=net)
trk.initialize(frames[0], {"init_bbox": [600.0, 320.0, 90.0, 70.0]})
for i in range(30):
    trk.track(frames[i % 8])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(300):
    trk.track(frames[i % 8])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue())
