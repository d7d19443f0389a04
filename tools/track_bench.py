"""End-to-end tracker loop on synthetic video: lib/test/tracker/uvltrack.py of this repository (uint8 crop-window upload,
fused pre-processing, forward_test, on-device decode, one small read-back per frame, prompt updates), UVLTrack-B with the
reference's yaml sizes unless --z256.  Prints frames/s of `track()` including every host step."""
import argparse
import os
import sys
import time
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


def ns(**kw):
    return types.SimpleNamespace(**kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--z256", action="store_true", help="256x256 template (BASELINE.json sizes) instead of the yaml's 128")
    ap.add_argument("--update-interval", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="sequences tracked in lockstep on this GPU (BatchUVLTrack)")
    a = ap.parse_args()
    spec = spec_b(256 if a.z256 else 128, 256)
    net = Net(ModalityUnifiedFeatureExtractor(spec), ModalityAdaptiveBoxHead(spec), max_batch=max(1, a.batch))
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in wg.make_state_dict(spec, 0, include_unused=True).items()}, strict=True)
    cfg = ns(TEST=ns(UPDATE_INTERVAL=a.update_interval, THRESHOLD=0.0, MODE="BBOX"), TRAIN=ns(CONT_WEIGHT=1.0),
             MODEL=ns(BACKBONE=ns(LANGUAGE=ns(VOCAB_PATH="", BERT=ns(MAX_QUERY_LEN=spec.text_len)))))
    p = TrackerParams()
    p.cfg, p.template_factor, p.template_size, p.search_factor, p.search_size, p.grounding_size, p.debug = cfg, 2.0, spec.template_size, 4.0, 256, 256, 0
    rng = np.random.default_rng(0)
    H, W = 720, 1280
    frames = [rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8) for _ in range(8)]
    if a.batch > 1:
        from lib.test.tracker.uvltrack_batch import BatchUVLTrack
        bt = BatchUVLTrack(p, a.batch, network=net)
        bt.initialize([frames[b % 8] for b in range(a.batch)], [{"init_bbox": [600.0 - 20 * b, 320.0 + 10 * b, 90.0, 70.0]} for b in range(a.batch)])
        for i in range(20):
            bt.track([frames[(i + b) % 8] for b in range(a.batch)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.frames):
            bt.track([frames[(i + b) % 8] for b in range(a.batch)])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("batched tracker loop, %d sequences in lockstep, UVLTrack-B z%d/x256, BBOX mode, 720p host frames: %.2f ms/step = %.0f frames/s"
              % (a.batch, spec.template_size, dt / a.frames * 1e3, a.batch * a.frames / dt))
        return
    trk = UVLTrack(p, "synthetic", network=net)
    trk.initialize(frames[0], {"init_bbox": [600.0, 320.0, 90.0, 70.0]})
    for i in range(30):
        trk.track(frames[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.frames):
        trk.track(frames[i % 8])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("tracker loop, UVLTrack-B z%d/x256, BBOX mode, 720p uint8 frames on the host, prompt update every %d frames: %.2f ms/frame = %.0f frames/s"
          % (spec.template_size, a.update_interval, dt / a.frames * 1e3, a.frames / dt))


if __name__ == "__main__":
    main()
