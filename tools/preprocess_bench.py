"""Cost of the device pre-processing beside the forward pass: H2D of a uint8 frame (pinned host memory), the
crop+resize+normalise kernel, and a tracking-style frame = upload + pre-process + forward_test, all on one stream."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import weightgen as wg  # noqa: E402
from uvltrack_amd.engine import HipEngine  # noqa: E402
from uvltrack_amd.preprocess import sample_target_fused  # noqa: E402
from uvltrack_amd.spec import spec_b  # noqa: E402


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    H, W = 720, 1280
    rng = np.random.default_rng(0)
    host = torch.from_numpy(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)).pin_memory()
    dev = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    box = [600.0, 300.0, 90.0, 70.0]
    us_h2d = timeit(lambda: dev.copy_(host, non_blocking=True))
    us_pre = timeit(lambda: sample_target_fused(dev, box, 4.0, 256, want_patch=False, want_mask=False))
    fp32_crop = torch.empty((1, 3, 256, 256)).pin_memory()
    dcrop = torch.empty((1, 3, 256, 256), device="cuda")
    us_h2d_f32 = timeit(lambda: dcrop.copy_(fp32_crop, non_blocking=True))
    print("H2D uint8 frame %dx%d (%.2f MB, pinned): %.1f us = %.1f GB/s" % (W, H, H * W * 3 / 1e6, us_h2d, H * W * 3 / us_h2d / 1e3))
    print("H2D fp32 256x256 crop (0.79 MB, pinned; what the reference uploads): %.1f us" % us_h2d_f32)
    print("crop + resize + normalise kernel (256x256 out, incl. host launch): %.1f us" % us_pre)

    spec = spec_b(256, 256)
    eng = HipEngine(spec, torch.device("cuda:0"), max_batch=1)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in wg.make_state_dict(spec, 0).items()})
    inp = wg.make_inputs(spec, batch=1, seed=1, flags=[2])
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in inp.items()}
    step = eng.make_eager_step(t["template"], t["search"], t["ids"], t["mask"], t["prompt"], t["flag"])
    us_fwd = timeit(step, iters=300, warm=50)

    def frame():
        dev.copy_(host, non_blocking=True)
        sample_target_fused(dev, box, 4.0, 256, want_patch=False, want_mask=False, image_out=t["search"])
        step()
    us_frame = timeit(frame, iters=300, warm=50)
    from uvltrack_amd.preprocess import WindowUploader
    up = WindowUploader(max_side=1024)
    himg = host.numpy()

    def frame_win():
        up.sample_target(himg, box, 4.0, 256, image_out=t["search"])
        step()
    us_win = timeit(frame_win, iters=300, warm=50)
    print("window upload (crop region only) + pre-process + forward_test: %.1f us/frame (%.0f FPS)" % (us_win, 1e6 / us_win))
    print("forward_test alone: %.1f us/frame (%.0f FPS); upload + pre-process + forward_test: %.1f us/frame (%.0f FPS)"
          % (us_fwd, 1e6 / us_fwd, us_frame, 1e6 / us_frame))


if __name__ == "__main__":
    main()
