"""GPU parity of the device pre-processing (uvl_sample_target / uvl_normalize_u8) against
oracle/preprocess_oracle.py: the uint8 patch and the attention mask are bit-exact (integer arithmetic), the normalised
image is exact up to one float rounding (atol 1e-6)."""
import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as P
from tests.test_preprocess_cpu import _rand_img

pytestmark = pytest.mark.gpu

CASES = [
    # (H, W, box xywh, factor, out)               what it exercises
    (480, 640, [300, 200, 60, 45], 4.0, 256),     # inside the image, down-scale
    (480, 640, [310, 215, 20, 12], 4.0, 256),     # small target: up-scale
    (480, 640, [-30, -25, 90, 70], 4.0, 256),     # top-left border
    (480, 640, [600, 430, 80, 90], 4.0, 256),     # bottom-right border (incl. the dropped last row / column)
    (480, 640, [565, 200, 50, 50], 2.0, 128),     # x2 == width: one padded column
    (720, 1280, [500, 300, 128, 128], 2.0, 128),  # crop_sz == 2 * out: 2x2 averaging shortcut
    (720, 1280, [500, 300, 64, 64], 2.0, 128),    # crop_sz == out: plain copy
    (720, 1280, [75.5, 75.5, 50, 50], 2.0, 128),  # round-half-even corner
    (1080, 1920, [900, 500, 333, 217], 5.0, 384), # UVLTrack-L search size, heavy down-scale
    (240, 320, [10, 5, 300, 230], 4.0, 256),      # crop much larger than the frame
]


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_sample_target_matches_oracle(case):
    from uvltrack_amd.preprocess import sample_target_fused
    H, W, box, factor, out = case
    img = _rand_img(H, W, seed=H + W + out)
    patch, rf, att, bbox = P.sample_target(img, box, factor, out)
    r = sample_target_fused(torch.from_numpy(img).cuda(), box, factor, out)
    torch.cuda.synchronize()
    got = r["patch"].cpu().numpy()
    assert np.array_equal(got, patch), "uint8 patch differs in %d values (max %d)" % ((got != patch).sum(), np.abs(got.astype(int) - patch.astype(int)).max())
    assert np.array_equal(r["att_mask"].cpu().numpy(), att)
    assert abs(r["resize_factor"] - rf) < 1e-9
    assert np.allclose(r["bbox"].numpy(), bbox, atol=1e-6)
    assert np.abs(r["image"].cpu().numpy() - P.normalize(patch)).max() <= 1e-6


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[3], CASES[8], CASES[9]], ids=["in", "tl", "br", "big", "huge"])
def test_window_upload_matches_oracle(case):
    from uvltrack_amd.preprocess import WindowUploader
    H, W, box, factor, out = case
    img = _rand_img(H, W, seed=H + W + out)
    patch, rf, att, _ = P.sample_target(img, box, factor, out)
    up = WindowUploader()
    r = up.sample_target(img, box, factor, out, want_patch=True, want_mask=True)
    torch.cuda.synchronize()
    assert np.array_equal(r["patch"].cpu().numpy(), patch)
    assert np.array_equal(r["att_mask"].cpu().numpy(), att)
    assert np.abs(r["image"].cpu().numpy() - P.normalize(patch)).max() <= 1e-6 and abs(r["resize_factor"] - rf) < 1e-9


def test_reference_call_signatures():
    """The tracker's own call sequence (lib/test/tracker/uvltrack.py:89-101,110-112) through the mirrored modules."""
    from lib.train.data.processing_utils import sample_target
    from lib.test.tracker.tracker_utils import Preprocessor_wo_mask
    img = _rand_img(480, 640, seed=42)
    box = [250.0, 180.0, 70.0, 50.0]
    z_patch, _, _, bbox = sample_target(img, box, 2.0, output_sz=128, return_bbox=True)          # numpy frame: uploaded as uint8
    x_patch, resize_factor, x_amask = sample_target(torch.from_numpy(img).cuda(), torch.tensor(box), 4.0, output_sz=256)
    pre = Preprocessor_wo_mask()
    template, search = pre.process(z_patch), pre.process(x_patch)
    torch.cuda.synchronize()
    oz = P.sample_target(img, box, 2.0, 128)
    ox = P.sample_target(img, box, 4.0, 256)
    assert template.shape == (1, 3, 128, 128) and search.shape == (1, 3, 256, 256) and search.is_cuda
    assert np.array_equal(z_patch.cpu().numpy(), oz[0]) and np.array_equal(x_patch.cpu().numpy(), ox[0])
    assert np.array_equal(x_amask.cpu().numpy(), ox[2]) and abs(resize_factor - ox[1]) < 1e-9
    assert np.allclose(bbox.numpy(), oz[3], atol=1e-6)
    assert np.abs(search.cpu().numpy() - P.normalize(ox[0])).max() <= 1e-6
    with pytest.raises(NotImplementedError):
        sample_target(img, box, 4.0)                       # output_sz=None is not a tracker call


@pytest.mark.parametrize("hw_out", [(480, 640, 256), (720, 1280, 256), (1080, 1920, 384), (640, 480, 256), (512, 512, 256), (333, 1001, 320), (256, 256, 256)])
def test_grounding_resize_matches_oracle(hw_out):
    from lib.train.data.processing_utils import grounding_resize
    H, W, out = hw_out
    img = _rand_img(H, W, seed=H * 3 + W)
    bbox = [0.1 * W, 0.2 * H, 0.3 * W, 0.25 * H]
    padded, box, att, top = P.grounding_resize(img, out, bbox)
    got = grounding_resize(img, out, torch.tensor(bbox), None, want_norm=True)
    torch.cuda.synchronize()
    assert np.array_equal(got[0].cpu().numpy(), padded)
    assert np.allclose(got[1].numpy(), box, atol=1e-6)
    assert np.array_equal(got[2].cpu().numpy(), att)
    assert got[4] == top and tuple(got[3].shape) == (out, out)
    assert np.abs(got[5].cpu().numpy() - P.normalize(padded)).max() <= 1e-6


def test_rejects_bad_input():
    from uvltrack_amd.preprocess import sample_target_fused
    from uvltrack_amd._native import NativeLibraryError
    img = torch.zeros((100, 100, 3), dtype=torch.uint8).cuda()
    with pytest.raises(NativeLibraryError):
        sample_target_fused(img, [500, 500, 10, 10], 2.0, 128)      # crop entirely outside the frame
    with pytest.raises(ValueError):
        sample_target_fused(torch.zeros((100, 100, 3)).cuda(), [10, 10, 10, 10], 2.0, 128)    # float frame
