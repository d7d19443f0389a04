"""CPU checks of the pre-processing oracle (oracle/preprocess_oracle.py) and of the library's host-side crop geometry.
The resize is "parity unpinned" against cv2 (absent here); what can be pinned is pinned: bilinear coordinates / weights
against torch's `interpolate(align_corners=False)`, integer geometry against a literal restatement."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import preprocess_oracle as P
from uvltrack_amd import _native


def _rand_img(h, w, seed):
    rng = np.random.default_rng(seed)
    # smooth + noise so that interpolation errors are visible but not dominated by aliasing
    base = rng.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3)).astype(np.float32)
    up = F.interpolate(torch.from_numpy(base).permute(2, 0, 1)[None], size=(h, w), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    noise = rng.integers(-20, 21, size=(h, w, 3))
    return np.clip(up + noise, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("src,dst", [(97, 128), (300, 128), (131, 256), (1000, 256), (255, 256), (257, 256), (64, 384), (513, 384)])
def test_resize_matches_float_bilinear_within_one_level(src, dst):
    img = _rand_img(src, src, seed=src * 7 + dst)
    got = P.resize_linear_u8(img, dst, dst).astype(np.float32)
    ref = F.interpolate(torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None], size=(dst, dst), mode="bilinear",
                        align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(got - ref).max() <= 1.0 + 1e-3          # 11-bit weights + two roundings stay within one grey level


def test_resize_identity_and_exact_half():
    img = _rand_img(256, 256, seed=3)
    assert np.array_equal(P.resize_linear_u8(img, 256, 256), img)
    half = P.resize_linear_u8(img, 128, 128)
    s = img.astype(np.int64)
    assert np.array_equal(half, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    # a constant image stays constant under the fixed-point arithmetic for every scale
    const = np.full((77, 77, 3), 201, dtype=np.uint8)
    assert np.all(P.resize_linear_u8(const, 128, 128) == 201)
    assert np.all(P.resize_linear_u8(const, 30, 30) == 201)


def test_crop_geometry_quirks():
    # box fully inside: no padding
    g = P.crop_geometry([100, 80, 40, 60], 4.0, 480, 640)
    assert g["crop_sz"] == math.ceil(math.sqrt(2400) * 4.0) and g["x1_pad"] == g["x2_pad"] == g["y1_pad"] == g["y2_pad"] == 0
    # crop that ends exactly at the right edge still pads (and drops) one column: the "+ 1" of processing_utils.py:185
    g = P.crop_geometry([540, 200, 50, 50], 2.0, 480, 640)         # crop 100, x1 = 515, x2 = 615 < 640 -> no pad
    assert g["x2_pad"] == 0
    g = P.crop_geometry([565, 200, 50, 50], 2.0, 480, 640)         # x1 = 540, x2 = 640 == width -> pad 1
    assert g["x2"] == 640 and g["x2_pad"] == 1
    # round-half-even of the corner: centre 100.5 - 50 = 50.5 -> 50, 101.5 - 50 = 51.5 -> 52
    assert P.crop_geometry([75.5, 75.5, 50, 50], 2.0, 480, 640)["x1"] == 50
    assert P.crop_geometry([76.5, 76.5, 50, 50], 2.0, 480, 640)["x1"] == 52
    with pytest.raises(Exception):
        P.crop_geometry([10, 10, 0.0, 5.0], 1.0, 480, 640)         # crop_sz = 0


def test_sample_target_padding_and_mask():
    img = _rand_img(240, 320, seed=11)
    patch, rf, att, bbox = P.sample_target(img, [-20, -10, 80, 60], 3.0, 128)      # sticks out top-left
    g = P.crop_geometry([-20, -10, 80, 60], 3.0, 240, 320)
    assert patch.shape == (128, 128, 3) and att.shape == (128, 128) and abs(rf - 128 / g["crop_sz"]) < 1e-12
    assert att[0, 0] and not att[-1, -1]                   # border top-left, image bottom-right
    assert np.all(patch[0, 0] == 0)                        # zero border
    assert np.allclose(bbox[0, 0, 2:], [80 / g["crop_sz"], 60 / g["crop_sz"]])
    # mask is True exactly where a non-zero-weight tap touches the border: compare with a float resize of the 0/1 mask
    cs = g["crop_sz"]
    m = np.ones((cs, cs), np.float32)
    m[g["y1_pad"]:cs - g["y2_pad"], g["x1_pad"]:cs - g["x2_pad"]] = 0
    ref = F.interpolate(torch.from_numpy(m)[None, None], size=(128, 128), mode="bilinear", align_corners=False)[0, 0].numpy() > 1e-6
    assert (ref != att).mean() < 0.01                      # identical except where a weight is below float resolution


def test_normalize_matches_reference_expression():
    img = _rand_img(64, 64, seed=5)
    t = torch.tensor(img).float().permute((2, 0, 1)).unsqueeze(dim=0)
    mean = torch.tensor([0.485, 0.456, 0.406]).view((1, 3, 1, 1))
    std = torch.tensor([0.229, 0.224, 0.225]).view((1, 3, 1, 1))
    ref = ((t / 255.0) - mean) / std                       # tracker_utils.py:27-28, verbatim arithmetic
    assert np.abs(P.normalize(img) - ref.numpy()).max() <= 1e-6


def test_library_host_geometry_matches_oracle():
    lib = _native.load()
    rng = np.random.default_rng(0)
    for _ in range(300):
        H, W = int(rng.integers(120, 1100)), int(rng.integers(160, 2000))
        w, h = float(rng.uniform(4, 400)), float(rng.uniform(4, 400))
        x, y = float(rng.uniform(-0.3 * W, 1.1 * W)), float(rng.uniform(-0.3 * H, 1.1 * H))
        if rng.random() < 0.3:                             # half-integer centres exercise round-half-even
            x, y, w, h = round(x) + 0.5, round(y) + 0.5, float(2 * round(w / 2)), float(2 * round(h / 2))
        f = float(rng.choice([2.0, 4.0, 5.0]))
        out = int(rng.choice([128, 256, 384]))
        try:
            g = P.crop_geometry([x, y, w, h], f, H, W)
            ok = g["x1"] + g["x1_pad"] < g["x2"] - g["x2_pad"] and g["y1"] + g["y1_pad"] < g["y2"] - g["y2_pad"]
        except Exception:
            ok = False
        cg = _native.UvlCropGeometry()
        box = (C.c_float * 4)(x, y, w, h)
        rc = lib.uvl_crop_geometry_of(box, f, out, H, W, C.byref(cg))
        if not ok:
            assert rc < 0
            continue
        # the ABI takes the box as float32: recompute the oracle on the float32-rounded box
        g = P.crop_geometry([float(np.float32(v)) for v in (x, y, w, h)], float(np.float32(f)), H, W)
        assert rc == 0
        assert (cg.crop_sz, cg.x1, cg.y1, cg.x1_pad, cg.x2_pad, cg.y1_pad, cg.y2_pad) == \
               (g["crop_sz"], g["x1"], g["y1"], g["x1_pad"], g["x2_pad"], g["y1_pad"], g["y2_pad"])
        assert abs(cg.resize_factor - out / g["crop_sz"]) < 1e-6


def test_grounding_geometry_and_resize():
    for (h, w, out) in [(480, 640, 256), (720, 1280, 256), (1080, 1920, 384), (640, 480, 256), (300, 300, 256), (512, 512, 256), (333, 1001, 320)]:
        g = P.grounding_geometry(h, w, out)
        assert g["x1_pad"] + g["x2_pad"] + g["new_w"] == out and g["y1_pad"] + g["y2_pad"] + g["new_h"] == out
        assert max(g["new_w"], g["new_h"]) == out
    img = _rand_img(480, 640, seed=9)
    padded, box, att, top = P.grounding_resize(img, 256, [100.0, 50.0, 200.0, 120.0])
    assert padded.shape == (256, 256, 3) and top == [0, 32, 256, 192]
    assert np.all(padded[:32] == 0) and np.all(padded[224:] == 0) and np.all(att[:32] == 1) and np.all(att[32:224] == 0)
    assert np.array_equal(padded[32:224], P.resize_linear_u8(img, 256, 192))
    assert np.allclose(box, [100 * 256 / 640 / 256, (50 * 192 / 480 + 32) / 256, 200 * 256 / 640 / 256, 120 * 192 / 480 / 256])
