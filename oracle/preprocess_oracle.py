"""CPU restatement of the tracker's per-frame pre-processing.  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and nothing else.

Restates
  * `sample_target`            reference lib/train/data/processing_utils.py:159-243  (square crop around the box,
                               constant zero padding, resize to output_sz, attention mask of the padded area)
  * `Preprocessor_wo_mask`     reference lib/test/tracker/tracker_utils.py:20-29      ((x/255 - mean) / std, NCHW)
  * `grounding_resize`         reference lib/train/data/processing_utils.py:60-141    (whole frame, aspect kept, centred zero pad)

PARITY UNPINNED for the resize: the reference calls `cv2.resize` / `cv2.copyMakeBorder` (opencv-python 4.5.5.64,
uvltrack_env.yaml:266), and OpenCV is neither vendored in /root/reference nor installed in this image, so the module
cannot even be imported.  `resize_linear_u8` restates OpenCV's published INTER_LINEAR algorithm for 8-bit images
(modules/imgproc/src/resize.cpp: half-pixel centres, coefficients as 11-bit fixed point, horizontal pass into int,
vertical pass `(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`, and the exact-2x shortcut that averages 2x2
blocks).  What IS pinned here: the coordinate convention and the weights are cross-checked against
`torch.nn.functional.interpolate(mode='bilinear', align_corners=False)` in tests/test_preprocess_cpu.py (the fixed
point result must stay within one grey level of the float result), and the crop geometry is a literal restatement of
integer arithmetic (Python `round` = round-half-even).
"""
from __future__ import annotations

import math

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def crop_geometry(box, search_area_factor, height, width):
    """processing_utils.py:173-193.  Returns dict(crop_sz, x1, y1, x1_pad, x2_pad, y1_pad, y2_pad)."""
    x, y, w, h = [float(v) for v in box]
    crop_sz = math.ceil(math.sqrt(w * h) * search_area_factor)
    if crop_sz < 1:
        raise Exception("Too small bounding box.")
    x1 = int(round(x + 0.5 * w - crop_sz * 0.5))
    x2 = int(x1 + crop_sz)
    y1 = int(round(y + 0.5 * h - crop_sz * 0.5))
    y2 = int(y1 + crop_sz)
    x1_pad = int(max(0, -x1))
    x2_pad = int(max(x2 - width + 1, 0))          # sic: the "+ 1" drops the last image column whenever x2 >= width
    y1_pad = int(max(0, -y1))
    y2_pad = int(max(y2 - height + 1, 0))
    return dict(crop_sz=crop_sz, x1=x1, y1=y1, x2=x2, y2=y2, x1_pad=x1_pad, x2_pad=x2_pad, y1_pad=y1_pad, y2_pad=y2_pad)


def _axis_tables(src, dst):
    """OpenCV resize.cpp (linear, ksize 2): per destination index the source index and the two 11-bit weights."""
    scale = float(src) / float(dst)               # double, as `scale_x = 1. / inv_scale_x`
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coef(f):
    # saturate_cast<short>(float * 2048): cvRound = round half to even
    w1 = np.rint(f.astype(np.float32) * np.float32(COEF_SCALE)).astype(np.int64)
    w0 = np.rint((np.float32(1.0) - f.astype(np.float32)) * np.float32(COEF_SCALE)).astype(np.int64)
    return w0, w1


def resize_linear_u8(src: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """cv2.resize(src, (dst_w, dst_h)) for uint8 HxWxC, default interpolation (INTER_LINEAR)."""
    assert src.dtype == np.uint8 and src.ndim == 3
    sh, sw, _ = src.shape
    if (sw, sh) == (dst_w, dst_h):
        return src.copy()
    if sw == 2 * dst_w and sh == 2 * dst_h:       # resize(): INTER_LINEAR with an exact 2x2 decimation runs the INTER_AREA fast path
        s = src.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, fx = _axis_tables(sw, dst_w)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx)
    sx = np.where(lo, 0, sx)
    hi = sx >= sw - 1
    fx = np.where(hi, np.float32(0), fx)
    sx = np.where(hi, sw - 1, sx)
    a0, a1 = _coef(fx)
    sx1 = np.minimum(sx + 1, sw - 1)              # weight 0 whenever clamped
    sy, fy = _axis_tables(sh, dst_h)
    b0, b1 = _coef(fy)                            # no fy clamp: rows are clipped instead
    r0 = np.clip(sy, 0, sh - 1)
    r1 = np.clip(sy + 1, 0, sh - 1)
    s = src.astype(np.int64)
    hrow = s[:, sx, :] * a0[None, :, None] + s[:, sx1, :] * a1[None, :, None]           # [sh, dst_w, C] int
    h0 = hrow[r0]
    h1 = hrow[r1]
    out = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_linear_mask(mask: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """cv2.resize(att_mask (float64 0/1), (dst_w, dst_h)).astype(bool): True where any tap with non-zero weight is padded."""
    sh, sw = mask.shape
    if (sw, sh) == (dst_w, dst_h):
        return mask.astype(np.bool_)
    if sw == 2 * dst_w and sh == 2 * dst_h:       # same INTER_AREA shortcut as for the image (it is type-independent)
        m = mask.astype(np.float64)
        return (m[0::2, 0::2] + m[0::2, 1::2] + m[1::2, 0::2] + m[1::2, 1::2]) != 0
    sx, fx = _axis_tables(sw, dst_w)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx)
    sx = np.where(lo, 0, sx)
    hi = sx >= sw - 1
    fx = np.where(hi, np.float32(0), fx)
    sx = np.where(hi, sw - 1, sx)
    sx1 = np.minimum(sx + 1, sw - 1)
    sy, fy = _axis_tables(sh, dst_h)
    r0 = np.clip(sy, 0, sh - 1)
    r1 = np.clip(sy + 1, 0, sh - 1)
    m = mask.astype(np.float64)
    hrow = m[:, sx] * (1.0 - fx.astype(np.float64))[None, :] + m[:, sx1] * fx.astype(np.float64)[None, :]
    out = hrow[r0] * (1.0 - fy.astype(np.float64))[:, None] + hrow[r1] * fy.astype(np.float64)[:, None]
    return out != 0


def sample_target(im: np.ndarray, target_bb, search_area_factor: float, output_sz: int):
    """processing_utils.py:159-243 with mask=None.  im: HxWx3 uint8.
    Returns (patch uint8 [out,out,3], resize_factor, att_mask bool [out,out], bbox float [1,1,4])."""
    H, W, _ = im.shape
    g = crop_geometry(target_bb, search_area_factor, H, W)
    cs = g["crop_sz"]
    ys, ye = g["y1"] + g["y1_pad"], g["y2"] - g["y2_pad"]
    xs, xe = g["x1"] + g["x1_pad"], g["x2"] - g["x2_pad"]
    if ye <= ys or xe <= xs:
        raise Exception("crop does not intersect the image")
    im_crop = im[ys:ye, xs:xe, :]
    padded = np.zeros((cs, cs, 3), dtype=np.uint8)                                   # cv.copyMakeBorder(..., BORDER_CONSTANT) -> zeros
    padded[g["y1_pad"]:g["y1_pad"] + im_crop.shape[0], g["x1_pad"]:g["x1_pad"] + im_crop.shape[1]] = im_crop
    att = np.ones((cs, cs))
    end_x, end_y = -g["x2_pad"], -g["y2_pad"]
    if g["y2_pad"] == 0:
        end_y = None
    if g["x2_pad"] == 0:
        end_x = None
    att[g["y1_pad"]:end_y, g["x1_pad"]:end_x] = 0
    x, y, w, h = [float(v) for v in target_bb]
    bbox = np.array([[[0.5 - w / cs / 2, 0.5 - h / cs / 2, w / cs, h / cs]]], dtype=np.float32)
    resize_factor = output_sz / cs
    patch = resize_linear_u8(padded, output_sz, output_sz)
    att_r = resize_linear_mask(att, output_sz, output_sz)
    return patch, resize_factor, att_r, bbox


def grounding_geometry(height: int, width: int, output_sz: int):
    """processing_utils.py:77-104: resized size (aspect kept, long side = output_sz) and the centred padding."""
    crop_sz = math.ceil(1 * output_sz)
    if width > height:
        ow, oh = crop_sz, int(crop_sz * height / width)
    else:
        oh, ow = crop_sz, int(crop_sz * width / height)
    y1_pad = int((output_sz - oh) / 2)
    y2_pad = int((output_sz - oh) / 2)
    x1_pad = int((output_sz - ow) / 2)
    x2_pad = int((output_sz - ow) / 2)
    if (y1_pad + y2_pad + oh) != output_sz:
        y1_pad += 1
    if (x1_pad + x2_pad + ow) != output_sz:
        x1_pad += 1
    return dict(new_w=ow, new_h=oh, x1_pad=x1_pad, x2_pad=x2_pad, y1_pad=y1_pad, y2_pad=y2_pad)


def grounding_resize(im: np.ndarray, output_sz: int, bbox):
    """processing_utils.py:60-141 with mask=None: whole frame resized with the aspect ratio kept, zero-padded to a square.
    NB the reference writes `cv2.resize(im, (ow, oh), interpolation)` with interpolation = PIL's BILINEAR (= 2): the third
    POSITIONAL parameter of cv2.resize is `dst`, so the value is swallowed and OpenCV's default INTER_LINEAR runs -- restated so.
    Returns (padded uint8 [out,out,3], box [4] normalised, att_mask [out,out] float 0/1, image_top_coords [x1_pad,y1_pad,new_w,new_h])."""
    h, w = im.shape[:2]
    g = grounding_geometry(h, w, output_sz)
    img = resize_linear_u8(im, g["new_w"], g["new_h"])
    padded = np.zeros((output_sz, output_sz, 3), dtype=np.uint8)
    padded[g["y1_pad"]:g["y1_pad"] + g["new_h"], g["x1_pad"]:g["x1_pad"] + g["new_w"]] = img
    box = np.array(bbox, dtype=np.float64).copy()
    box[0] = bbox[0] * g["new_w"] / w
    box[1] = bbox[1] * g["new_h"] / h
    box[2] = bbox[2] * g["new_w"] / w
    box[3] = bbox[3] * g["new_h"] / h
    box[0] += g["x1_pad"]
    box[1] += g["y1_pad"]
    box /= output_sz
    att = np.ones((output_sz, output_sz))
    att[g["y1_pad"]:output_sz - g["y2_pad"], g["x1_pad"]:output_sz - g["x2_pad"]] = 0
    return padded, box, att, [g["x1_pad"], g["y1_pad"], g["new_w"], g["new_h"]]


def normalize(img_u8: np.ndarray) -> np.ndarray:
    """tracker_utils.py:25-28: HxWx3 uint8 -> float32 [1,3,H,W], ((x / 255) - mean) / std."""
    t = img_u8.astype(np.float32).transpose(2, 0, 1)[None]
    return ((t / np.float32(255.0)) - MEAN.reshape(1, 3, 1, 1)) / STD.reshape(1, 3, 1, 1)
