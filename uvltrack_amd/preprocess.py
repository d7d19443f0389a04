"""Host side of the device pre-processing: the reference's `sample_target`
(lib/train/data/processing_utils.py:159-243) and `Preprocessor_wo_mask` (lib/test/tracker/tracker_utils.py:20-29)
with the same call signatures, backed by `uvl_sample_target` / `uvl_normalize_u8` of the HIP library.

The frame is uploaded as uint8 (1 byte per pixel) -- or is already a CUDA uint8 tensor -- and the crop, zero border,
bilinear resize and normalisation happen in one kernel.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _frame_on_device(im) -> torch.Tensor:
    """HxWx3 uint8 numpy array or tensor -> contiguous CUDA uint8 tensor (an upload of 3 bytes per pixel at most)."""
    if isinstance(im, np.ndarray):
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise ValueError("sample_target expects an HxWx3 uint8 image")
        im = torch.from_numpy(np.ascontiguousarray(im))
    if not torch.is_tensor(im) or im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
        raise ValueError("sample_target expects an HxWx3 uint8 image")
    if not torch.cuda.is_available():
        raise _native.NativeLibraryError("sample_target runs on the GPU only (no CPU fallback)")
    return im.cuda(non_blocking=True).contiguous()


def crop_geometry(target_bb, search_area_factor: float, output_sz: int, height: int, width: int) -> _native.UvlCropGeometry:
    """The integer crop geometry of sample_target (processing_utils.py:173-193), computed by the library's host code."""
    lib = _native.load()
    box = (C.c_float * 4)(*[float(v) for v in (target_bb.tolist() if hasattr(target_bb, "tolist") else target_bb)])
    g = _native.UvlCropGeometry()
    _native.check(lib.uvl_crop_geometry_of(box, float(search_area_factor), int(output_sz or 0), int(height), int(width), C.byref(g)),
                  "uvl_crop_geometry_of")
    return g


def sample_target_fused(im, target_bb, search_area_factor: float, output_sz: int, want_patch: bool = True, want_norm: bool = True,
                        want_mask: bool = True, image_out: torch.Tensor = None):
    """One launch: returns dict(patch uint8 [out,out,3] | None, image f32 [1,3,out,out] | None, att_mask bool [out,out] | None,
    resize_factor, bbox [1,1,4], geometry).  `image_out` (contiguous CUDA f32 [1,3,out,out]) receives the normalised image in
    place -- e.g. the `search` buffer a captured / pre-validated forward step reads."""
    lib = _native.load()
    frame = _frame_on_device(im)
    H, W = int(frame.shape[0]), int(frame.shape[1])
    out = int(output_sz)
    dev = frame.device
    patch = torch.empty((out, out, 3), dtype=torch.uint8, device=dev) if want_patch else None
    norm = None
    if image_out is not None:
        if not (image_out.is_cuda and image_out.dtype == torch.float32 and image_out.is_contiguous() and image_out.numel() == 3 * out * out):
            raise ValueError("image_out must be a contiguous CUDA float32 tensor of 3*out*out elements")
        norm = image_out
    elif want_norm:
        norm = torch.empty((1, 3, out, out), dtype=torch.float32, device=dev)
    att = torch.empty((out, out), dtype=torch.uint8, device=dev) if want_mask else None
    bb = [float(v) for v in (target_bb.tolist() if hasattr(target_bb, "tolist") else target_bb)]
    box = (C.c_float * 4)(*bb)
    g = _native.UvlCropGeometry()
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    _native.check(lib.uvl_sample_target(ptr(frame), H, W, int(frame.stride(0)), box, float(search_area_factor), out, ptr(patch), ptr(norm),
                                        ptr(att), C.byref(g), _stream()), "uvl_sample_target")
    x, y, w, h = bb
    cs = float(g.crop_sz)
    bbox = torch.tensor([[[0.5 - w / cs / 2, 0.5 - h / cs / 2, w / cs, h / cs]]])
    return dict(patch=patch, image=norm, att_mask=att.bool() if att is not None else None, resize_factor=out / cs, bbox=bbox, geometry=g,
                _keep=frame)


class WindowUploader:
    """Uploads only the part of a host frame that a crop needs (about crop_sz^2 * 3 bytes instead of H*W*3) through a
    pinned staging buffer, then runs the fused kernel on that window: gather on the host, ONE library call
    (`uvl_sample_target_staged` = one host-to-device copy + one launch).  The first HEADER bytes of both buffers belong to the
    caller's small operands (`with_meta`)."""

    HEADER = 256

    def __init__(self, max_side: int = 0, device="cuda"):
        """Buffers always exist: a small one from construction (or max_side x max_side pixels when given), replaced on the first
        frame by one that holds a whole frame -- a window never exceeds the frame it is cut from, so after that the buffers only
        grow if a larger frame arrives (a batch tracker holds one uploader per sequence; growing mid-sequence would stall them all)."""
        self.device = device
        self.capacity = 0
        self._resize(max(int(max_side), 64) ** 2 * 3)

    def _resize(self, pixel_bytes: int):
        n = self.HEADER + int(pixel_bytes)
        self.stage = torch.empty(n, dtype=torch.uint8).pin_memory()       # flat: every window is one contiguous copy
        self.dev = torch.empty(n, dtype=torch.uint8, device=self.device)
        self.capacity = int(pixel_bytes)
        self._stage_np = self.stage.numpy()                               # views made once: no tensor slicing per frame
        self._meta_np = self._stage_np[:28].view(np.float32)
        self._meta_dev = self.dev[:28].view(torch.float32)
        self._h_ptr, self._d_ptr = C.c_void_p(self.stage.data_ptr()), C.c_void_p(self.dev.data_ptr())

    def sample_target(self, im: np.ndarray, target_bb, search_area_factor: float, output_sz: int, image_out: torch.Tensor = None,
                      want_patch: bool = False, want_mask: bool = False, with_meta: bool = False):
        """`with_meta`: the 7 floats the tracker's decode kernel needs -- target_bb (4), resize factor, frame H, frame W -- ride
        in the header of the same host-to-device copy; returned as `meta` (float32 device tensor [7], a view of the staging
        buffer: valid until the next call)."""
        if not (isinstance(im, np.ndarray) and im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3):
            raise ValueError("WindowUploader expects an HxWx3 uint8 numpy frame")
        lib = _native.load()
        H, W = im.shape[:2]
        bb = [float(v) for v in (target_bb.tolist() if hasattr(target_bb, "tolist") else target_bb)]
        g = crop_geometry(bb, search_area_factor, output_sz, H, W)
        x0, x1 = g.x1 + g.x1_pad, g.x1 + g.crop_sz - g.x2_pad
        y0, y1 = g.y1 + g.y1_pad, g.y1 + g.crop_sz - g.y2_pad
        ww, wh = x1 - x0, y1 - y0
        if ww * wh * 3 > self.capacity:
            # grow to a whole frame (the first frame, or a larger video): the previous device buffer may still be read by a queued
            # kernel, so let the stream finish before it is released
            torch.cuda.current_stream(self.dev.device).synchronize()
            self._resize(max(H * W * 3, ww * wh * 3))
        nbytes = wh * ww * 3
        hdr = self.HEADER
        np.copyto(self._stage_np[hdr:hdr + nbytes].reshape(wh, ww, 3), im[y0:y1, x0:x1])   # host gather into pinned memory
        out = int(output_sz)
        if with_meta:
            self._meta_np[:] = (bb[0], bb[1], bb[2], bb[3], out / float(g.crop_sz), float(H), float(W))
        dev = self.dev.device
        patch = torch.empty((out, out, 3), dtype=torch.uint8, device=dev) if want_patch else None
        att = torch.empty((out, out), dtype=torch.uint8, device=dev) if want_mask else None
        norm = image_out if image_out is not None else torch.empty((1, 3, out, out), dtype=torch.float32, device=dev)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        gg = _native.UvlCropGeometry()
        _native.check(lib.uvl_sample_target_staged(self._h_ptr, self._d_ptr, hdr, x0, y0, ww, wh, H, W, (C.c_float * 4)(*bb),
                                                   float(search_area_factor), out, ptr(patch), ptr(norm), ptr(att), C.byref(gg), _stream()),
                      "uvl_sample_target_staged")
        return dict(patch=patch, image=norm, att_mask=att.bool() if att is not None else None, resize_factor=out / float(gg.crop_sz), geometry=gg,
                    meta=self._meta_dev if with_meta else None)


def sample_target(im, target_bb, search_area_factor, output_sz=None, mask=None, return_bbox=False):
    """Drop-in for processing_utils.sample_target (the tracker's call sites: lib/test/tracker/uvltrack.py:89-90,98-99,110-111).
    Returns (patch, resize_factor, att_mask[, bbox]) like the reference, with `patch` / `att_mask` as CUDA tensors
    (uint8 [out,out,3] / bool [out,out]) instead of numpy arrays; feed `patch` to `Preprocessor_wo_mask.process`."""
    if output_sz is None or mask is not None:
        raise NotImplementedError("only the tracker's form sample_target(im, bb, factor, output_sz=..., mask=None) is built")
    r = sample_target_fused(im, target_bb, search_area_factor, output_sz, want_patch=True, want_norm=False, want_mask=True)
    if return_bbox:
        return r["patch"], r["resize_factor"], r["att_mask"], r["bbox"]
    return r["patch"], r["resize_factor"], r["att_mask"]


def grounding_resize(im, output_sz, bbox, mask=None, want_norm: bool = False):
    """Drop-in for processing_utils.grounding_resize (reference lib/train/data/processing_utils.py:60-141; tracker:48): the whole
    frame, aspect ratio kept, long side = output_sz, centred on zeros.  Returns (im_crop_padded uint8 CUDA [out,out,3], box
    normalised to [0,1], att_mask float CUDA [out,out] (1 = padding), mask_crop_padded zeros [out,out], image_top_coords).
    With want_norm=True the normalised image [1,3,out,out] is appended (one launch instead of resize + Preprocessor)."""
    lib = _native.load()
    frame = _frame_on_device(im)
    H, W = int(frame.shape[0]), int(frame.shape[1])
    out = int(output_sz)
    dev = frame.device
    patch = torch.empty((out, out, 3), dtype=torch.uint8, device=dev)
    att = torch.empty((out, out), dtype=torch.uint8, device=dev)
    norm = torch.empty((1, 3, out, out), dtype=torch.float32, device=dev) if want_norm else None
    top = (C.c_int32 * 4)()
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    _native.check(lib.uvl_grounding_resize(ptr(frame), H, W, int(frame.stride(0)), out, ptr(patch), ptr(norm), ptr(att), top, _stream()),
                  "uvl_grounding_resize")
    x1_pad, y1_pad, new_w, new_h = [int(v) for v in top]
    box = torch.as_tensor(bbox, dtype=torch.float32).detach().clone().cpu()
    src = [float(v) for v in box.tolist()]
    box[0] = src[0] * new_w / W
    box[1] = src[1] * new_h / H
    box[2] = src[2] * new_w / W
    box[3] = src[3] * new_h / H
    box[0] += x1_pad
    box[1] += y1_pad
    box /= out
    res = (patch, box, att.to(torch.float64), torch.zeros(out, out), [x1_pad, y1_pad, new_w, new_h])
    return res + (norm,) if want_norm else res


class Preprocessor_wo_mask(object):
    """tracker_utils.py:20-29: uint8 HxWx3 patch -> normalised float [1,3,H,W] on the GPU."""

    def __init__(self):
        self.mean = torch.tensor([0.485, 0.456, 0.406]).view((1, 3, 1, 1))
        self.std = torch.tensor([0.229, 0.224, 0.225]).view((1, 3, 1, 1))

    def process(self, img_arr):
        lib = _native.load()
        patch = _frame_on_device(img_arr)
        H, W = int(patch.shape[0]), int(patch.shape[1])
        out = torch.empty((1, 3, H, W), dtype=torch.float32, device=patch.device)
        _native.check(lib.uvl_normalize_u8(C.c_void_p(patch.data_ptr()), H, W, C.c_void_p(out.data_ptr()), _stream()), "uvl_normalize_u8")
        return out
