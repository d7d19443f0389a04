"""Tolerances of the bf16 HIP path against fp32 reference outputs (stated once, used by every parity test).

The HIP path keeps the residual stream in f32 and feeds bf16 operands to the MFMAs (f32 accumulate).  Gates
(tightened where the f32 residual allows):
    bbox_map, cls_score(_test), pred_boxes   atol 1e-2   (measured 6e-3 on UVLTrack-B with the wide-range synthetic head)
    cont_score                               atol 5e-2   (values in [-1, 2], scale 14.3)
    logits                                   atol 0.15   (values up to ~10)
    search/template/text/tokens              3 % of the tensor's abs-max
    pred_boxes                               tie-aware: the reference score at our argmax must be within 1e-2
                                             of the reference max, then that bbox_map row must match to 1e-2
    pred_boxes IoU                           REPORTED, not gated: IoU of the predicted box with the reference's box (the reference's
                                             bbox_map row at our argmax -- tie-aware), and mean / minimum IoU over the cells of bbox_map
                                             whose reference box is at least 0.1 wide and high.  The round-2 review proposed a gate of
                                             0.98; measured on MI355X it is not a usable gate on these synthetic heads: the random-weight
                                             towers of UVLTrack-L emit boxes down to ~0.03 of the crop (IoU 0.48 at a 1.4e-2 coordinate
                                             error, 0.0 for degenerate cells), and UVLTrack-B reaches 0.90-0.97 on boxes ~0.4 wide at its
                                             7e-3 error -- the same level as the HIP-free bf16 emulation of the precision plan.  In
                                             tracker terms the gates above are 2.6 px of a 256-px crop (B) and 7.7 px of a 384-px crop (L).
"""
import numpy as np

ATOL = {"bbox_map": 1e-2, "cls_score": 1e-2, "cls_score_test": 1e-2, "cont_score": 5e-2, "logits": 0.15}
REL_ABSMAX = {"search": 0.03, "template": 0.03, "text": 0.03, "vis_token": 0.03, "txt_token": 0.03}


def box_iou_cxcywh(a, b):
    """IoU of boxes [..., 4] = (cx, cy, w, h) (normalised units, head:108-119); degenerate boxes give 0."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ax0, ay0, ax1, ay1 = a[..., 0] - a[..., 2] / 2, a[..., 1] - a[..., 3] / 2, a[..., 0] + a[..., 2] / 2, a[..., 1] + a[..., 3] / 2
    bx0, by0, bx1, by1 = b[..., 0] - b[..., 2] / 2, b[..., 1] - b[..., 3] / 2, b[..., 0] + b[..., 2] / 2, b[..., 1] + b[..., 3] / 2
    iw = np.clip(np.minimum(ax1, bx1) - np.maximum(ax0, bx0), 0, None)
    ih = np.clip(np.minimum(ay1, by1) - np.maximum(ay0, by0), 0, None)
    inter = iw * ih
    union = np.clip(a[..., 2], 0, None) * np.clip(a[..., 3], 0, None) + np.clip(b[..., 2], 0, None) * np.clip(b[..., 3], 0, None) - inter
    return np.where(union > 0, inter / np.maximum(union, 1e-30), 0.0)


def softmax_np(x):
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def compare_outputs(got, ref, skip=(), depth=12):
    """got/ref: dict name -> ndarray (ref may hold '<name>.slice' = [:, :8, :32]).  Returns (ok, report).
    bf16 rounding noise grows with the number of layers: the absolute gates are stated for the 12-layer
    UVLTrack-B and scale linearly with depth (x2 for the 24-layer UVLTrack-L, never below x1)."""
    report, ok = {}, True
    scale = max(1.0, depth / 12.0)
    for k, v in ref.items():
        if k in ("flag", "prompt_init") or k.startswith("fwd.") or k in skip or k.split(".")[0] in skip:
            continue
        name = k[:-6] if k.endswith(".slice") else k
        if name not in got or name == "pred_boxes":
            continue
        g = np.asarray(got[name], dtype=np.float32)
        if k.endswith(".slice"):
            g = g[:, :8, :32]
        if g.shape != v.shape:
            report[k] = "shape %s vs %s" % (g.shape, v.shape)
            ok = False
            continue
        err = float(np.abs(g - v).max()) if np.isfinite(g).all() else float("inf")
        if name in ATOL:
            tol = ATOL[name] * scale
        else:
            tol = REL_ABSMAX.get(name, 0.03) * float(np.abs(v).max())
        report[k] = (err, tol)
        ok &= err <= tol
    # tie-aware argmax / pred_boxes
    if "pred_boxes" in ref and "argmax" in got and "cls_score_test" in ref and "cont_score" in ref:
        B = ref["pred_boxes"].shape[0]
        score = ref["cls_score_test"].reshape(B, -1) * softmax_np(ref["cont_score"])[:, :, 0]
        idx = np.asarray(got["argmax"]).reshape(-1)
        gap = float((score.max(-1) - score[np.arange(B), idx]).max())
        box_err = float(np.abs(np.asarray(got["pred_boxes"])[:, 0] - ref["bbox_map"][np.arange(B), idx]).max())
        report["pred_boxes(tie-aware)"] = (max(gap, box_err), 1e-2 * scale)
        ok &= gap <= 1e-2 * scale and box_err <= 1e-2 * scale
        # the same statement in tracker terms (informational): IoU of our box with the reference's box for the cell we chose, and
        # over all cells whose reference box is at least 0.1 x 0.1 of the crop
        mine = np.asarray(got["pred_boxes"])[:, 0]
        iou = box_iou_cxcywh(mine, ref["bbox_map"][np.arange(B), idx])
        report["pred_boxes IoU (informational)"] = "min %.4f over %d samples; box sizes %s" % (
            float(iou.min()), B, np.round(ref["bbox_map"][np.arange(B), idx][:, 2:], 3).tolist())
        if "bbox_map" in got:
            big = (ref["bbox_map"][..., 2] >= 0.1) & (ref["bbox_map"][..., 3] >= 0.1)
            if big.any():
                cell_iou = box_iou_cxcywh(np.asarray(got["bbox_map"]), ref["bbox_map"])[big]
                report["bbox_map IoU, cells >= 0.1 wide (informational)"] = "min %.4f mean %.4f over %d cells" % (
                    float(cell_iou.min()), float(cell_iou.mean()), int(big.sum()))
    return ok, report


def pred_box_iou(got, ref):
    """min over samples of the IoU of got's predicted box with the reference box of the cell got chose (tie-aware)."""
    B = ref["pred_boxes"].shape[0]
    idx = np.asarray(got["argmax"]).reshape(-1) if "argmax" in got else \
        (np.asarray(got["cls_score_test"]).reshape(B, -1) * softmax_np(np.asarray(got["cont_score"]))[:, :, 0]).argmax(-1)
    return float(box_iou_cxcywh(np.asarray(got["pred_boxes"])[:, 0], ref["bbox_map"][np.arange(B), idx]).min())


def fmt_report(report):
    out = []
    for k, v in report.items():
        if isinstance(v, tuple):
            out.append("%-24s err %.3e  tol %.3e  %s" % (k, v[0], v[1], "ok" if v[0] <= v[1] else "FAIL"))
        else:
            out.append("%-24s %s" % (k, v))
    return "\n".join(out)
