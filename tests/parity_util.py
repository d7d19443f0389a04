"""Tolerances of the bf16 HIP path against fp32 reference outputs (stated once, used by every parity test).

The HIP path keeps the residual stream in f32 and feeds bf16 operands to the MFMAs (f32 accumulate).  Gates
(tightened where the f32 residual allows):
    bbox_map, cls_score(_test), pred_boxes   atol 1e-2   (measured 6e-3 on UVLTrack-B with the wide-range synthetic head)
    cont_score                               atol 5e-2   (values in [-1, 2], scale 14.3)
    logits                                   atol 0.15   (values up to ~10)
    search/template/text/tokens              3 % of the tensor's abs-max
    pred_boxes                               tie-aware: the reference score at our argmax must be within 1e-2
                                             of the reference max, then that bbox_map row must match to 1e-2
"""
import numpy as np

ATOL = {"bbox_map": 1e-2, "cls_score": 1e-2, "cls_score_test": 1e-2, "cont_score": 5e-2, "logits": 0.15}
REL_ABSMAX = {"search": 0.03, "template": 0.03, "text": 0.03, "vis_token": 0.03, "txt_token": 0.03}


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
    return ok, report


def fmt_report(report):
    out = []
    for k, v in report.items():
        if isinstance(v, tuple):
            out.append("%-24s err %.3e  tol %.3e  %s" % (k, v[0], v[1], "ok" if v[0] <= v[1] else "FAIL"))
        else:
            out.append("%-24s %s" % (k, v))
    return "\n".join(out)
