"""Box helpers used by the tracker (reference lib/utils/box_ops.py:7-31,117-126)."""
import torch


def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_cxcywh_to_xywh(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, w, h], dim=-1)


def box_xywh_to_xyxy(x):
    x1, y1, w, h = x.unbind(-1)
    return torch.stack([x1, y1, x1 + w, y1 + h], dim=-1)


def box_xywh_to_cxcywh(x):
    x1, y1, w, h = x.unbind(-1)
    return torch.stack([x1 + w / 2, y1 + h / 2, w, h], dim=-1)


def clip_box(box: list, H, W, margin=0):
    """Keep at least `margin` pixels of the box inside the H x W frame (reference box_ops.py:117-126)."""
    x1, y1, w, h = box
    x2, y2 = x1 + w, y1 + h
    x1 = min(max(0, x1), W - margin)
    x2 = min(max(margin, x2), W)
    y1 = min(max(0, y1), H - margin)
    y2 = min(max(margin, y2), H)
    return [x1, y1, max(margin, x2 - x1), max(margin, y2 - y1)]
