"""CPU restatement of the UVLTrack tracker loop (reference lib/test/tracker/uvltrack.py:45-140,167-233) built from the numpy
oracles of this directory.  TEST INFRASTRUCTURE ONLY.  The reference tracker module cannot be imported here (cv2,
pytorch_pretrained_bert and the dataset environment are absent), so this is a restatement ("parity unpinned" as a whole); its
parts are pinned individually: the network calls against reference outputs (tests/golden), clip_box against the reference
function, the tokenizer against transformers, the resize against torch's bilinear (see preprocess_oracle.py)."""
from __future__ import annotations

import numpy as np

from . import preprocess_oracle as P
from . import uvl_oracle as O

f32 = np.float32


def anno2mask(gt_bboxes: np.ndarray, size: int) -> np.ndarray:
    """tracker:183-194.  gt_bboxes [b,4] xywh normalised -> bool [b, size*size]."""
    gt = np.asarray(gt_bboxes, dtype=f32).reshape(-1, 4)
    bb = np.stack([gt[:, 0], gt[:, 1], gt[:, 0] + gt[:, 2], gt[:, 1] + gt[:, 3]], -1) * f32(size)
    cood = np.arange(size, dtype=f32)[None, :] + f32(0.5)
    x_mask = ((cood > bb[:, 0:1]) & (cood < bb[:, 2:3]))[:, None, :]
    y_mask = ((cood > bb[:, 1:2]) & (cood < bb[:, 3:4]))[:, :, None]
    mask = x_mask & y_mask
    cx = ((bb[:, 0] + bb[:, 2]) / f32(2)).astype(np.int64)
    cy = ((bb[:, 1] + bb[:, 3]) / f32(2)).astype(np.int64)
    mask[np.arange(gt.shape[0]), cy, cx] = True
    return mask.reshape(gt.shape[0], -1)


class OracleTracker:
    def __init__(self, spec, sd, mode="BBOX", template_factor=2.0, search_factor=4.0, update_interval=20, threshold=0.5, has_cont=True,
                 tokenizer=None):
        self.spec, self.sd, self.mode = spec, sd, mode
        self.template_factor, self.search_factor = template_factor, search_factor
        self.update_interval, self.threshold, self.has_cont = update_interval, threshold, has_cont
        self.tokenizer = tokenizer
        self.window = O.hann_window(spec.feat_sz)

    def _text(self, language):
        from uvltrack_amd.tokenizer import extract_token_from_nlp          # host string code, itself pinned in tests/test_tokenizer.py
        ids, mask = extract_token_from_nlp(self.tokenizer, language, self.spec.text_len)
        return np.asarray(ids, np.int64)[None], np.asarray(mask, np.int64)[None]

    def grounding(self, image, language):
        s = self.spec
        h, w = image.shape[:2]
        padded = P.grounding_resize(image, s.search_size, [0.0, 0.0, 0.0, 0.0])[0]
        ground = P.normalize(padded)
        template = np.zeros((1, 3, s.template_size, s.template_size), f32)
        self.ids, self.tmask = self._text(language)
        out = O.forward(self.sd, s, template, ground, self.ids, self.tmask, np.zeros((1, s.nz), bool), np.zeros((1, s.nx), bool),
                        np.array([[1]], np.int64))
        cx, cy, bw, bh = (out["pred_boxes"][0, 0] * f32(max(h, w))).tolist()
        box = [cx - 0.5 * bw, cy - 0.5 * bh, bw, bh]
        box[0] += min(0, (w - h) / 2)
        box[1] += min(0, (h - w) / 2)
        self.grounding_out = out
        return box

    def initialize(self, image, info):
        s = self.spec
        if self.mode == "NL":
            init_bbox = self.grounding(image, info["language"])
            self.flag = np.array([[2]], np.int64)
        elif self.mode == "NLBBOX":
            self.ids, self.tmask = self._text(info["language"])
            init_bbox = info["init_bbox"]
            self.flag = np.array([[2]], np.int64)
        else:
            self.ids, self.tmask = np.zeros((1, s.text_len), np.int64), np.zeros((1, s.text_len), np.int64)
            init_bbox = info["init_bbox"]
            self.flag = np.array([[0]], np.int64)
        z_patch, _, _, bbox = P.sample_target(image, init_bbox, self.template_factor, s.template_size)
        self.template_mask = anno2mask(bbox.reshape(1, 4), s.template_size // 16)
        self.template = P.normalize(z_patch)
        y_patch, _, _, y_bbox = P.sample_target(image, init_bbox, self.search_factor, s.search_size)
        context = P.normalize(y_patch)
        context_mask = anno2mask(y_bbox.reshape(1, 4), s.search_size // 16)
        self.prompt = O.forward_prompt_init(self.sd, s, self.template, context, self.ids, self.tmask, self.template_mask, context_mask, self.flag)
        self.state = [float(v) for v in init_bbox]
        self.frame_id = 0
        self.max_score = 0

    def track(self, image):
        s = self.spec
        H, W, _ = image.shape
        self.frame_id += 1
        x_patch, resize_factor, _, _ = P.sample_target(image, self.state, self.search_factor, s.search_size)
        search = P.normalize(x_patch)
        out = O.forward_test(self.sd, s, self.template, search, self.ids, self.tmask, self.prompt, self.flag)
        self.last_out, self.last_resize, self.last_hw, self.prev_state = out, resize_factor, (H, W), list(self.state)
        new_state, score, net, idx = O.tracker_decode(out["cls_score_test"], out["cont_score"], out["bbox_map"], self.window,
                                                      np.asarray([self.state], f32), np.asarray([resize_factor], f32),
                                                      np.asarray([[H, W]], f32), s.search_size, has_cont=self.has_cont)
        self.state = [float(v) for v in new_state[0]]
        score = float(score[0])
        self.last_index, self.last_score = int(idx[0]), score
        if score > self.max_score and self.has_cont:
            self.pred_box_net, self.out_dict, self.max_score = net[0].copy(), out, score
            self.best_frame = self.frame_id
        self.updated = False
        if self.frame_id % self.update_interval == 0 and self.has_cont and self.max_score > self.threshold:
            b = self.pred_box_net
            context_bbox = np.array([[b[0] - 0.5 * b[2], b[1] - 0.5 * b[3], b[2], b[3]]], f32)
            context_mask = anno2mask(context_bbox, s.search_size // 16)
            self.prompt = O.forward_prompt(self.sd, s, self.out_dict, self.template_mask, context_mask)
            self.max_score = 0
            self.updated = True
        return {"target_bbox": self.state}

    def merged_scores(self):
        """cls * hann * softmax(cont)[0] of the last frame (for tie-aware comparisons)."""
        out = self.last_out
        cont = O.softmax(out["cont_score"][0], -1)[:, 0] if self.has_cont else 1.0
        return out["cls_score_test"][0].reshape(-1) * self.window * cont

    def decode_at(self, index):
        """The box the tracker would output for the last frame if its argmax were `index`."""
        out = self.last_out
        net = out["bbox_map"][0].reshape(-1, 4)[index].astype(np.float64) * self.spec.search_size / self.last_resize
        sx, sy, sw, sh = self.prev_state
        half = 0.5 * self.spec.search_size / self.last_resize
        box = [net[0] + (sx + 0.5 * sw - half) - 0.5 * net[2], net[1] + (sy + 0.5 * sh - half) - 0.5 * net[3], net[2], net[3]]
        return O.clip_box(box, self.last_hw[0], self.last_hw[1], margin=10)
