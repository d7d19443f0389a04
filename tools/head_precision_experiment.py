#!/usr/bin/env python3
"""Where does the bf16 precision plan's box error come from -- backbone or head?  (Round-5 review, item 4: "if UVLTrack-B bbox_map can be brought under SURVEY 8c's
5e-3 cheaply, e.g. the tower layers in f32 / split-bf16, measure it with the emulating oracle first".)  CPU only, no HIP: the numpy oracle with its bf16-emulating mode
switched on for the backbone and / or the head separately, against the reference's outputs of a committed fixture.

    python tools/head_precision_experiment.py [fixture ...]      (default: b_z256_x256 l_z256_x384)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import uvl_oracle as O                                     # noqa: E402
from tests.golden_util import load_case, rebuild_inputs, rebuild_weights   # noqa: E402

f32 = np.float32


def run(sd, spec, inp, emu_backbone, emu_head):
    sd = {k: (np.asarray(v, dtype=f32) if np.asarray(v).dtype.kind == "f" else np.asarray(v)) for k, v in sd.items()}
    with O.emulate_bf16(on=emu_backbone):
        out = O.backbone_forward(sd, spec, inp["template"].astype(f32), inp["search"].astype(f32), np.asarray(inp["ids"]), np.asarray(inp["mask"]),
                                 np.asarray(inp["flag"]).reshape(-1, 1), None)
    with O.emulate_bf16(on=emu_head):
        return O.head_forward(sd, spec, out, inp["prompt"].astype(f32))


def main(names):
    print("| fixture | backbone | head | bbox_map - ref | cls_score_test - ref | cont_score - ref |")
    print("|---|---|---|---|---|---|")
    for name in names:
        meta, spec, ref = load_case(name)
        inp = rebuild_inputs(meta, spec)
        sd = rebuild_weights(meta, spec, include_unused=False)
        for eb, eh in ((True, True), (True, False), (False, True), (False, False)):
            o = run(sd, spec, inp, eb, eh)
            e = {k: float(np.abs(o[k] - ref[k]).max()) for k in ("bbox_map", "cls_score_test", "cont_score")}
            print("| %s | %s | %s | %.2e | %.2e | %.2e |" % (name, "bf16 operands" if eb else "fp32", "bf16 operands" if eh else "fp32", e["bbox_map"], e["cls_score_test"], e["cont_score"]), flush=True)


def head_variants(names):
    """bf16 backbone throughout; which tower operands / layers carry the head's share of the error."""
    print("| fixture | tower operands rounded to bf16 | bbox_map - ref | cls_score_test - ref |")
    print("|---|---|---|---|")
    for name in names:
        meta, spec, ref = load_case(name)
        inp = rebuild_inputs(meta, spec)
        sd = rebuild_weights(meta, spec, include_unused=False)
        for label, cfg in (("activations + weights, all layers (the plan)", dict(act=True, w=True, from_layer=0)),
                           ("weights only (activations f32)", dict(act=False, w=True, from_layer=0)),
                           ("activations only (weights f32)", dict(act=True, w=False, from_layer=0)),
                           ("layers 1-3 only (layer 0 f32)", dict(act=True, w=True, from_layer=1)),
                           ("layers 2-3 only", dict(act=True, w=True, from_layer=2)),
                           ("none (head f32)", dict(act=False, w=False, from_layer=0))):
            O.HEAD_ROUND.update(cfg)
            try:
                o = run(sd, spec, inp, True, True)
            finally:
                O.HEAD_ROUND.update(dict(act=True, w=True, from_layer=0))
            print("| %s | %s | %.2e | %.2e |" % (name, label, float(np.abs(o["bbox_map"] - ref["bbox_map"]).max()), float(np.abs(o["cls_score_test"] - ref["cls_score_test"]).max())), flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--head" in sys.argv:
        head_variants(args or ["b_z256_x256"])
    else:
        main(args or ["b_z256_x256", "l_z256_x384"])
