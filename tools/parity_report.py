#!/usr/bin/env python3
"""The stated tolerance as MEASURED numbers (round-5 review, item 4): per committed reference fixture and box output, on this GPU,

    HIP - reference | bf16-emulating oracle - reference | HIP - emulation | gate of tests/parity_util.py

for (a) the default path at the fixture's own batch size and (b) the default ONE-SEQUENCE frame, sample by sample (round 6: the LayerNorm-free schedule,
against the oracle's fold-emulating mode), plus the predicted-box IoU line.  The oracle runs on the CPU of the GPU box and is only the checker here.

    python tools/parity_report.py [--out gpurun_out/r06_parity.md] [--cases a,b,...] [--skip-emulation-above ROWS]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KEYS = ("bbox_map", "cls_score_test", "cont_score", "logits")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_parity.md"))
    ap.add_argument("--cases", default="")
    ap.add_argument("--skip-emulation-above", type=int, default=6000, help="token rows (batch x joint tokens) above which the numpy emulation is not run")
    args = ap.parse_args()
    import torch
    from oracle import uvl_oracle as O
    from tests.golden_util import list_cases, load_case, rebuild_inputs, rebuild_weights
    from tests.parity_util import ATOL, pred_box_iou
    from uvltrack_amd.engine import HipEngine

    names = [c for c in (args.cases.split(",") if args.cases else list_cases()) if c]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    lines = ["# Parity of the HIP path against the reference's own outputs, measured (`tools/parity_report.py`, MI355X)", "",
             "Absolute maximum deviations.  *reference* = the committed fixture (outputs of the imported reference, fp32 CPU: `oracle/make_golden.py`); *emulation* = the numpy",
             "oracle with this precision plan's bf16 roundings and no HIP code (`emulate_bf16_mode=True`; for one-sequence frames `\"fold\"`: the LayerNorm-free frame's rounding",
             "points) -- its deviation from the reference IS the quantisation noise of the plan on these weights.  *gate* = `tests/parity_util.py` (x depth / 12).", ""]
    rows_a = ["| fixture (batch) | output | HIP - ref | emulation - ref | HIP - emulation | gate | pred-box IoU HIP / emulation |", "|---|---|---|---|---|---|---|"]
    rows_b = ["| fixture, sample | output | HIP - ref | fold emulation - ref | HIP - emulation | gate |", "|---|---|---|---|---|---|"]
    for name in names:
        meta, spec, ref = load_case(name)
        inp = rebuild_inputs(meta, spec)
        sd = rebuild_weights(meta, spec, include_unused=True)
        B = meta["batch"]
        scale = max(1.0, spec.depth / 12.0)
        eng = HipEngine(spec, torch.device("cuda:0"), max_batch=max(8, B))
        eng.load_state_dict(sd)
        run = lambda d: {k: v.cpu().numpy() for k, v in eng.forward(t(d["template"]), t(d["search"]), t(d["ids"]), t(d["mask"]), t(d["prompt"]), t(d["flag"])).items() if torch.is_tensor(v)}
        got = run(inp)
        torch.cuda.synchronize()
        rows = B * spec.nj
        emu = emu_fold = None
        t0 = time.time()
        a = (sd, spec, inp["template"], inp["search"], inp["ids"], inp["mask"], inp["prompt"], inp["flag"])
        if rows <= args.skip_emulation_above:
            emu = O.forward_test(*a, None, emulate_bf16_mode=True)
            if spec.txt_token_mode != "mean":
                emu_fold = O.forward_test(*a, None, emulate_bf16_mode="fold")
        print("%s: HIP + emulation in %.0f s" % (name, time.time() - t0), flush=True)
        for k in KEYS:
            e_h = float(np.abs(got[k] - ref[k]).max())
            e_e = "%.2e" % float(np.abs(emu[k] - ref[k]).max()) if emu is not None else "-"
            e_he = "%.2e" % float(np.abs(got[k] - emu[k]).max()) if emu is not None else "-"
            iou = ""
            if k == "bbox_map":
                iou = "%.4f / %s" % (pred_box_iou(got, ref), "%.4f" % pred_box_iou(emu, ref) if emu is not None else "-")
            rows_a.append("| %s (%d) | %s | %.2e | %s | %s | %.0e | %s |" % (name, B, k, e_h, e_e, e_he, ATOL[k] * scale, iou))
        if emu_fold is not None and B <= 3:
            for b in range(B):
                one = run({k: v[b:b + 1] for k, v in inp.items()})
                for k in KEYS:
                    rows_b.append("| %s, %d | %s | %.2e | %.2e | %.2e | %.0e |" % (
                        name, b, k, float(np.abs(one[k] - ref[k][b:b + 1]).max()), float(np.abs(emu_fold[k][b:b + 1] - ref[k][b:b + 1]).max()),
                        float(np.abs(one[k] - emu_fold[k][b:b + 1]).max()), ATOL[k] * scale))
        eng.close()
        del eng
    lines += ["## (a) default path at the fixture's batch size", ""] + rows_a + ["", "## (b) default one-sequence frame (LayerNorm-free schedule), sample by sample", ""] + rows_b + [""]
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
