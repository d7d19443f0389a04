"""CPU experiment (round 5, the review's item 2): would a bf16 RESIDUAL STREAM stay inside SURVEY 8c's 5e-3 on bbox_map / cls?
The numpy oracle's bf16-emulating mode (oracle/uvl_oracle.py::emulate_bf16: the HIP path's roundings, f32 residual stream) is run as it is and with the
residual stream rounded to bf16 after every residual add (what a bf16 read-modify-write epilogue + a bf16-reading LayerNorm would do), both against the
reference outputs in the committed fixtures.  No HIP code involved.  Usage: python tools/bf16_residual_experiment.py [fixture ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_case, rebuild_inputs, rebuild_weights  # noqa: E402
from oracle import uvl_oracle as O  # noqa: E402


def run(name):
    meta, spec, ref = load_case(name)
    sd = rebuild_weights(meta, spec)
    inp = rebuild_inputs(meta, spec)
    args = (sd, spec, inp["template"], inp["search"], inp["ids"], inp["mask"], inp["prompt"], inp["flag"])
    out = {}
    t0 = time.time()
    out["f32 residual"] = O.forward_test(*args, emulate_bf16_mode=True)
    vit_block, bert_layer = O.vit_block, O.bert_layer

    def vit_block_bf16(sd_, i, x, key_mask, heads):
        # x = bf16(x + attn(LN(x))); x = bf16(x + mlp(LN(x))) -- the stream itself is stored in bf16 between the kernels
        p = "backbone.vit.blocks.%d." % i
        x = O.bf16_round(x)
        x = O.bf16_round(x + O.vit_attention(sd_, p + "attn.", O.layer_norm(x, sd_[p + "norm1.weight"], sd_[p + "norm1.bias"], 1e-6), key_mask, heads))
        h = O.layer_norm(x, sd_[p + "norm2.weight"], sd_[p + "norm2.bias"], 1e-6)
        h = O.gelu(O.linear(h, sd_[p + "mlp.fc1.weight"], sd_[p + "mlp.fc1.bias"]))
        return O.bf16_round(x + O.linear(h, sd_[p + "mlp.fc2.weight"], sd_[p + "mlp.fc2.bias"])).astype(np.float32)
    O.vit_block = vit_block_bf16
    try:
        out["bf16 residual"] = O.forward_test(*args, emulate_bf16_mode=True)
    finally:
        O.vit_block = vit_block
    row = []
    for k in ("bbox_map", "cls_score_test", "cont_score", "logits"):
        row.append("%s %s" % (k, " / ".join("%.2e" % float(np.abs(out[m][k] - ref[k]).max()) for m in ("f32 residual", "bf16 residual"))))
    print("%-22s (%4.0f s)  max |error| vs the reference, f32 / bf16 residual stream:  %s" % (name, time.time() - t0, "   ".join(row)), flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or ["tiny_mixed", "tiny_switches", "tiny_allmasked_text", "b_z128_x256", "b_z256_x256", "l_z128_x384", "l_z256_x384"]
    for n in names:
        run(n)
