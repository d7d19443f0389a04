"""debug.gemm_swp 0 vs 1: the frame's outputs must be bit-identical (same MFMAs in the same order)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uvltrack_amd import weightgen as wg
from uvltrack_amd.engine import HipEngine
from bench import build_spec
for model, B in (("B", 1), ("B", 2), ("L", 1)):
    spec = build_spec(model, 256, None)
    eng = HipEngine(spec, torch.device("cuda:0"), max_batch=B)
    eng.load_state_dict(wg.make_state_dict(spec, 0, include_unused=False))
    inp = wg.make_inputs(spec, batch=B, seed=3, flags=[2] * B)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = (t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
    outs = {}
    for v in (0, 1):
        eng.debug_set("gemm_swp", v)
        o = eng.forward(*args)
        torch.cuda.synchronize()
        outs[v] = {k: x.clone() for k, x in o.items() if torch.is_tensor(x)}
    bad = [k for k in outs[0] if not torch.equal(outs[0][k], outs[1][k])]
    print(model, B, "identical" if not bad else "DIFFERENT: %s" % bad, flush=True)
    eng.close()
