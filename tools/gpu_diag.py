"""First-contact diagnostics on the GPU box: per-output and per-layer errors of the HIP forward vs the oracle.
Writes gpurun_out/diag.json.  Usage: python tools/gpu_diag.py [case ...]"""
import ctypes as C
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import uvl_oracle as O                       # noqa: E402
from tests.golden_util import load_case, rebuild_inputs, rebuild_weights, list_cases   # noqa: E402
from tests.parity_util import compare_outputs, fmt_report                             # noqa: E402
from uvltrack_amd.engine import HipEngine                # noqa: E402


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def main(names):
    res = {}
    for name in names:
        try:
            meta, spec, ref = load_case(name)
            inp = rebuild_inputs(meta, spec)
            sd = rebuild_weights(meta, spec)
            eng = HipEngine(spec, torch.device("cuda:0"), max_batch=8)
            t0 = time.time()
            eng.load_state_dict(sd)
            print("[%s] weights loaded in %.1fs" % (name, time.time() - t0))
            args = (t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
            out = eng.forward(*args)
            torch.cuda.synchronize()
            got = {k: v.cpu().numpy() for k, v in out.items() if torch.is_tensor(v)}
            ok, rep = compare_outputs(got, ref, depth=spec.depth)
            print("[%s] %s\n%s" % (name, "PASS" if ok else "FAIL", fmt_report(rep)))
            res[name] = {"ok": bool(ok), "report": {k: list(v) if isinstance(v, tuple) else v for k, v in rep.items()}}
            if name.startswith("tiny") or not ok:
                # per-layer localisation against oracle taps
                taps = {}
                O.forward_test(sd, spec, inp["template"], inp["search"], inp["ids"], inp["mask"], inp["prompt"], inp["flag"], taps)
                layers = {}
                for i in range(spec.depth):
                    eng.lib.uvl_debug_set(eng.handle, b"stop_layer", i)
                    o = eng.forward(*args)
                    torch.cuda.synchronize()
                    img = np.concatenate([o["vis_token"].cpu().numpy(), o["template"].cpu().numpy(), o["search"].cpu().numpy()], axis=1)
                    txt = o["text"].cpu().numpy()
                    ei = float(np.abs(img - taps["img_%d" % i]).max())
                    et = float(np.abs(txt - taps["txt_%d" % i]).max())
                    layers[i] = (ei, float(np.abs(taps["img_%d" % i]).max()), et, float(np.abs(taps["txt_%d" % i]).max()))
                    print("   layer %2d  img err %.3e (absmax %.2f)   txt err %.3e (absmax %.2f)" % ((i,) + layers[i]))
                eng.lib.uvl_debug_set(eng.handle, b"stop_layer", -1)
                res[name]["layers"] = layers
            # timing: eager + graph
            for _ in range(5):
                eng.forward(*args)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(20):
                eng.forward(*args)
            torch.cuda.synchronize()
            eager_ms = (time.time() - t0) / 20 * 1e3
            eng.capture(*args)
            for _ in range(5):
                eng.replay()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(50):
                eng.replay()
            torch.cuda.synchronize()
            graph_ms = (time.time() - t0) / 50 * 1e3
            print("[%s] batch %d: eager %.3f ms  graph %.3f ms" % (name, meta["batch"], eager_ms, graph_ms))
            res[name]["eager_ms"], res[name]["graph_ms"] = eager_ms, graph_ms
            eng.forward(*args, profile=True)
            prof = sorted(eng.profile_entries(), key=lambda e: -e["ms"])
            for e in prof[:12]:
                print("   %-16s %-34s %3d launches %8.3f ms  %7.1f TF/s" % (e["site"], e["kernel"], e["launches"], e["ms"],
                                                                          e["flops"] / max(e["ms"], 1e-9) / 1e9))
            res[name]["profile"] = prof
            eng.close()
        except Exception as ex:       # keep going: one GPU round trip must report everything it can
            traceback.print_exc()
            res[name] = {"ok": False, "exception": repr(ex)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w") as f:
        json.dump(res, f, indent=1, default=str)


if __name__ == "__main__":
    main(sys.argv[1:] or list_cases())
