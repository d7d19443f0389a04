"""Helpers shared by CPU and GPU parity tests: load a golden fixture and rebuild its inputs/weights."""
import json
import os

import numpy as np

from uvltrack_amd import weightgen as wg
from uvltrack_amd.spec import ModelSpec

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def list_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    spec = ModelSpec(**meta["spec"])
    ref = {k[4:]: z[k] for k in z.files if k.startswith("ref.")}
    return meta, spec, ref


def rebuild_inputs(meta, spec):
    inp = wg.make_inputs(spec, batch=meta["batch"], seed=meta["input_seed"], flags=meta["flags"])
    if meta.get("zero_text"):
        inp["mask"][1:, :] = False
    for r in meta.get("zero_text_rows", ()):
        inp["mask"][r, :] = False
    for k, v in meta["input_checksums"].items():
        got = float(np.asarray(inp[k], dtype=np.float64).sum())
        assert abs(got - v) <= 1e-6 * max(1.0, abs(v)), "input %s does not regenerate bit-identically" % k
    return inp


def rebuild_weights(meta, spec, include_unused=False):
    sd = wg.make_state_dict(spec, meta["weight_seed"], include_unused=include_unused)
    for k, v in meta["weight_checksums"].items():
        got = float(np.asarray(sd[k], dtype=np.float64).sum())
        assert abs(got - v) <= 1e-9 * max(1.0, abs(v)), "weight %s does not regenerate bit-identically" % k
    return sd
