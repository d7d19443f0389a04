"""Does a many-sequence frame run faster as TWO half-batches on two streams (one half's memory-bound epilogues / LayerNorms under the other's MFMA
loops) than as one batch?  Usage (GPU box): python tools/probes/two_halves.py [L|B] [total batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from uvltrack_amd import weightgen as wg
from uvltrack_amd.engine import HipEngine
from uvltrack_amd.spec import spec_b, spec_l
model = sys.argv[1] if len(sys.argv) > 1 else "L"
BT = int(sys.argv[2]) if len(sys.argv) > 2 else 8
spec = spec_l(template_size=256, search_size=384) if model == "L" else spec_b(template_size=256, search_size=256)
dev = torch.device("cuda:0")
sd = wg.make_state_dict(spec, seed=1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def mk(B, stream, nparts):
    eng = HipEngine(spec, dev, max_batch=B); eng.load_state_dict(sd)
    inp = wg.make_inputs(spec, batch=B, seed=5, flags=[2] * B)
    with torch.cuda.stream(stream):
        step = eng.make_eager_step(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
    return eng, step
def run(steps_fns, streams, n=40, warm=8):
    for _ in range(warm):
        for f, s in zip(steps_fns, streams):
            with torch.cuda.stream(s): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for f, s in zip(steps_fns, streams):
            with torch.cuda.stream(s): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
e8, f8 = mk(BT, s0, 1)
ea, fa = mk(BT // 2, s1, 2); eb, fb = mk(BT // 2, s2, 2)
for rep in range(3):
    one = run([f8], [s0]); two = run([fa, fb], [s1, s2])
    print("rep %d  one batch of %d: %.3f ms = %.1f frames/s   two halves on two streams: %.3f ms = %.1f frames/s" % (rep, BT, one * 1e3, BT / one, two * 1e3, BT / two), flush=True)
