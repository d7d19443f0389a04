"""Host-side cost of enqueuing one eager frame (no GPU wait) vs the GPU time of the frame."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import weightgen as wg
from uvltrack_amd.engine import HipEngine
from uvltrack_amd.spec import spec_b

spec = spec_b(256, 256)
eng = HipEngine(spec, torch.device("cuda:0"), max_batch=1)
eng.load_state_dict(wg.make_state_dict(spec, 0, include_unused=False))
inp = wg.make_inputs(spec, batch=1, seed=1, flags=[2])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = (t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
for skip, mode in ((False, "NLBBOX two streams"), (True, "skip-text single stream")):
    a = list(args)
    if skip:
        a[5] = torch.zeros_like(a[5])
    step = eng.make_eager_step(*a, skip_text=skip)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    host = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 200
    print("%-26s host enqueue of one frame (GPU idle): %.3f ms (min %.3f)   steady-state frame: %.3f ms" % (mode, np.median(host) * 1e3, min(host) * 1e3, tot * 1e3))
