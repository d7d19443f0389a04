"""What does a second queue cost the visual stream?  The UVLTrack-L x 8 frame without its text branch (skip_text), alone and beside N
tiny launches per frame on another stream (N = 126 = the text branch's launch count).  If the tiny launches cost what the real branch
costs, the price of the branch is its kernel BOUNDARIES (cache write-back / invalidate per completed kernel), not its CU time.
Usage (GPU box): python tools/text_interference.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import weightgen as wg  # noqa: E402
from uvltrack_amd.engine import HipEngine  # noqa: E402
from uvltrack_amd.spec import spec_l  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    spec = spec_l(256, 384)
    B = 8
    eng = HipEngine(spec, dev, max_batch=B)
    eng.load_state_dict(wg.make_state_dict(spec, 0, include_unused=False))
    inp = wg.make_inputs(spec, batch=B, seed=0, flags=[0] * B)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    targs = (t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
    outs = eng.alloc_outputs(B)
    step = eng.make_eager_step(*targs, skip_text=True, outs=outs)
    side = torch.cuda.Stream()
    tiny = torch.zeros(64, device=dev)
    big = torch.zeros(320 * 1024, device=dev)

    def run(n_side, tensor, steps=60):
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            if n_side:
                with torch.cuda.stream(side):
                    for _ in range(n_side):
                        tensor.add_(1.0)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    for rep in range(2):
        print("frame alone                          %.3f ms" % run(0, tiny))
        print("+ 126 tiny launches on a second queue %.3f ms" % run(126, tiny))
        print("+ 126 x 1.3 MB elementwise launches   %.3f ms" % run(126, big))
        print("+ 30 tiny launches                    %.3f ms" % run(30, tiny))


if __name__ == "__main__":
    main()
