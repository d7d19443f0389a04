"""Would two half-batches on two streams beat one batch?  Zero-code stand-in: two engines of B/2 sequences each (own weights, own graph) replayed concurrently on
two streams, against one engine of B sequences.  UVLTrack-L z256/x384 (configs[4]) unless --model B."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from uvltrack_amd import weightgen as wg
from uvltrack_amd.engine import HipEngine

def make(spec, dev, B, seed):
    eng = HipEngine(spec, dev, max_batch=B)
    eng.load_state_dict(wg.make_state_dict(spec, 0, include_unused=False))
    inp = wg.make_inputs(spec, batch=B, seed=seed, flags=[2] * B)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    targs = (t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
    eng.capture(*targs)
    return eng

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    spec = bench.build_spec("L", 256, 384)
    full = make(spec, dev, B, 1)
    halves = [make(spec, dev, B // 2, 2), make(spec, dev, B - B // 2, 3)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def run_full(n):
        for _ in range(n): full.replay()
    def run_halves(n):
        for _ in range(n):
            for e, s in zip(halves, streams):
                with torch.cuda.stream(s): e.replay()
    for rep in range(3):
        for name, fn in (("one batch of %d" % B, run_full), ("two halves on two streams", run_halves)):
            fn(5); torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(30); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
            print("%-28s %.3f ms/step  %.1f frames/s" % (name, dt * 1e3, B / dt), flush=True)
main()
