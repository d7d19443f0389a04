"""Interleaved A/B of one uvl_tune_set key (bench.py --tune key=value: the override lives in that run's model handle) in a bench.py
workload, same box, same process.
Usage (GPU box): python tools/ab_tune.py <key> <value A> <value B> [bench.py flags ...]
e.g.  python tools/ab_tune.py gemm_big 0 1 --model L --batch 8 --template-size 256 --search-size 384 --steps 30 --warmup 5"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (before the library: one HIP runtime in the process)
import bench  # noqa: E402


def run(key, val, extra):
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-batched", "--tune", "%s=%d" % (key, val), *extra]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    return d["value"], d["ms_per_step"]


if __name__ == "__main__":
    key, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    extra = sys.argv[4:]
    for rep in range(3):
        for v in (a, b):
            fps, ms = run(key, v, extra)
            print("rep %d  %s = %d   %8.1f frames/s  %.3f ms/step" % (rep, key, v, fps, ms), flush=True)
