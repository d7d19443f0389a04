"""A/B of the split-K factors of the residual GEMMs (proj: K = D, fc2: K = 4 D) in the one-sequence frame, same box, interleaved.
Usage (GPU box): python tools/ab_splitk.py"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (before the library: one HIP runtime in the process)
import bench  # noqa: E402


def run(k1, k4, extra=()):
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-batched", "--steps", "300", "--warmup", "50", "--blocks", "5",
                "--tune", "sk_k1=%d" % k1, "--tune", "sk_k4=%d" % k4, *extra]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    return d["value"], d["config"]["launches_per_frame"]


if __name__ == "__main__":
    for rep in range(2):
        for k1, k4 in ((-1, -1), (1, -1), (-1, 2), (1, 2), (2, 4), (-1, 1)):
            fps, n = run(k1, k4)
            print("rep %d  sk(K=D) %2d  sk(K=4D) %2d   %7.1f frames/s  %d launches" % (rep, k1, k4, fps, n), flush=True)
