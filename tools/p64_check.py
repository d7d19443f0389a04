"""Development aid: attn_p64_kernel (attn_cfg 11) against torch on a ladder of shapes, each case in its own process under a timeout
(a hand-written barrier schedule that deadlocks must not take the box with it).   python tools/p64_check.py [cfg]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(1, 2, 128, "none"), (1, 2, 256, "none"), (2, 3, 321, "none"), (1, 2, 257, "fill"), (2, 4, 553, "fill"), (3, 2, 321, "bert_all"),
         (2, 3, 65, "bert"), (8, 16, 681, "fill"), (8, 16, 873, "bert"), (1, 2, 1100, "fill"), (32, 12, 553, "none"), (24, 16, 873, "fill")]

if len(sys.argv) > 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_kernels_gpu as T
    from uvltrack_amd import _native
    lib = _native.load()
    B, H, N, mode, cfg = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6])
    T._attention_case(lib, B, H, N, mode, 300 + N, tune=T._tune(attn_cfg=cfg))
    print("ok")
    sys.exit(0)

cfg = sys.argv[1] if len(sys.argv) > 1 else "11"
for (B, H, N, mode) in CASES:
    try:
        r = subprocess.run([sys.executable, __file__, "--one", str(B), str(H), str(N), mode, cfg], capture_output=True, text=True, timeout=90)
        tail = (r.stdout + r.stderr).strip().splitlines()[-1:] or [""]
        print("B=%d H=%d N=%d %s: rc %d  %s" % (B, H, N, mode, r.returncode, tail[0][:200]), flush=True)
    except subprocess.TimeoutExpired:
        print("B=%d H=%d N=%d %s: TIMEOUT (hang) -- stopping" % (B, H, N, mode), flush=True)
        break
