import ctypes as C, sys, torch
sys.path.insert(0, '/root/repo')
from uvltrack_amd import _native
lib = _native.load()
TUNE = _native.UvlTuning()      # per-call overrides of the launch heuristics (no process-global tuning state)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for M in (553, 4424, 17696):
    N, K = 3072, 768
    x = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(N, K, device='cuda') / K ** 0.5).bfloat16()
    bias = torch.randn(N, device='cuda'); y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for rep in range(2):
        t0 = timeit(lambda: lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, 0, 0, 0, TUNE.ref(), st))
        t1 = timeit(lambda: lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, 1, 0, 0, TUNE.ref(), st))
        t2 = timeit(lambda: lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, 2, 0, 0, TUNE.ref(), st))
    print("fc1 M=%5d: no act %.1f us | GELU %.1f us | ReLU %.1f us" % (M, t0, t1, t2))
