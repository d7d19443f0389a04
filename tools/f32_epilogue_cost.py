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
for M in (4424, 17696):
    for name, N, K in (("proj", 768, 768), ("fc2", 768, 3072)):
        x = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(N, K, device='cuda') / K ** 0.5).bfloat16()
        bias = torch.randn(N, device='cuda')
        yb = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); yf = torch.zeros(M, N, device='cuda')
        for rep in range(2):
            t0 = timeit(lambda: lib.uvl_linear(p(x), p(w), p(bias), p(yb), M, N, K, 0, 0, 0, TUNE.ref(), st))
            t1 = timeit(lambda: lib.uvl_linear(p(x), p(w), p(bias), p(yf), M, N, K, 0, 1, 0, TUNE.ref(), st))
            t2 = timeit(lambda: lib.uvl_linear(p(x), p(w), p(bias), p(yf), M, N, K, 0, 1, 1, TUNE.ref(), st))
        print("%s M=%d: bf16 out %.1f us | f32 out %.1f us | f32 accumulate %.1f us" % (name, M, t0, t1, t2))
