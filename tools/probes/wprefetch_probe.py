"""Is a one-sequence GEMM faster when its weight was read by the PREVIOUS kernel (memory-side cache / L2 warm) than when it comes from HBM?
fc1-like GEMM 553 x 3072 x 768 (4.7 MB of weight), rotating over 40 weights (190 MB, more than fits beside the rest); between GEMMs a kernel that
(a) reads the NEXT weight, (b) reads an unrelated buffer of the same size.  Usage (GPU box): python tools/probes/wprefetch_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uvltrack_amd import _native
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K, R = 553, 3072, 768, 60
x = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16() for _ in range(R)]
other = [torch.randn(N, K, device="cuda").bfloat16() for _ in range(R)]
sink = torch.zeros(1, device="cuda")
def gemm(i): lib.uvl_linear(p(x), p(ws[i % R]), p(b), p(y), M, N, K, 1, 0, 0, None, st)
def run(mode, it=240):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for i in range(it):
        if mode == "next": sink.add_(ws[(i) % R].view(torch.int32)[::16, ::4].sum())        # touches every 128-byte line? no: a strided sample -- see below
        elif mode == "other": sink.add_(other[i % R].view(torch.int32)[::16, ::4].sum())
        ev[i][0].record(); gemm(i); ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[40:])
    return ts[len(ts) // 2]
def run_full(mode, it=240):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for i in range(it):
        if mode == "next": sink.add_(ws[i % R].float().sum())           # reads all of the weight the next GEMM uses
        elif mode == "other": sink.add_(other[i % R].float().sum())
        ev[i][0].record(); gemm(i); ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[40:])
    return ts[len(ts) // 2]
for rep in range(3):
    print("rep %d  GEMM (event pair, us): weight read by the kernel before %.2f   unrelated buffer read before %.2f   nothing before %.2f" % (rep, run_full("next"), run_full("other"), run_full("none")), flush=True)
