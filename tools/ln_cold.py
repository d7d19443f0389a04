"""LayerNorm of the UVLTrack-L x 8 frame (6984 x 1024 f32 -> bf16): the same buffers every launch against rotating buffers
(12 x 43 MB = 515 MB, more than the 256-MB memory-side cache), and against rows another kernel has just written (a residual-style
x += 1 pass in front of every LayerNorm, the way the frame's GEMM epilogue leaves x)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native
if "--lib" in sys.argv:
    i = sys.argv.index("--lib"); _native.LIB_PATH = os.path.abspath(sys.argv[i + 1]); del sys.argv[i:i + 2]
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, D, R = 6984, 1024, 12
xs = [torch.randn(M, D, device="cuda") for _ in range(R)]
ys = [torch.empty(M, D, device="cuda", dtype=torch.bfloat16) for _ in range(R)]
g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")


def run(rot, touch, iters=48):
    def step(i):
        j = i % R if rot else 0
        if touch:
            xs[j].add_(1.0)
        lib.uvl_layernorm(p(xs[j]), p(g), p(b), C.c_float(1e-6), p(ys[j]), None, M, D, st)
    for i in range(R):
        step(i)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        step(i)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3


for name, rot, touch in (("same buffers", 0, 0), ("rotating buffers", 1, 0), ("same buffers", 0, 0)):
    print("layernorm 6984 x 1024, %-40s %.1f us per launch" % (name, min(run(rot, touch) for _ in range(3))))
t_add = None
for name, rot in (("x += 1 alone, same buffer", 0), ("x += 1 alone, rotating", 1)):
    def only(i, rot=rot):
        xs[i % R if rot else 0].add_(1.0)
    for i in range(R): only(i)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(48): only(i)
    e.record(); torch.cuda.synchronize()
    t = a.elapsed_time(e) / 48 * 1e3
    print("%-56s %.1f us per launch" % (name, t))
    print("layernorm behind it (%s): %.1f us for the pair -> %.1f us for the LayerNorm" % ("rotating" if rot else "same", (tt := min(run(rot, 1) for _ in range(3))), tt - t))
