"""Floor of the LayerNorm kernel (M rows x D) back-to-back on one stream."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M, D in [(553, 768), (40, 768), (4424, 768), (681, 1024), (6984, 1024), (17696, 768)]:
    x = torch.randn(M, D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
    yb = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    fn = lambda: lib.uvl_layernorm(p(x), p(g), p(b), C.c_float(1e-6), p(yb), None, M, D, st)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): fn()
    e.record(); torch.cuda.synchronize()
    print("layernorm M=%5d D=%4d: %.2f us per launch (back-to-back)" % (M, D, a.elapsed_time(e) * 10))
# empty-ish kernel floor: f32->bf16 of 1K elements
z = torch.randn(1024, device="cuda"); zb = torch.empty(1024, device="cuda", dtype=torch.bfloat16)
fn = lambda: lib.uvl_f32_to_bf16(p(z), p(zb), 1024, st)
for _ in range(5): fn()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200): fn()
e.record(); torch.cuda.synchronize()
print("tiny kernel floor: %.2f us per launch (back-to-back)" % (a.elapsed_time(e) * 5))
