import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from uvltrack_amd import _native
lib = _native.load()
T = _native.UvlTuning(); T.gemm_cfg = 34
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(x, w, M, N, K):
    y = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(N, device="cuda")
    rc = lib.uvl_linear(p(x), p(w), p(b), p(y), M, N, K, 0, 0, 0, T.ref(), st)
    torch.cuda.synchronize()
    return rc, y.float()
M, N, K = 256, 256, 256
x = torch.ones(M, K, device="cuda").bfloat16(); w = torch.ones(N, K, device="cuda").bfloat16()
rc, y = run(x, w, M, N, K); print("ones rc", rc, "unique", torch.unique(y)[:8].tolist())
# row pattern: x[r, :] = r % 8 + 1 ; w ones -> y[r, n] = K * (r%8+1)
x = ((torch.arange(M, device="cuda") % 8 + 1).float()[:, None].expand(M, K)).contiguous().bfloat16()
rc, y = run(x, w, M, N, K); ref = x.float() @ w.float().t(); print("rowpat maxerr", (y - ref).abs().max().item(), "bad rows", ((y - ref).abs().amax(1) > 0).nonzero().flatten()[:16].tolist())
# k pattern: x[r, k] = 1 if k == k0 else 0, w[n, k] = k  -> y = k0
for k0 in (0, 7, 8, 15, 16, 31, 32, 63, 64, 100, 255):
    x = torch.zeros(M, K, device="cuda"); x[:, k0] = 1; x = x.bfloat16()
    w = torch.arange(K, device="cuda").float()[None, :].expand(N, K).contiguous().bfloat16()
    rc, y = run(x, w, M, N, K); print("k0", k0, "got", torch.unique(y)[:6].tolist())
x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
rc, y = run(x, w, M, N, K); ref = x.float() @ w.float().t(); print("random maxerr", (y - ref).abs().max().item())
