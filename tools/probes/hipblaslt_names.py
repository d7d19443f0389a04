"""Which hipBLASLt kernels torch picks on the frames' GEMM shapes (run under rocprofv3 --kernel-trace --stats).
Measurement aid: the product never calls torch math."""
import torch
import torch.nn.functional as F

for M, N, K in ((21792, 1024, 4096), (21792, 4096, 1024), (6984, 4096, 1024), (6984, 1024, 4096), (6984, 3072, 1024), (16384, 4096, 4096)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    for _ in range(20):
        F.linear(x, w, b)
    torch.cuda.synchronize()
