"""One compute-heavy pointwise GEMM on the f16x2 256-row kernel (the kNN chunk: 32768 x 1024 -> 2400) for --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
M, C, N = 32768, 1024, 2400
if len(sys.argv) > 3:
    M, C, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(M, C, device=d)
w = torch.randn(N, C, device=d) * 0.03
pc = k.pack_linear(w)
for _ in range(6):
    y = k.linear(x, pc)
torch.cuda.synchronize()
