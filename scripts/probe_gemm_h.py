"""fp16 pre-filter GEMM alone (120000 x 2400 x 1024): time and TF/s; LVC_GH_ABLATE for the ablations."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvc_amd import kernels as K
D = "cuda:0"
M, N, C = 120000, 2400, 1024
a = (torch.randn(M, C, device=D) / 32).half()
b = (torch.randn(N, C, device=D) / 32).half()
for _ in range(3):
    K.gemm_f16(a, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    K.gemm_f16(a, b)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
print("ablate=%s  %.3f ms  %.0f TF/s" % (os.environ.get("LVC_GH_ABLATE", "0"), t, 2.0 * M * N * C / t / 1e9))
