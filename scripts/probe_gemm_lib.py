"""The kNN pre-filter GEMM (120000 x 2400 x 1024, fp16 operands) on this repo's kernel against the vendor library through torch.matmul
(hipBLASLt / rocBLAS): a yardstick for gemm_h.hip, not a product path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K
d = torch.device("cuda:0")
torch.manual_seed(0)
Q, S, D = 120000, 2400, 1024
a = torch.randn(Q, D, device=d).half()
b = torch.randn(S, D, device=d).half()
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
fl = 2.0 * Q * S * D / 1e9
out16 = torch.empty(Q, S, device=d, dtype=torch.float16)
t = timeit(lambda: torch.matmul(a, b.t(), out=out16))
print("torch.matmul fp16 -> fp16   %.4f ms  %.0f TF/s" % (t, fl / t))
t = timeit(lambda: K.gemm_f16(a, b, q15=True))
print("gemm_h.hip   fp16 -> q15    %.4f ms  %.0f TF/s" % (t, fl / t))
t = timeit(lambda: K.gemm_f16(a, b, q15=False))
print("gemm_h.hip   fp16 -> fp32   %.4f ms  %.0f TF/s" % (t, fl / t))
