"""The kNN pre-filter GEMM (120000 x 2400 x 1024, 16-bit output) timed alone in a loop and inside the sweep's sequence (row
normalisation of the queries in front of every launch): does what ran before it change its time?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Q, S, D = 120000, 2400, 1024
shots, q = torch.randn(S, D, generator=g).to(dev), torch.randn(Q, D, generator=g).to(dev)
mu = K.colmean(shots)
sn, sh, _ = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1)
_, qh, den = K.rownorm_h(q, mu=mu, eps=1e-8, mode=1, want_rows=False)


def gemm_ms(before, n=10):
    tot = 0.0
    for _ in range(3):
        before(); K.gemm_f16(qh, sh, q15=True)
    for _ in range(n):
        before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.gemm_f16(qh, sh, q15=True)
        e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n


big = torch.empty(256 << 20, dtype=torch.float32, device=dev)      # 1 GiB
for rnd in range(2):
    print("round %d: GEMM alone (synchronised before every launch) %.3f ms | after rownorm_h of the queries %.3f ms | after a 1 GiB fill (caches flushed) %.3f ms"
          % (rnd, gemm_ms(lambda: None), gemm_ms(lambda: K.rownorm_h(q, mu=mu, eps=1e-8, mode=1, want_rows=False)), gemm_ms(lambda: big.fill_(1.0))))
