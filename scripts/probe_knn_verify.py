"""What the kNN verification launch spends its time on (cfg 4, randn): the scan alone (margin 0 and one class: nothing flagged),
the query-row load alone, and the real thing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K, label_verification as LV
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Q, S, D = 120000, 2400, 1024
classes = torch.arange(80).repeat_interleave(30).to(dev)
one_class = torch.zeros(S, dtype=torch.int64, device=dev)
shots, q = torch.randn(S, D, generator=g).to(dev), torch.randn(Q, D, generator=g).to(dev)
det = torch.randint(0, 80, (Q,), generator=g).to(dev)
mu = K.colmean(shots)
sn, sh, _, sres = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1, want_resid=True)
sres_max = K.max_f32(sres)
_, qh, den, qres = K.rownorm_h(q, mu=mu, eps=1e-8, mode=1, want_rows=False, want_resid=True)
margins = LV.pre_filter_margins(qres, 1.0 + 1e-6 + sres_max, sres_max, D) + LV.Q15_MARGIN
ap = K.gemm_f16(qh, sh, q15=True)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print("margins: mean %.2e max %.2e" % (float(margins.mean()), float(margins.max())))
t_real = timed(lambda: K.knn_verify_topk_vote(ap, q, sn, LV.VERIFY_MARGIN, classes, det, 10, mu=mu, den=den, margins=margins))
t_one = timed(lambda: K.knn_verify_topk_vote(ap, q, sn, LV.VERIFY_MARGIN, one_class, det, 10, mu=mu, den=den, margins=margins))
zero = torch.zeros_like(margins)
t_zero = timed(lambda: K.knn_verify_topk_vote(ap, q, sn, 0.0, classes, det, 10, mu=mu, den=den, margins=zero))
big = margins * 4
t_big = timed(lambda: K.knn_verify_topk_vote(ap, q, sn, LV.VERIFY_MARGIN, classes, det, 10, mu=mu, den=den, margins=big))
print("verify: real margins %.3f ms | one class (nothing flagged: scan + ranking only) %.3f ms | zero margins %.3f ms | 4x margins %.3f ms" %
      (t_real, t_one, t_zero, t_big))
