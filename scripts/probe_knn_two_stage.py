"""A/B of the kNN sweep: full-precision similarity matrix vs fp16 pre-filter + exact verification (120k x 2400 x 1024)."""
import sys, time
import torch
sys.path.insert(0, ".")
from lvc_amd import kernels as K
from lvc_amd import label_verification as LV

D = "cuda:0"
g = torch.Generator().manual_seed(0)
S, Dm, Q = 2400, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 120000
classes = torch.arange(80).repeat_interleave(30).to(D)
centers = torch.randn(80, Dm, generator=g)
shots = (centers[classes.cpu()] + 2.0 * torch.randn(S, Dm, generator=g) + 0.3).to(D)
qcls = torch.randint(0, 80, (Q,), generator=g)
q = (centers[qcls] + 2.5 * torch.randn(Q, Dm, generator=g) + 0.3).to(D)
qcls = qcls.to(D)
LV.QUERY_CHUNK = int(sys.argv[2]) if len(sys.argv) > 2 else LV.QUERY_CHUNK


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {}
for name, two, emit in (("single", False, False), ("matrix", True, False), ("emit", True, True)) * 2:
    LV.KNN_TWO_STAGE, LV.KNN_EMIT = two, emit
    res[name] = LV.knn_sweep(classes, shots, q, qcls, 10, True)
    print("%-7s %.3f ms" % (name, timeit(lambda: LV.knn_sweep(classes, shots, q, qcls, 10, True))))
print("rows differing emit/single", (res["emit"][0] != res["single"][0]).any(dim=1).sum().item(), "emit/matrix", (res["emit"][0] != res["matrix"][0]).any(dim=1).sum().item())
mu = K.colmean(shots)
sn, sh, _ = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1)
pc = K.pack_linear(sn)
N = min(Q, 120000)
qn, qh, den = K.rownorm_h(q[:N], mu=mu, eps=1e-8, mode=1)
print("rows", N)
print("rownorm   %.3f ms" % timeit(lambda: K.rownorm(q[:N], mu=mu, eps=1e-8, mode=1)))
print("rownorm_h %.3f ms" % timeit(lambda: K.rownorm_h(q[:N], mu=mu, eps=1e-8, mode=1, want_rows=False)))
t = timeit(lambda: K.gemm_f16(qh, sh))
print("gemm f16  %.3f ms  %.0f TF/s" % (t, 2.0 * N * S * Dm / t / 1e9))
ap = K.gemm_f16(qh, sh)
print("verify    %.3f ms" % timeit(lambda: K.knn_verify_topk_vote(ap, q[:N], sn, LV.VERIFY_MARGIN, classes, qcls[:N], 10, mu=mu, den=den)))
n1 = min(N, 32768)
t = timeit(lambda: K.linear(qn[:n1], pc))
print("gemm x3   %.3f ms per %d rows  %.0f TF/s fp32-equivalent" % (t, n1, 2.0 * n1 * S * Dm / t / 1e9))
sm = K.linear(qn[:n1], pc)
print("topk      %.3f ms per %d rows" % (timeit(lambda: K.knn_topk_vote(sm, S, classes, qcls[:n1], 10)), n1))
print("max |approx - x3|", (ap[:n1] - sm).abs().max().item())
nsub = 256
print("sub gemm  %.3f ms" % timeit(lambda: K.gemm_f16(qh, sh, n=nsub, ldb=(S // nsub) * Dm)))
sub = K.gemm_f16(qh, sh, n=nsub, ldb=(S // nsub) * Dm)
print("lb        %.3f ms" % timeit(lambda: K.knn_lower_bound(sub)))
lb = K.knn_lower_bound(sub)
print("emit gemm %.3f ms (incl. lists alloc + counts memset)" % timeit(lambda: K.gemm_f16_emit(qh, sh, lb, LV.VERIFY_MARGIN)))
lists, counts = K.gemm_f16_emit(qh, sh, lb, LV.VERIFY_MARGIN)
print("counts mean %.1f max %d" % (counts.float().mean().item(), counts.max().item()))
print("verify l  %.3f ms" % timeit(lambda: K.knn_verify_lists(lists, counts, q[:N], sn, LV.VERIFY_MARGIN, classes, qcls[:N], 10, mu=mu, den=den)))
