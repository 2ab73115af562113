"""The two-stage kNN sweep with the pre-filter similarities in fp32 against 16-bit fixed point (label_verification.KNN_Q15): time per
sweep and per kernel, identical outputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K, label_verification as LV
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Q, S, D = 120000, 2400, 1024
classes = torch.arange(80).repeat_interleave(30).to(dev)
for kind in ("randn", "structured"):
    if kind == "randn":
        shots, q = torch.randn(S, D, generator=g).to(dev), torch.randn(Q, D, generator=g).to(dev)
    else:
        centers = torch.randn(80, D, generator=g)
        shots = (centers[classes.cpu()] + 2.0 * torch.randn(S, D, generator=g)).to(dev)
        q = (centers[torch.randint(0, 80, (Q,), generator=g)] + 2.5 * torch.randn(Q, D, generator=g)).to(dev)
    det = torch.randint(0, 80, (Q,), generator=g).to(dev)
    res = {}
    for q15 in (False, True, False, True):
        LV.KNN_Q15 = q15
        for _ in range(2):
            top, keep = LV.knn_sweep(classes, shots, q, det, 10, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            top, keep = LV.knn_sweep(classes, shots, q, det, 10, True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        res.setdefault(q15, []).append(dt * 1e3)
        res[("out", q15)] = (top.clone(), keep.clone())
    same = torch.equal(res[("out", False)][0], res[("out", True)][0]) and torch.equal(res[("out", False)][1], res[("out", True)][1])
    # the GEMM alone
    mu = K.colmean(shots)
    sn, sh, _ = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1)
    _, qh, den = K.rownorm_h(q, mu=mu, eps=1e-8, mode=1, want_rows=False)
    tg = {}
    for q15 in (False, True):
        for _ in range(3): K.gemm_f16(qh, sh, q15=q15)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): K.gemm_f16(qh, sh, q15=q15)
        e1.record(); torch.cuda.synchronize()
        tg[q15] = e0.elapsed_time(e1) / 10
    print("%-10s sweep fp32 %s ms | q15 %s ms | outputs identical: %s | GEMM alone fp32 %.3f ms (%.0f TF/s) q15 %.3f ms (%.0f TF/s)" % (
        kind, ["%.3f" % v for v in res[False]], ["%.3f" % v for v in res[True]], same, tg[False], 2.0 * Q * S * D / tg[False] / 1e9, tg[True], 2.0 * Q * S * D / tg[True] / 1e9))
