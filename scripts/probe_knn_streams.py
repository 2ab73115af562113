"""Does the row normalisation of the next query chunk run UNDER the pre-filter GEMM of the current one (second stream)?  The GEMM
holds one workgroup of 8 waves x 200 registers per CU: 112 registers per SIMD lane are left for a co-resident wave."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K, label_verification as LV
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Q, S, D = 120000, 2400, 1024
classes = torch.arange(80).repeat_interleave(30).to(dev)
shots, q = torch.randn(S, D, generator=g).to(dev), torch.randn(Q, D, generator=g).to(dev)
det = torch.randint(0, 80, (Q,), generator=g).to(dev)
mu = K.colmean(shots)
sn, sh, _, sres = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1, want_resid=True)
sres_max = K.max_f32(sres)
acc = LV.pre_filter_acc(D)
side = torch.cuda.Stream()


def stage1(qc):
    _, qh, den, qres = K.rownorm_h(qc, mu=mu, eps=1e-8, mode=1, want_rows=False, want_resid=True)
    return qh, den, K.knn_margins(qres, sres_max, acc, LV.Q15_MARGIN)


def rest(qc, dc, qh, den, margins):
    return K.knn_verify_topk_vote(K.gemm_f16(qh, sh, q15=True), qc, sn, LV.VERIFY_MARGIN + LV.Q15_MARGIN, classes, dc, 10, mu=mu, den=den, margins=margins)


def serial(nch):
    outs = []
    for qc, dc in zip(q.chunk(nch), det.chunk(nch)):
        outs.append(rest(qc, dc, *stage1(qc)))
    return outs


def overlapped(nch):
    """stage 1 of chunk i + 1 on the side stream while the main stream runs GEMM + verification of chunk i"""
    main = torch.cuda.current_stream()
    qs, ds = q.chunk(nch), det.chunk(nch)
    outs = []
    s1 = stage1(qs[0])
    for i in range(nch):
        nxt = None
        if i + 1 < nch:
            side.wait_stream(main) if i == 0 else None
            with torch.cuda.stream(side):
                nxt = stage1(qs[i + 1])
            ev = torch.cuda.Event(); ev.record(side)
        outs.append(rest(qs[i], ds[i], *s1))
        if nxt is not None:
            main.wait_event(ev)
            for t in nxt:
                t.record_stream(main)      # allocated on the side stream, read on the main one
            s1 = nxt
    return outs


def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = serial(1)[0]
for nch in (1, 2, 3, 4, 6, 8):
    a = timed(lambda: serial(nch))
    b = timed(lambda: overlapped(nch)) if nch > 1 else float("nan")
    o = overlapped(nch) if nch > 1 else serial(1)
    same = torch.equal(torch.cat([t[0] for t in o]), ref[0])
    print("%d chunk(s): serial %.3f ms | next chunk's row normalisation on a side stream %.3f ms | top-10 identical to one chunk: %s" % (nch, a, b, same))
