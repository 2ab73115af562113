import os, sys
import torch
sys.path.insert(0, ".")
from lvc_amd import kernels as K
from lvc_amd import label_verification as LV
D = "cuda:0"
g = torch.Generator().manual_seed(0)
S, Dm, Q = 2400, 1024, 120000
classes = torch.arange(80).repeat_interleave(30).to(D)
centers = torch.randn(80, Dm, generator=g)
shots = (centers[classes.cpu()] + 2.0 * torch.randn(S, Dm, generator=g) + 0.3).to(D)
qcls = torch.randint(0, 80, (Q,), generator=g)
q = (centers[qcls] + 2.5 * torch.randn(Q, Dm, generator=g) + 0.3).to(D)
qcls = qcls.to(D)
mu = K.colmean(shots)
sn, sh, _ = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1)
qn, qh, den = K.rownorm_h(q, mu=mu, eps=1e-8, mode=1)
ap = K.gemm_f16(qh, sh)
f = lambda: K.knn_verify_topk_vote(ap, q, sn, LV.VERIFY_MARGIN, classes, qcls, 10, mu=mu, den=den)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record()
torch.cuda.synchronize()
print("stop=%s  %.3f ms" % (os.environ.get("LVC_KV_STOP", "0"), e0.elapsed_time(e1) / 10))
