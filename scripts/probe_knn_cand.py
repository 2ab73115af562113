"""Candidate statistics of the two-stage sweep on the benchmark's synthetic descriptors."""
import sys
import torch
sys.path.insert(0, ".")
from lvc_amd import kernels as K
from lvc_amd import label_verification as LV
D = "cuda:0"
g = torch.Generator().manual_seed(0)
S, Dm, Q = 2400, 1024, 20000
classes = torch.arange(80).repeat_interleave(30)
centers = torch.randn(80, Dm, generator=g)
shots = (centers[classes] + 2.0 * torch.randn(S, Dm, generator=g) + 0.3).to(D)
qcls = torch.randint(0, 80, (Q,), generator=g)
q = (centers[qcls] + 2.5 * torch.randn(Q, Dm, generator=g) + 0.3).to(D)
mu = K.colmean(shots)
sn, sh, _ = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1)
qn, qh, den = K.rownorm_h(q, mu=mu, eps=1e-8, mode=1)
ap = K.gemm_f16(qh, sh)
m = LV.VERIFY_MARGIN
srt = torch.sort(ap, dim=1, descending=True)[0]
A10 = srt[:, 9:10]
ncand = (ap >= A10 - m).sum(1).float()
print("top values row0", srt[0, :14].tolist())
print("ncand mean %.1f  p50 %.0f  p99 %.0f  max %.0f" % (ncand.mean(), ncand.median(), ncand.quantile(0.99), ncand.max()))
top = srt[:, :40]
gap_prev = torch.cat([torch.full_like(top[:, :1], 9.0), top[:, :-1] - top[:, 1:]], 1)
gap_next = torch.cat([top[:, :-1] - top[:, 1:], torch.full_like(top[:, :1], 9.0)], 1)
is_c = top >= A10 - m
amb = is_c & ((gap_prev <= m) | ((gap_next <= m) & torch.cat([is_c[:, 1:], is_c[:, :1] & False], 1)))
print("namb mean %.1f" % amb.sum(1).float().mean())
exact = qn @ sn.t()
print("max |approx - exact| %.2e" % (ap - exact).abs().max())
