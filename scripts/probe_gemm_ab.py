"""Two builds of the fp16 pre-filter GEMM against each other in one process, alternating (box-to-box and warm-up effects are several
per cent): lvc_amd/liblvc_amd.so and a second library given as argv[1]."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K, _lib
from lvc_amd.kernels import ptr, c_int
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Q, S, D = 120000, 2400, 1024
shots, q = torch.randn(S, D, generator=g).to(dev), torch.randn(Q, D, generator=g).to(dev)
mu = K.colmean(shots)
sn, sh, _ = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1)
_, qh, den = K.rownorm_h(q, mu=mu, eps=1e-8, mode=1, want_rows=False)
libs = {"new": _lib.lib(), "other": ctypes.CDLL(os.path.abspath(sys.argv[1]))}
y16 = torch.empty(Q, S, dtype=torch.int16, device=dev)
y32 = torch.empty(Q, S, dtype=torch.float32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(L, q15):
    if q15:
        rc = L.lvc_gemm_f16_q15(ptr(qh), ptr(sh), c_int(D), ptr(y16), c_int(Q), c_int(S), c_int(D), c_int(S), st)
    else:
        rc = L.lvc_gemm_f16(ptr(qh), ptr(sh), c_int(D), ptr(y32), c_int(Q), c_int(S), c_int(D), c_int(S), st)
    assert rc == 0
res = {}
for rnd in range(6):
    for name, L in libs.items():
        for q15 in (True, False):
            for _ in range(2): run(L, q15)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(L, q15)
            e1.record(); torch.cuda.synchronize()
            if rnd: res.setdefault((name, q15), []).append(e0.elapsed_time(e1) / 10)
for k, v in res.items():
    print(k, " ".join("%.3f" % t for t in v), "| median %.3f ms = %.0f TF/s" % (sorted(v)[len(v) // 2], 2.0 * Q * S * D / sorted(v)[len(v) // 2] / 1e9))
a = y16.clone(); run(libs["other"], True); torch.cuda.synchronize()
b = y16.clone(); run(libs["new"], True); torch.cuda.synchronize()
print("q15 outputs of the two builds differ in %d of %d entries, max |diff| %d" % (int((y16 != b).sum()), y16.numel(), int((y16.int() - b.int()).abs().max())))
