"""BASELINE config 4: label-verification kNN sweep, 120k x 1024 queries vs 2400 shots (80 classes x 30), one GPU.
Reports time, algorithmic TFLOP/s and GB/s, and the CPU oracle (dense GEMM form) on a bounded sample."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd.label_verification import knn_sweep
from oracle import knn as oknn

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Q, S, D = 120000, 2400, 1024
classes = torch.arange(80).repeat_interleave(30)
shots = torch.randn(S, D, generator=g)
q = torch.randn(Q, D, generator=g)
det = torch.randint(0, 80, (Q,), generator=g)
sd, cd, qd, dd = shots.to(dev), classes.to(dev), q.to(dev), det.to(dev)
for _ in range(2):
    top, keep = knn_sweep(cd, sd, qd, dd, 10, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    top, keep = knn_sweep(cd, sd, qd, dd, 10, True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flops = 2.0 * Q * S * D
bytes_alg = Q * D * 4 + S * D * 4 + Q * 10 * 8
# CPU oracle (dense) on a 10k-query sample
torch.set_num_threads(min(os.cpu_count() or 1, 32))
t1 = time.perf_counter()
ref = oknn.dense(classes, shots, q[:10000], True)
cdt = time.perf_counter() - t1
agree = float((top[:10000].cpu() == ref).all(dim=1).float().mean())
print(json.dumps({"workload": "kNN sweep Q=120000 S=2400 D=1024 cosine, k=10", "ms": round(dt * 1e3, 2),
                  "queries_per_s": round(Q / dt), "algorithmic_tflops": round(flops / dt / 1e12, 1),
                  "algorithmic_GBps": round(bytes_alg / dt / 1e9, 1), "hbm_frac_of_8TBps": round(bytes_alg / dt / 8e12, 4),
                  "cpu_oracle_dense_queries_per_s": round(10000 / cdt), "cpu_threads": torch.get_num_threads(),
                  "top10_rows_identical_to_oracle_on_10k_sample": agree}))
