"""Ten kNN sweeps (cfg 4: 120000 x 2400 x 1024, randn) for `rocprofv3 --kernel-trace --stats`: which launches a sweep consists of."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import label_verification as LV
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Q, S, D = 120000, 2400, 1024
classes = torch.arange(80).repeat_interleave(30).to(dev)
shots, q = torch.randn(S, D, generator=g).to(dev), torch.randn(Q, D, generator=g).to(dev)
det = torch.randint(0, 80, (Q,), generator=g).to(dev)
for _ in range(3):
    LV.knn_sweep(classes, shots, q, det, 10, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    LV.knn_sweep(classes, shots, q, det, 10, True)
torch.cuda.synchronize()
print("ms per sweep %.3f" % ((time.perf_counter() - t0) * 100))
