"""kNN sweep time vs query chunk size (rows of the similarity matrix materialised at once): a chunk whose similarity block and
normalised rows fit the 256 MB Infinity Cache never sends them to HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import label_verification as LV
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
Q, S, D = 120000, 2400, 1024
q = torch.randn(Q, D, device=dev, generator=g)
shots = torch.randn(S, D, device=dev, generator=g)
cls = torch.arange(80, device=dev).repeat_interleave(30)
det = torch.randint(0, 80, (Q,), device=dev)
ref = None
for chunk in (32768, 24576, 16384, 12288, 8192, 4096, 32768, 12288):
    LV.QUERY_CHUNK = chunk
    for _ in range(2): top, keep = LV.knn_sweep(cls, shots, q, det, 10, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): top, keep = LV.knn_sweep(cls, shots, q, det, 10, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    if ref is None: ref = top
    print("chunk %6d: %.3f ms  same result: %s" % (chunk, dt * 1e3, bool(torch.equal(top, ref))), flush=True)
