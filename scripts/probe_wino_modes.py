"""conv3x3_wino.hip work distribution: one workgroup per tile (0) / stream-K with hand-off (1) / persistent workgroups on whole tiles (2),
alternated five times on the layers the detector routes to it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k, _lib
d = torch.device("cuda:0")
torch.manual_seed(0)
LAYERS = [("p2 256>256", 8, 200, 336, 256, 256), ("p3 256>256", 8, 100, 168, 256, 256), ("res3 128>128", 8, 100, 168, 128, 128), ("res4 256>256", 8, 50, 84, 256, 256)]
def timeit(f, n=20):
    for i in range(3): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): f(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, N, H, W, C, K in LAYERS:
    xs = [torch.randn(N, H, W, C, device=d).relu_() for _ in range(3)]
    w = torch.randn(K, C, 3, 3, device=d) * (2.0 / (9 * C)) ** 0.5
    pc = k.pack_conv(w, bias=torch.randn(K, device=d) * 0.1, pad=1)
    res = {0: [], 1: [], 2: []}
    for rep in range(5):
        for mode in (0, 1, 2):
            _lib.lib().lvc_set_wino_streamk(mode)
            res[mode].append(timeit(lambda i: k.conv3x3_wino(xs[i % 3], pc, relu=True)))
    print("%-14s" % name, " | ".join("mode %d: %s (min %.4f)" % (m, " ".join("%.4f" % t for t in res[m]), min(res[m])) for m in (0, 1, 2)), flush=True)
