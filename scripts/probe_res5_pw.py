"""res5's pointwise layers (2048 output channels = 16 column tiles) and the box head: time per layer (A/B of the grouping of a row tile's
column tiles on one XCD: LVC_AMD_LIB = the previous build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
LAYERS = [("res5 c3 512>2048+r", 8, 25, 42, 512, 2048, True, False), ("res5.0 c3+proj 1536>2048", 8, 25, 42, 1536, 2048, False, False),
          ("res5 c1 2048>512", 8, 25, 42, 2048, 512, False, False), ("res4 c3 256>1024+r", 8, 50, 84, 256, 1024, True, False), ("fc1 12544>1024", 8000, 1, 1, 12544, 1024, False, True)]
for name, N, H, W, C, K, res, two in LAYERS:
    xs = [torch.randn(N, H, W, C, device=d).relu_() for _ in range(3)]
    r = torch.randn(N, H, W, K, device=d) if res else None
    w = torch.randn(K, C, 1, 1, device=d) * (2.0 / C) ** 0.5
    pc = k.pack_conv(w, bias=torch.randn(K, device=d) * 0.1)
    pc.two_acc = two
    f = lambda i: k.conv2d_nhwc(xs[i % 3], pc, relu=True, residual=r)
    ts = []
    for rep in range(3):
        for i in range(3): f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): f(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    print("%-28s %s ms" % (name, " ".join("%.4f" % t for t in ts)), flush=True)
