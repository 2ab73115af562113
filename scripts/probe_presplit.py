"""conv2 -> conv3 of a res4 / res5 bottleneck: the fp32 hand-over (conv3 splits its operand) against the pre-split planes
(kernels.conv3x3_conv1x1_presplit), per launch.  python scripts/probe_presplit.py  (GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k

d = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)


def bn(c):
    return [t.to(d) for t in (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1,
                              torch.rand(c, generator=g) + 0.5)]


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000.0


for (N, H, W, cb, cout) in [(8, 50, 84, 256, 1024), (8, 25, 42, 512, 2048)]:
    p2 = k.pack_conv((torch.randn(cb, cb, 3, 3, generator=g) * (2.0 / (9 * cb)) ** 0.5).to(d), bn=bn(cb), pad=1)
    p3 = k.pack_conv((torch.randn(cout, cb, 1, 1, generator=g) * (2.0 / cb) ** 0.5).to(d), bn=bn(cout))
    t = torch.randn(N, H, W, cb, generator=g).relu_().to(d)
    res = torch.randn(N, H, W, cout, generator=g).relu_().to(d)
    assert k.presplit_pair_ok(t, p2, p3, res)
    z = k.conv2d_nhwc(t, p2, relu=True)
    a2 = timeit(lambda: k.conv2d_nhwc(t, p2, relu=True))
    a3 = timeit(lambda: k.conv2d_nhwc(z, p3, relu=True, residual=res, res_mode=1))
    ab = timeit(lambda: k.conv2d_nhwc(k.conv2d_nhwc(t, p2, relu=True), p3, relu=True, residual=res, res_mode=1))
    pb = timeit(lambda: k.conv3x3_conv1x1_presplit(t, p2, p3, residual=res))
    same = torch.equal(k.conv3x3_conv1x1_presplit(t, p2, p3, residual=res), k.conv2d_nhwc(z, p3, relu=True, residual=res, res_mode=1))
    print("%dx%dx%d %d->%d: conv2 %.1f us, conv3 %.1f us, pair %.1f us | presplit pair %.1f us (%.1f %%), bit-identical %s"
          % (N, H, W, cb, cout, a2, a3, ab, pb, 100.0 * (pb / ab - 1.0), same))
