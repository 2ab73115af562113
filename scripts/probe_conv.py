"""Micro-benchmark of the implicit-GEMM conv kernel on the path's dominant shapes (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k

d = torch.device("cuda:0")
shapes = [  # N, C, H, W, K, R, stride, pad, name
    (8, 256, 200, 336, 256, 3, 1, 1, "fpn_output2/rpn p2 3x3"),
    (8, 64, 200, 336, 256, 1, 1, 0, "res2 conv3 1x1"),
    (8, 256, 200, 336, 64, 1, 1, 0, "res2 conv1 1x1"),
    (8, 64, 200, 336, 64, 3, 1, 1, "res2 conv2 3x3"),
    (8, 128, 100, 168, 128, 3, 1, 1, "res3 conv2 3x3"),
    (8, 256, 50, 84, 256, 3, 1, 1, "res4 conv2 3x3"),
    (8, 1024, 50, 84, 256, 1, 1, 0, "res4 conv1 1x1"),
    (8, 512, 25, 42, 512, 3, 1, 1, "res5 conv2 3x3"),
    (8, 512, 25, 42, 2048, 1, 1, 0, "res5 conv3 1x1"),
    (8000, 12544, 1, 1, 1024, 1, 1, 0, "fc1"),
]
for (N, C, H, W, K, R, s, p, name) in shapes:
    x = torch.randn(N, H, W, C, device=d)
    w = torch.randn(K, C, R, R, device=d) * 0.02
    pc = k.pack_conv(w, stride=s, pad=p)
    y = k.conv2d_nhwc(x, pc, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5
    e0.record()
    for _ in range(it):
        k.conv2d_nhwc(x, pc, relu=True, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    Ho, Wo = y.shape[1], y.shape[2]
    fl = 2.0 * N * Ho * Wo * K * C * R * R
    print("%-26s %8.3f ms  %7.1f TF/s" % (name, ms, fl / ms / 1e9))
