import sys, os
sys.path.insert(0, "/root/repo")
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
for (N, H, W, C, K) in [(8, 200, 336, 256, 256), (8, 100, 168, 256, 256), (8, 100, 168, 128, 128), (8, 50, 84, 256, 256), (8, 25, 42, 512, 512)]:
    x = torch.randn(N, H, W, C, device=d)
    w = torch.randn(K, C, 3, 3, device=d) * 0.02
    pc = k.pack_conv(w, stride=1, pad=1)
    y = k.conv2d_nhwc(x, pc, relu=True)
    for _ in range(3): k.conv2d_nhwc(x, pc, relu=True, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): k.conv2d_nhwc(x, pc, relu=True, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-24s %.3f ms %6.1f TF/s" % ((N, H, W, C, K), ms, 2.0 * N * H * W * C * K * 9 / ms / 1e9), flush=True)
