"""Fused stem (conv7x7/2 + BN + ReLU + maxpool) vs the unfused pair, batch 8 x 800 x 1344."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
x4 = torch.randn(8, 800, 1344, 4, device=d) * 50
x4[..., 3] = 0
w = torch.randn(64, 3, 7, 7, device=d) * 0.05
bn = [torch.rand(64, device=d) + 0.5, torch.randn(64, device=d), torch.randn(64, device=d), torch.rand(64, device=d) + 0.5]
pc = k.pack_conv(w, bn=bn, stride=2, pad=3, stem=True)
def tm(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
a = k.stem_conv_pool(x4, pc)
b = k.maxpool2d_nhwc(k.conv2d_nhwc(x4, pc, relu=True), 3, 2, 1)
print("max diff fused vs unfused", float((a - b).abs().max()), "scale", float(b.abs().max()))
print("fused   %.3f ms" % tm(lambda: k.stem_conv_pool(x4, pc)))
print("unfused %.3f ms" % tm(lambda: k.maxpool2d_nhwc(k.conv2d_nhwc(x4, pc, relu=True), 3, 2, 1)))
