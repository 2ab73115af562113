"""Practical HBM ceilings (torch elementwise kernels) next to the 1x1 expansion layer, with / without residual."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
def tm(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
a = torch.randn(8, 200, 336, 256, device=d); b = torch.randn_like(a); c = torch.empty_like(a)
gb = a.numel() * 4 / 1e9
ms = tm(lambda: c.copy_(a)); print("copy  %.3f ms %.2f TB/s" % (ms, 2 * gb / ms))
ms = tm(lambda: torch.add(a, b, out=c)); print("add   %.3f ms %.2f TB/s" % (ms, 3 * gb / ms))
ms = tm(lambda: a.sum()); print("sum   %.3f ms %.2f TB/s" % (ms, gb / ms))
ms = tm(lambda: c.fill_(1.0)); print("fill  %.3f ms %.2f TB/s" % (ms, gb / ms))
x = torch.randn(8, 200, 336, 64, device=d)
for K in (256, 128):
    w = torch.randn(K, 64, 1, 1, device=d) * 0.1
    pc = k.pack_conv(w)
    res = torch.randn(8, 200, 336, K, device=d)
    y = torch.empty(8, 200, 336, K, device=d)
    for rm in (0, 1):
        ms = tm(lambda: k.conv2d_nhwc(x, pc, relu=True, residual=res if rm else None, res_mode=rm, out=y))
        byt = (x.numel() + y.numel() * (2 if rm else 1)) * 4 / 1e9
        print("1x1 64->%d res%d: %.3f ms %.2f TB/s" % (K, rm, ms, byt / ms))
    for eng in ("f32",):
        k.CONV_ENGINE = eng
        ms = tm(lambda: k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=1, out=y))
        print("   engine %s res1: %.3f ms" % (eng, ms))
        k.CONV_ENGINE = "bf16x3"
