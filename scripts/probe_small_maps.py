"""The 3x3 layers on the small pyramid levels (p5: 25 x 42, p6: 13 x 21, batch 8): the bf16x3 halo kernel they run on
(kernels._HALO_H2_MIN_TILES = 128) against the fp16-split pipelined kernel (threshold 0), one- and two-accumulator forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
for name, H, W in (("p5 25x42", 25, 42), ("p6 13x21", 13, 21), ("p4 50x84", 50, 84)):
    x = torch.randn(8, H, W, 256, device=d).relu_()
    w = torch.randn(256, 256, 3, 3, device=d) * (2.0 / (9 * 256)) ** 0.5
    out = []
    for two_acc in (False, True):
        pc = k.pack_conv(w, stride=1, pad=1)
        pc.two_acc = two_acc
        for thr in (128, 0):
            k._HALO_H2_MIN_TILES = thr
            y = k.conv2d_nhwc(x, pc, relu=True)
            for _ in range(5): k.conv2d_nhwc(x, pc, relu=True, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): k.conv2d_nhwc(x, pc, relu=True, out=y)
            e1.record(); torch.cuda.synchronize()
            out.append("%s thr %3d: %.4f ms" % ("two-acc" if two_acc else "one-acc", thr, e0.elapsed_time(e1) / 50))
    print(name, " | ".join(out))
