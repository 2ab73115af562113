"""conv3x3_wino.hip against the direct single-accumulator kernel and fp64 on the 3x3 layers of the batch of eight: ms, TF/s, errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
LAYERS = [("fpn out p2 256>256", 8, 200, 336, 256, 256), ("fpn out p3", 8, 100, 168, 256, 256), ("fpn out p4", 8, 50, 84, 256, 256),
          ("res3 conv2 128>128", 8, 100, 168, 128, 128), ("res4 conv2 256>256", 8, 50, 84, 256, 256), ("ragged 3x37x53 64>192", 3, 37, 53, 64, 192)]
for name, N, H, W, C, K in LAYERS:
    xs = [torch.randn(N, H, W, C, device=d).relu_() for _ in range(3)]
    w = torch.randn(K, C, 3, 3, device=d) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K, device=d) * 0.1
    pc = k.pack_conv(w, bias=b, pad=1)
    line = "%-24s" % name
    outs = {}
    for mode in ("direct", "wino"):
        f = (lambda i: k.conv2d_nhwc(xs[i % 3], pc, relu=True)) if mode == "direct" else (lambda i: k.conv3x3_wino(xs[i % 3], pc, relu=True))
        k.CONV_WINO = False
        for i in range(4): f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): f(i)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        outs[mode] = f(0).clone()
        line += " (word %d)" % k.conv_error_word(d)
        line += " | %s %.4f ms %6.1f TF/s" % (mode, ms, 2.0 * N * H * W * C * K * 9 / ms / 1e9)
    n = min(N, 2)
    ref = torch.nn.functional.conv2d(xs[0][:n].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).relu().permute(0, 2, 3, 1)
    sc = float(ref.abs().max())
    for mode in outs:
        e = (outs[mode][:n].double() - ref).abs()
        line += " | %s err max %.1e rms %.1e" % (mode, float(e.max()) / sc, float(e.pow(2).mean().sqrt()) / sc)
    print(line, "| word", k.conv_error_word(d), flush=True)
    k.clear_conv_error_word(d)
