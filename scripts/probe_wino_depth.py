"""conv3x3_wino.hip: time per tile against the number of stages (input channels 64 .. 1024 at fixed 8x200x336 pixels x 256 outputs):
t = a + b * stages -> a = the tile's fixed cost (prologue, epilogue, exchange), b = the stage loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
N, H, W, K = 8, 200, 336, 256
pts = []
for C in (64, 128, 256, 512, 768):
    xs = [torch.randn(N, H, W, C, device=d).relu_() for _ in range(2)]
    w = torch.randn(K, C, 3, 3, device=d) * (2.0 / (9 * C)) ** 0.5
    pc = k.pack_conv(w, bias=torch.zeros(K, device=d), pad=1)
    for mode in ("wino", "direct"):
        f = (lambda i: k.conv3x3_wino(xs[i % 2], pc, relu=True)) if mode == "wino" else (lambda i: k.conv2d_nhwc(xs[i % 2], pc, relu=True))
        k.CONV_WINO = False
        for i in range(3): f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10): f(i)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tiles = k.wino_tiles(N, H, W, K)
        print("%-6s C %4d  %.4f ms  %6.1f TF/s  stages/tile %3d  us per tile-slot (256 CUs) %.2f" % (mode, C, ms, 2.0 * N * H * W * C * K * 9 / ms / 1e9, 3 * C // 16, ms * 1e3 * 256 / tiles), flush=True)
        if mode == "wino": pts.append((3 * C // 16, ms * 1e3 * 256 / tiles))
(x0, y0), (x1, y1) = pts[1], pts[-1]
b = (y1 - y0) / (x1 - x0)
print("stage %.3f us, fixed per tile %.2f us" % (b, y0 - b * x0))
