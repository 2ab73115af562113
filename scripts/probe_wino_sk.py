"""conv3x3_wino.hip, stream-K against one workgroup per tile against the direct kernel on the detector's 256 / 128-channel 3x3 layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
LAYERS = [("p2 256>256", 8, 200, 336, 256, 256), ("p3 256>256", 8, 100, 168, 256, 256), ("p4 / res4 256>256", 8, 50, 84, 256, 256), ("p5 256>256", 8, 25, 42, 256, 256),
          ("res3 128>128", 8, 100, 168, 128, 128), ("res5 512>512", 8, 25, 42, 512, 512), ("p2 of 2 images", 2, 200, 336, 256, 256), ("res4 of 2 images", 2, 50, 84, 256, 256)]
def timeit(f):
    for i in range(3): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): f(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
for name, N, H, W, C, K in LAYERS:
    xs = [torch.randn(N, H, W, C, device=d).relu_() for _ in range(3)]
    w = torch.randn(K, C, 3, 3, device=d) * (2.0 / (9 * C)) ** 0.5
    pc = k.pack_conv(w, bias=torch.randn(K, device=d) * 0.1, pad=1)
    k.CONV_WINO = False
    t_d = timeit(lambda i: k.conv2d_nhwc(xs[i % 3], pc, relu=True))
    k.set_wino_streamk(False)
    t_t = timeit(lambda i: k.conv3x3_wino(xs[i % 3], pc, relu=True))
    a = k.conv3x3_wino(xs[0], pc, relu=True).clone()
    from lvc_amd import _lib
    _lib.lib().lvc_set_wino_streamk(2)
    t_w = timeit(lambda i: k.conv3x3_wino(xs[i % 3], pc, relu=True))
    k.set_wino_streamk(True)
    t_s = timeit(lambda i: k.conv3x3_wino(xs[i % 3], pc, relu=True))
    b = k.conv3x3_wino(xs[0], pc, relu=True)
    fl = 2.0 * N * H * W * C * K * 9 / 1e9
    print("%-20s tiles %5d | direct %.4f ms %6.1f TF/s | wino per tile %.4f %6.1f | persistent whole tiles %.4f | wino stream-K %.4f %6.1f | max |sk - tile| %.1e  word %d"
          % (name, k.wino_tiles(N, H, W, K), t_d, fl / t_d, t_t, fl / t_t, t_w, t_s, fl / t_s, float((a - b).abs().max()), k.conv_error_word(d)), flush=True)
