"""The 3x3 stride-1 layers of R50-FPN at batch 8 (800x1333), each timed alone on the round-1 kernel (conv3x3_halo_h2.hip, main +
cross accumulators) and on the round-3 kernel (conv3x3_halo_s1.hip, one accumulator, pipelined tap loop): ms, TF/s, and the
largest difference between the two against an fp64 reference on a slice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
LAYERS = [("res2 conv2 64>64", 8, 200, 336, 64, 64, 3), ("res3 conv2 128>128", 8, 100, 168, 128, 128, 4),
          ("res4 conv2 256>256", 8, 50, 84, 256, 256, 6), ("res5 conv2 512>512", 8, 25, 42, 512, 512, 3),
          ("p2 256>256 (fpn out, rpn)", 8, 200, 336, 256, 256, 2), ("p3 256>256", 8, 100, 168, 256, 256, 2),
          ("p4 256>256", 8, 50, 84, 256, 256, 2), ("p5 256>256", 8, 25, 42, 256, 256, 2)]
k._HALO_H2_MIN_TILES = 0
MODES = (0, 1, 2)
tot = {m: 0.0 for m in MODES}
for name, N, H, W, C, K, count in LAYERS:
    xs = [torch.randn(N, H, W, C, device=d).relu_() for _ in range(2)]
    w = torch.randn(K, C, 3, 3, device=d) * (2.0 / (9 * C)) ** 0.5
    pc = k.pack_conv(w, stride=1, pad=1)
    y = torch.empty(N, H, W, K, device=d)
    res = {}
    for s1 in MODES:
        k.HALO_S1 = s1
        for i in range(4): k.conv2d_nhwc(xs[i % 2], pc, relu=True, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): k.conv2d_nhwc(xs[i % 2], pc, relu=True, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        tot[s1] += ms * count
        res[s1] = (ms, k.conv2d_nhwc(xs[0], pc, relu=True).clone())
    ref = F.conv2d(xs[0][:1, :24].permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1).clamp_min(0)[:, 1:-1]
    sc = float(ref.abs().max())
    rms = [float((res[s][1][:1, 1:23].double() - ref).pow(2).mean().sqrt()) / sc for s in MODES]
    cpu = F.conv2d(xs[0][:1, :24].permute(0, 3, 1, 2).cpu(), w.cpu(), padding=1).permute(0, 2, 3, 1).clamp_min(0)[:, 1:-1]
    rms_cpu = float((cpu.double() - ref.cpu()).pow(2).mean().sqrt()) / sc
    fl = 2.0 * N * H * W * C * K * 9
    print("%-28s x%d  h2 %.4f ms %6.1f TF/s | pipe(2acc) %.4f ms %6.1f TF/s (%.3fx) | pipe(1acc) %.4f ms %6.1f TF/s (%.3fx)  rms err/scale h2 %.2e 2acc %.2e 1acc %.2e cpu-fp32 %.2e" % (
        name, count, res[0][0], fl / res[0][0] / 1e9, res[1][0], fl / res[1][0] / 1e9, res[0][0] / res[1][0],
        res[2][0], fl / res[2][0] / 1e9, res[0][0] / res[2][0], rms[0], rms[1], rms[2], rms_cpu), flush=True)
print("per step (launch counts applied): h2 %.3f ms, pipe two-acc %.3f ms, pipe one-acc %.3f ms; conv error word %d" % (tot[0], tot[1], tot[2], k.conv_error_word(d)))
