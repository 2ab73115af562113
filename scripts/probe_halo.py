"""Halo 3x3 kernel vs the generic bf16x3 kernel: max difference (and vs an fp64 torch conv) and time per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lvc_amd import kernels as k
d = torch.device("cuda:0")
shapes = [  # N, C, H, W, K
    (1, 32, 7, 9, 128), (2, 64, 13, 21, 128), (1, 256, 25, 42, 256), (2, 128, 50, 84, 132),
    (8, 256, 200, 336, 256), (8, 256, 100, 168, 256), (8, 128, 100, 168, 128), (8, 256, 50, 84, 256),
    (8, 512, 25, 42, 512), (8, 256, 25, 42, 256), (8, 256, 13, 21, 256),
]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
torch.manual_seed(0)
for (N, C, H, W, K) in shapes:
    x = torch.randn(N, H, W, C, device=d)
    w = torch.randn(K, C, 3, 3, device=d) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K, device=d) * 0.1
    res = torch.randn(N, H, W, K, device=d)
    pc = k.pack_conv(w, bias=b, stride=1, pad=1)
    out = {}
    for halo in (False, True, "f16x2"):
        k.CONV_HALO = bool(halo)
        k.CONV_SPLIT = "f16x2" if halo == "f16x2" else "bf16x3"
        y = k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=1, out=y)
        e1.record(); torch.cuda.synchronize()
        out[halo] = (y.clone(), e0.elapsed_time(e1) / reps)
    fl = 2.0 * N * H * W * K * C * 9
    line = "N%d C%d %dx%d K%d: generic %.3f ms %.1f TF | halo %.3f ms %.1f TF | halo f16x2 %.3f ms %.1f TF | max|halo-generic| %.3g" % (
        N, C, H, W, K, out[False][1], fl / out[False][1] / 1e9, out[True][1], fl / out[True][1] / 1e9,
        out["f16x2"][1], fl / out["f16x2"][1] / 1e9, (out[True][0] - out[False][0]).abs().max().item())
    if N * H * W * K * C < 3e9:
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + res.double()
        ref = ref.clamp_min(0)
        line += " | vs fp64: halo %.3g f16x2 %.3g generic %.3g" % ((out[True][0].double() - ref).abs().max().item(),
                                                        (out["f16x2"][0].double() - ref).abs().max().item(),
                                                        (out[False][0].double() - ref).abs().max().item())
    print(line, flush=True)
