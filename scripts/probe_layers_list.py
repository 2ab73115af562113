"""Time the non-3x3 conv/GEMM layer shapes of the R50-FPN forward (batch 8) one by one; check against torch fp64 on a slice."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lvc_amd import kernels as k
d = torch.device("cuda:0")
L = [  # N,H,W,C,K,R,stride,pad,residual
    (8, 200, 336, 64, 256, 1, 1, 0, 1), (8, 100, 168, 128, 512, 1, 1, 0, 1), (8, 50, 84, 256, 1024, 1, 1, 0, 1),
    (8000, 1, 1, 12544, 1024, 1, 1, 0, 0), (8, 50, 84, 1024, 256, 1, 1, 0, 0), (8, 200, 336, 256, 256, 1, 1, 0, 2),
    (8, 25, 42, 512, 2048, 1, 1, 0, 1), (8, 100, 168, 512, 128, 1, 1, 0, 0), (8, 200, 336, 256, 64, 1, 1, 0, 0),
    (8, 100, 168, 512, 256, 1, 1, 0, 2), (8, 200, 336, 256, 512, 1, 2, 0, 0), (8, 100, 168, 512, 1024, 1, 2, 0, 0),
    (8, 25, 42, 2048, 512, 1, 1, 0, 0), (8, 50, 84, 1024, 2048, 1, 2, 0, 0), (8000, 1, 1, 1024, 1024, 1, 1, 0, 0),
    (8000, 1, 1, 1024, 404, 1, 1, 0, 0), (8, 200, 336, 64, 64, 3, 1, 1, 0), (8, 200, 336, 64, 64, 1, 1, 0, 0),
]
torch.manual_seed(0)
tot = 0.0
for (N, H, W, C, K, R, st, pad, rm) in L:
    x = torch.randn(N, H, W, C, device=d)
    w = torch.randn(K, C, R, R, device=d) * (2.0 / (R * R * C)) ** 0.5
    pc = k.pack_conv(w, stride=st, pad=pad)
    Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    res = None
    if rm == 1: res = torch.randn(N, Ho, Wo, K, device=d)
    if rm == 2: res = torch.randn(N, Ho // 2, Wo // 2, K, device=d)
    y = k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=rm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=rm, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tot += ms
    # check 512 output pixels of image 0 against fp64
    xs = x[:1].permute(0, 3, 1, 2).double()
    ref = F.conv2d(xs, w.double(), stride=st, padding=pad).permute(0, 2, 3, 1)
    if rm == 1: ref = ref + res[:1].double()
    if rm == 2: ref = ref + res[:1].double().repeat_interleave(2, 1).repeat_interleave(2, 2)
    err = (y[:1].double() - ref.clamp_min(0)).abs().max().item()
    fl = 2.0 * N * Ho * Wo * K * C * R * R
    gb = 4.0 * (x.numel() / (st * st if R == 1 else 1) + y.numel() + (res.numel() if res is not None else 0)) / 1e9
    print("%-28s K%-5d R%d s%d res%d: %.3f ms %6.1f TF/s %5.2f TB/s  err %.2g" % ((N, H, W, C), K, R, st, rm, ms, fl / ms / 1e9, gb / ms, err), flush=True)
print("total %.3f ms" % tot)
