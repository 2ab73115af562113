"""Time a few pointwise shapes on the DMA kernel (one process per ablation setting: LVC_PW_ABLATE is read once)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
L = [(8, 50, 84, 1024, 256, 1, 0), (8000, 1, 1, 12544, 1024, 1, 0), (8, 100, 168, 128, 512, 1, 1), (8, 50, 84, 256, 1024, 1, 1),
     (8, 200, 336, 256, 256, 1, 2), (8, 100, 168, 512, 128, 1, 0)]
out = []
for (N, H, W, C, K, st, rm) in L:
    torch.manual_seed(0)
    x = torch.randn(N, H, W, C, device=d)
    w = torch.randn(K, C, 1, 1, device=d) * (2.0 / C) ** 0.5
    pc = k.pack_conv(w, stride=st)
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    res = None
    if rm == 1: res = torch.randn(N, Ho, Wo, K, device=d)
    if rm == 2: res = torch.randn(N, Ho // 2, Wo // 2, K, device=d)
    y = torch.empty(N, Ho, Wo, K, device=d)
    for _ in range(3): k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=rm, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=rm, out=y)
    e1.record(); torch.cuda.synchronize()
    out.append("%.3f" % (e0.elapsed_time(e1) / 10))
print("ablate=%s: " % os.environ.get("LVC_PW_ABLATE", "0") + "  ".join(out))
