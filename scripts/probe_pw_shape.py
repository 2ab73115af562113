"""Run one pointwise layer a few times (for the rocprofv3 --pmc passes of scripts/probe_pwdma_pmc.sh): N H W C K stride res_mode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
N, H, W, C, K, s, rm = [int(v) for v in sys.argv[1:8]]
torch.manual_seed(0)
x = torch.randn(N, H, W, C, device=d)
w = torch.randn(K, C, 1, 1, device=d) * (2.0 / C) ** 0.5
pc = k.pack_conv(w, stride=s)
Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
res = None
if rm == 1: res = torch.randn(N, Ho, Wo, K, device=d)
if rm == 2: res = torch.randn(N, Ho // 2, Wo // 2, K, device=d)
y = torch.empty(N, Ho, Wo, K, device=d)
for _ in range(4):
    k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=rm, out=y)
torch.cuda.synchronize()
