"""Inter-kernel gaps by transition type (run under rocprofv3 --kernel-trace; scripts/probe_gaps.sh prints the medians)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
N, H, W = 8, 50, 84
x256 = torch.randn(N, H, W, 256, device=d)
x1024 = torch.randn(N, H, W, 1024, device=d)
w33 = k.pack_conv(torch.randn(256, 256, 3, 3, device=d) * 0.02, pad=1)           # halo<2>
w33s = k.pack_conv(torch.randn(64, 64, 3, 3, device=d) * 0.05, pad=1)            # halo<1>
x64 = torch.randn(N, 200, 336, 64, device=d)
wc3 = k.pack_conv(torch.randn(1024, 256, 1, 1, device=d) * 0.05)                 # pw<4>
wc1 = k.pack_conv(torch.randn(256, 1024, 1, 1, device=d) * 0.03)                 # pw<4>
y256 = torch.empty(N, H, W, 256, device=d); y1024 = torch.empty(N, H, W, 1024, device=d); y64 = torch.empty_like(x64)
for rep in range(30):
    k.conv2d_nhwc(x1024, wc1, relu=True, out=y256)      # pw
    k.conv2d_nhwc(y256, w33, relu=True, out=x256)       # halo<2>
    k.conv2d_nhwc(x256, wc3, relu=True, out=y1024)      # pw
    k.conv2d_nhwc(y1024, wc1, relu=True, out=y256)      # pw (after pw)
    k.conv2d_nhwc(y256, w33, relu=True, out=x256)       # halo<2>
    k.conv2d_nhwc(x256, w33, relu=True, out=y256)       # halo<2> after halo<2>
    k.conv2d_nhwc(x64, w33s, relu=True, out=y64)        # halo<1> after halo<2>
    k.conv2d_nhwc(y64, w33s, relu=True, out=x64)        # halo<1> after halo<1>
torch.cuda.synchronize()
