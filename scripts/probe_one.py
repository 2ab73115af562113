"""Run one conv shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
N, C, H, W, K, R, s, p = [int(v) for v in sys.argv[1:9]]
x = torch.randn(N, H, W, C, device=d)
w = torch.randn(K, C, R, R, device=d) * 0.02
pc = k.pack_conv(w, stride=s, pad=p)
y = k.conv2d_nhwc(x, pc, relu=True)
for _ in range(3):
    k.conv2d_nhwc(x, pc, relu=True, out=y)
torch.cuda.synchronize()
