"""One memory-bound 1x1 layer (64 -> 256 with residual, batch 8 res2 map) and a torch add of the same size, for --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
x = torch.randn(8, 200, 336, 64, device=d)
w = torch.randn(256, 64, 1, 1, device=d) * 0.1
pc = k.pack_conv(w)
res = torch.randn(8, 200, 336, 256, device=d)
y = torch.empty_like(res)
for _ in range(4):
    k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=1, out=y)
torch.cuda.synchronize()
for _ in range(4):
    torch.add(res, res, out=y)
torch.cuda.synchronize()
