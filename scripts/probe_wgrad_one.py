"""One weight-gradient launch shape (3x3 256 -> 256 on 8 x 100 x 168) for --pmc passes.  WGRAD_SPLIT=f16x2 selects the fp16 form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K
d = torch.device("cuda:0")
x = torch.randn(8, 100, 168, 256, device=d)
dy = torch.randn(8, 100, 168, 256, device=d)
for _ in range(4):
    K.conv_wgrad(x, dy, None, 3, 3, 1, 1, split=os.environ.get("WGRAD_SPLIT"))
torch.cuda.synchronize()
