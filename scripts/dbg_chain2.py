import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
for (M, K1, N1, N2, has_res) in [(8*200*336, 64, 256, 64, True), (8*200*336, 128, 256, 64, False), (8*100*168, 128, 512, 128, True)]:
    x = torch.randn(M, 1, 1, K1, device=d).relu_()
    res = torch.randn(M, 1, 1, N1, device=d).relu_() if has_res else None
    wa = torch.randn(N1, K1, 1, 1, device=d) * (2.0 / K1) ** 0.5
    wb = torch.randn(N2, N1, 1, 1, device=d) * (2.0 / N1) ** 0.5
    pa, pb = k.pack_conv(wa), k.pack_conv(wb)
    ch = k.pack_chain(pa, pb)
    print("before", k.conv_error_word(d))
    y1, y2 = k.conv1x1_chain(x, ch, residual=res)
    torch.cuda.synchronize()
    print((K1, N1, N2), "after fused", k.conv_error_word(d), "max |x|", float(x.abs().max()), "max |y1|", float(y1.abs().max()), "max y2", float(y2.abs().max()), "nan", bool(torch.isnan(y1).any()))
    k.clear_conv_error_word(d)
