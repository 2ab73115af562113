"""Pointwise layers on the library in use (LVC_AMD_LIB selects a diagnostics build): time per launch.  usage: python scripts/probe_pw_diag.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, M, C, N, two in (("fc1", 8000, 12544, 1024, True), ("res4 conv1", 33600, 1024, 256, False), ("res4 conv3", 33600, 256, 1024, False), ("lat2", 537600, 256, 256, False)):
    x = torch.randn(M, C, generator=g).to(d)
    w = (torch.randn(N, C, generator=g) * 0.01).to(d)
    pc = K.pack_linear(w, None, two_acc=two)
    for _ in range(3): K.linear(x, pc, relu=True)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): K.linear(x, pc, relu=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    print(name, "ms", round(sorted(ts)[2], 4))
