"""Descriptor network (ViT-S/8, 224x224 crops) throughput: crops/s at batch 64 with the matrix-core attention and with the scalar one,
and the attention kernels alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K
from lvc_amd.modeling.vit import seeded_state_dict_, vit_small
d = torch.device("cuda:0")
m = vit_small(8)
seeded_state_dict_(m, 0)
m = m.to(d).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 3, 224, 224, device=d)
for mode in (True, False):
    K.MHA_MFMA = mode
    with torch.no_grad():
        for _ in range(2): y = m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 5 if mode else 2
        for _ in range(n): y = m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("ViT-S/8 batch %d, attention %s: %.1f ms per batch, %.0f crops/s (%.1f TF/s of 45.2 GFLOP per crop)" % (B, "mfma" if mode else "valu", dt * 1e3, B / dt, 45.2e9 * B / dt / 1e12), flush=True)
qkv = torch.randn(B * 785, 1152, device=d)
for mode in (True, False):
    for _ in range(2): K.mha(qkv, B, 785, 6, 64, 0.125, mfma=mode)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): K.mha(qkv, B, 785, 6, 64, 0.125, mfma=mode)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 4.0 * B * 6 * 785 * 785 * 64
    print("attention alone (%s): %.3f ms per layer, %.1f TF/s" % ("mfma" if mode else "valu", ms, fl / ms / 1e9))
