"""Stage timeline of the fused bottleneck (build/variants/bn_TL.so = conv_bneck.hip with -DBN_DIAG_TIMELINE; run with LVC_AMD_LIB set):
wave 0 of workgroups 0, grid/2, 8, grid/2 + 8 stamps s_memtime before the wait, after it and after the barrier of every stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
proj = len(sys.argv) > 1 and sys.argv[1] == "proj"
cin = 64 if proj else 256
def bn(c): return (torch.rand(c, device=d) + 0.5, torch.randn(c, device=d) * 0.1, torch.randn(c, device=d) * 0.1, torch.rand(c, device=d) + 0.5)
w1 = torch.randn(64, cin, 1, 1, device=d) * (2.0 / cin) ** 0.5
w2 = torch.randn(64, 64, 3, 3, device=d) * (2.0 / 576) ** 0.5
w3 = torch.randn(256, 128 if proj else 64, 1, 1, device=d) * (2.0 / 64) ** 0.5
bk = k.pack_bottleneck(k.pack_conv(w1, bn=bn(64)), k.pack_conv(w2, bn=bn(64), pad=1), k.pack_conv(w3, bn=None if proj else bn(256)), proj)
x = torch.randn(8, 200, 336, cin, device=d).relu_()
for _ in range(3): k.bottleneck_fused(x, bk)
torch.cuda.synchronize()
ws = k.conv_workspace(d)
ws[: 4 * 6 * 48 * 32].zero_()
k.bottleneck_fused(x, bk)
torch.cuda.synchronize()
t = ws[: 4 * 6 * 48 * 32].view(torch.int64).view(4, 6, 48, 4).cpu()
nst = (2 if proj else 8) + 18 + (16 if proj else 8)
for slot in range(4):
    hw = int(t[slot, 0, 0, 3])
    print("workgroup slot %d: HW_ID %#x (cu %d sh %d se %d)" % (slot, hw, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7))
    for it in (2, 3):
        r = t[slot, it, :nst]
        base = int(r[0, 0])
        nxt = int(t[slot, it + 1, 0, 0])
        print("  tile %d: total %d ticks (100 MHz: %.1f us)" % (it, nxt - base, (nxt - base) / 100.0))
        print("   stage: start  wait  barrier  body   (ticks of 10 ns)")
        for s_ in range(nst):
            end = int(r[s_ + 1, 0]) if s_ + 1 < nst else nxt
            print("   %2d %6d %5d %5d %5d" % (s_, int(r[s_, 0]) - base, int(r[s_, 1] - r[s_, 0]), int(r[s_, 2] - r[s_, 1]), end - int(r[s_, 2])))
