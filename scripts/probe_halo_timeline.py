"""Where a wave of the pipelined 3x3 kernel spends its cycles (diagnostics build: csrc/conv3x3_halo_s1.hip compiled with -DHALO_TIMELINE
and linked as the library -- see the header of this script's caller): cycles between barriers (MFMA phases), in the vmcnt wait in front
of a barrier (weight DMA / halo loads not landed), in the barrier itself (waiting for the slowest wave), prologue and epilogue per tile.
usage: python scripts/probe_halo_timeline.py N C H W K [two_acc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
N, C, H, W, K = [int(v) for v in sys.argv[1:6]]
two = len(sys.argv) > 6 and sys.argv[6] == "1"
torch.manual_seed(0)
x = torch.randn(N, H, W, C, device=d)
w = torch.randn(K, C, 3, 3, device=d) * 0.02
pc = k.pack_conv(w, stride=1, pad=1)
pc.two_acc = two
y = k.conv2d_nhwc(x, pc, relu=True)
for _ in range(3): k.conv2d_nhwc(x, pc, relu=True, out=y)
torch.cuda.synchronize()
ws = k.conv_workspace(d)
off = 512 * 256 * 128 * 4
dbg = ws[off: off + 64 * 8 * 16 * 8].view(torch.int64)
dbg.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); k.conv2d_nhwc(x, pc, relu=True, out=y); e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
t = dbg.cpu().view(64, 8, 16).double()
ok = t[:, :, 7] == 1
print("%d x %d x %d x %d -> %d, %s accumulator(s): %.1f us; stamped workgroups %d" % (N, H, W, C, K, "two" if two else "one", ms * 1e3, int(ok[:, 0].sum())))
for wv in (0, 3, 7):
    r = t[ok[:, wv], wv]
    tot = r[:, 6].mean()
    print("  wave %d (mean over workgroups): total %.0f ticks = 100 %%; between barriers %.1f %%, vmcnt wait %.1f %%, barrier %.1f %%, prologue %.1f %%, tail %.1f %% (stream-K hand-off %.1f, accumulators through LDS %.1f, output rows %.1f); %.0f taps, per tap: run %.0f, vm %.0f, barrier %.0f ticks" % (
        wv, tot, 100 * r[:, 0].mean() / tot, 100 * r[:, 1].mean() / tot, 100 * r[:, 2].mean() / tot, 100 * r[:, 4].mean() / tot, 100 * r[:, 5].mean() / tot,
        100 * r[:, 8].mean() / tot, 100 * r[:, 9].mean() / tot, 100 * (r[:, 5] - r[:, 8] - r[:, 9]).mean() / tot,
        r[:, 3].mean(), (r[:, 0] / r[:, 3]).mean(), (r[:, 1] / r[:, 3]).mean(), (r[:, 2] / r[:, 3]).mean()))
print("  ticks per us (total of the longest workgroup / event time): %.0f" % (t[:, :, 6].max() / (ms * 1e3)))
