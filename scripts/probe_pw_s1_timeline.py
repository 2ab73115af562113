"""Where a wave of the pipelined pointwise kernel spends its cycles (diagnostics build: csrc/conv_pw_s1.hip compiled with -DPW_TIMELINE
and linked as the library): a tile's prologue (first chunks requested, split, barrier), its chunk loop, the stream-K hand-off, the
accumulators' trip through LDS, the output rows.
usage: python scripts/probe_pw_s1_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
LAYERS = [("res2 c1 256>64", 8, 200, 336, 256, 64, 0), ("res2 c3 64>256+r", 8, 200, 336, 64, 256, 1), ("res3 c1 512>128", 8, 100, 168, 512, 128, 0),
          ("res3 c3 128>512+r", 8, 100, 168, 128, 512, 1), ("res4 c1 1024>256", 8, 50, 84, 1024, 256, 0), ("res4 c3 256>1024+r", 8, 50, 84, 256, 1024, 1),
          ("res5 c1 2048>512", 8, 25, 42, 2048, 512, 0), ("res5 c3 512>2048+r", 8, 25, 42, 512, 2048, 1), ("fpn lat2 256>256+up", 8, 200, 336, 256, 256, 2),
          ("fc1 12544>1024", 8000, 1, 1, 12544, 1024, 0), ("vit qkv 384>1152", 50240, 1, 1, 384, 1152, 0), ("vit fc2 1536>384", 50240, 1, 1, 1536, 384, 0)]
for name, N, H, W, C, K, rm in LAYERS:
    x = torch.randn(N, H, W, C, device=d)
    w = torch.randn(K, C, 1, 1, device=d) * (2.0 / C) ** 0.5
    pc = k.pack_conv(w)
    res = None
    if rm == 1: res = torch.randn(N, H, W, K, device=d)
    if rm == 2: res = torch.randn(N, H // 2, W // 2, K, device=d)
    y = torch.empty(N, H, W, K, device=d)
    f = lambda: k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=rm, out=y)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ws = k.conv_workspace(d)
    off = 512 * 256 * 128 * 4
    dbg = ws[off: off + 64 * 8 * 16 * 8].view(torch.int64)
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = dbg.cpu().view(64, 8, 16).double()
    ok = t[:, 0, 8] == 1
    if int(ok.sum()) == 0:
        print("%-22s %.1f us: not on the pipelined kernel" % (name, ms * 1e3)); continue
    r = t[ok, 0]
    r7 = t[ok, 7]
    tot = r[:, 7].mean()
    print("%-22s %7.1f us; wave 0 of %2d workgroups: %.1f tiles, %.1f chunks each; prologue %.1f %%, chunk loop %.1f %% (%.0f ticks / chunk), hand-off %.1f %%, accumulators through LDS %.1f %%, output rows %.1f %%; per chunk, wave 0 / wave 7: wait for the activation registers %.0f / %.0f, vmcnt %.0f / %.0f, barrier %.0f / %.0f ticks; ticks/us %.0f" % (
        name, ms * 1e3, int(ok.sum()), r[:, 5].mean(), r[:, 6].mean(), 100 * r[:, 0].mean() / tot, 100 * r[:, 1].mean() / tot, (r[:, 1] / r[:, 6]).mean(),
        100 * r[:, 2].mean() / tot, 100 * r[:, 3].mean() / tot, 100 * r[:, 4].mean() / tot,
        (r[:, 9] / r[:, 6]).mean(), (r7[:, 9] / r7[:, 6]).mean(), (r[:, 10] / r[:, 6]).mean(), (r7[:, 10] / r7[:, 6]).mean(), (r[:, 11] / r[:, 6]).mean(), (r7[:, 11] / r7[:, 6]).mean(),
        t[:, :, 7].max() / (ms * 1e3)))
