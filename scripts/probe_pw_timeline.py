"""Timeline of the LDS-DMA pointwise kernel (LVC_PW_TIMELINE=1: s_memtime stamps of every 64th workgroup, written to the unused upper
half of the partial-tile workspace): where a worker's time goes -- prologue, chunks, partial hand-off, epilogue.
usage: LVC_PW_TIMELINE=1 python scripts/probe_pw_timeline.py"""
import os, sys
os.environ["LVC_PW_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
TAG = {1: "entry", 2: "first chunks issued", 3: "chunk top passed", 4: "K loop of a segment done", 5: "partials handed off", 6: "tile complete (partials added)",
       7: "residual chunks done", 8: "drained", 9: "exit", 10: "stores issued"}
LAYERS = [("res4 c1 1024>256", 8, 50, 84, 1024, 256, 0), ("res4 c3 256>1024+r", 8, 50, 84, 256, 1024, 1), ("res3 c3 128>512+r", 8, 100, 168, 128, 512, 1),
          ("res2 c3 64>256+r", 8, 200, 336, 64, 256, 1), ("res5 c1 2048>512", 8, 25, 42, 2048, 512, 0), ("fc1 12544>1024", 8000, 1, 1, 12544, 1024, 0)]
for name, N, H, W, C, K, rm in LAYERS:
    x = torch.randn(N, H, W, C, device=d)
    w = torch.randn(K, C, 1, 1, device=d) * (2.0 / C) ** 0.5
    pc = k.pack_conv(w)
    res = torch.randn(N, H, W, K, device=d) if rm else None
    y = torch.empty(N, H, W, K, device=d)
    for _ in range(3): k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=rm, out=y)
    torch.cuda.synchronize()
    ws = k.conv_workspace(d)
    dbg = ws[512 * 256 * 128 * 4: 512 * 256 * 128 * 4 + 4 * 512 * 8].view(torch.int64)
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); k.conv2d_nhwc(x, pc, relu=True, residual=res, res_mode=rm, out=y); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = dbg.cpu().view(4, 256, 2)
    span = max([int(t[wg, int(t[wg, 0, 0]) - 1, 0]) - int(t[wg, 1, 0]) for wg in range(4) if int(t[wg, 0, 0]) >= 3] or [1])
    tpu = span / (ms * 1e3)      # ticks per microsecond, calibrated on the longest stamped workgroup ~ the launch
    print("== %s: %.1f us (event-timed, with stamps); %.0f ticks/us" % (name, ms * 1e3, tpu))
    for wg in range(4):
        n = int(t[wg, 0, 0])
        if n < 2: continue
        ev = [(int(t[wg, i, 0]), int(t[wg, i, 1])) for i in range(1, n)]
        t0 = ev[0][0]
        # compress runs of chunk tops
        line, i = [], 0
        while i < len(ev):
            ts, tag = ev[i]
            if tag == 3:
                j = i
                while j + 1 < len(ev) and ev[j + 1][1] == 3: j += 1
                cnt = j - i + 1
                line.append("%d chunks [%.2f..%.2f us, %.3f us/chunk]" % (cnt, (ts - t0) / tpu, (ev[j][0] - t0) / tpu, (ev[j][0] - ts) / tpu / max(1, cnt - 1)))
                i = j + 1
            else:
                line.append("%s @%.2f" % (TAG.get(tag, str(tag)), (ts - t0) / tpu))
                i += 1
        print("  wg %3d: %s" % (wg * 64, " | ".join(line)))
