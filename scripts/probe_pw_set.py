"""The pointwise layers of R50-FPN at batch 8 (800x1333 -> 200x336 at res2), each timed alone: ms per launch; env switches
(LVC_PW_PF, ...) give the A/B.  Layers cycle through 3 distinct input buffers so that a layer does not find its input in L2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
LAYERS = [("res2 c1 256>64", 8, 200, 336, 256, 64, 0), ("res2 c3 64>256+r", 8, 200, 336, 64, 256, 1),
          ("res3 c1 512>128", 8, 100, 168, 512, 128, 0), ("res3 c3 128>512+r", 8, 100, 168, 128, 512, 1),
          ("res4 c1 1024>256", 8, 50, 84, 1024, 256, 0), ("res4 c3 256>1024+r", 8, 50, 84, 256, 1024, 1),
          ("res5 c1 2048>512", 8, 25, 42, 2048, 512, 0), ("res5 c3 512>2048+r", 8, 25, 42, 512, 2048, 1),
          ("fpn lat2 256>256+up", 8, 200, 336, 256, 256, 2), ("fpn lat3 512>256+up", 8, 100, 168, 512, 256, 2),
          ("fc1 12544>1024", 8000, 1, 1, 12544, 1024, 0), ("fc2 1024>1024", 8000, 1, 1, 1024, 1024, 0)]
LAYERS += [("res4 c1s2 1024>512?", 8, 50, 84, 512, 256, 0), ("vit qkv 384>1152", 50240, 1, 1, 384, 1152, 0), ("vit fc1 384>1536", 50240, 1, 1, 384, 1536, 0),
           ("vit fc2 1536>384", 50240, 1, 1, 1536, 384, 0)]
MODES = [int(v) for v in os.environ.get("PW_S1_MODES", "0,1,2,3").split(",")]     # 3 = one accumulator on the 256 x 256 tile where it applies (csrc/conv_pw_w2.hip)
tot = {m: 0.0 for m in MODES}
k._PW_S1_MIN_C = int(os.environ.get("LVC_PW_S1_MIN_C", "64"))
for name, N, H, W, C, K, rm in LAYERS:
    xs = [torch.randn(N, H, W, C, device=d) for _ in range(3)]
    w = torch.randn(K, C, 1, 1, device=d) * (2.0 / C) ** 0.5
    pc = k.pack_conv(w)
    res = None
    if rm == 1: res = [torch.randn(N, H, W, K, device=d) for _ in range(3)]
    if rm == 2: res = [torch.randn(N, H // 2, W // 2, K, device=d) for _ in range(3)]
    y = torch.empty(N, H, W, K, device=d)
    line = "%-22s" % name
    ref = exact = None
    for mode in MODES:
        k.PW_S1 = min(mode, 2)
        k.PW_W2 = mode == 3
        f = lambda i: k.conv2d_nhwc(xs[i % 3], pc, relu=True, residual=res[i % 3] if res else None, res_mode=rm, out=y)
        for i in range(6): f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): f(i)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        tot[mode] += ms
        out = f(0).clone()
        if ref is None: ref = out
        if exact is None and os.environ.get("PW_FP64", "1") == "1":      # the layer in fp64 on a 4096-row sample: error of each form against the exact result
            xi = xs[0].reshape(-1, C)[:4096].double()
            e = xi @ w.reshape(K, C).double().t()
            if rm == 1: e = e + res[0].reshape(-1, K)[:4096].double()
            if rm == 2:
                idx = torch.arange(4096, device=d)
                hh, ww = (idx // W) % H, idx % W
                e = e + res[0][idx // (H * W), hh // 2, ww // 2].double()
            exact = e.clamp(min=0)
        err = float((out.reshape(-1, K)[:4096].double() - exact).abs().max()) / float(exact.abs().max()) if exact is not None else float("nan")
        line += " | mode %d: %.4f ms %6.1f TF/s (vs mode %d: %.1e, vs fp64: %.1e)" % (mode, ms, 2.0 * N * H * W * C * K / ms / 1e9, MODES[0], float((out - ref).abs().max()) / float(ref.abs().max()), err)
    print(line, flush=True)
print("sum " + ", ".join("mode %d: %.4f ms" % (m, tot[m]) for m in MODES) + "   (mode 0 = LDS-DMA kernel, 1 = pipelined two-acc, 2 = pipelined one-acc, 3 = one-acc on the 256 x 256 tile); conv error word %d" % k.conv_error_word(d))
