"""A whole res2 bottleneck as ONE launch (csrc/conv_bneck.hip) against the round-5 launches (conv1, conv2, conv3 + residual): results
(both vs an fp64 evaluation) and time per block at batch 8 of R50-FPN (8 x 200 x 336)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)


def bn(c):
    return (torch.rand(c, device=d) + 0.5, torch.randn(c, device=d) * 0.1, torch.randn(c, device=d) * 0.1, torch.rand(c, device=d) + 0.5)


def run(name, N, H, W, cin, proj, timed=True):
    w1 = torch.randn(64, cin, 1, 1, device=d) * (2.0 / cin) ** 0.5
    w2 = torch.randn(64, 64, 3, 3, device=d) * (2.0 / 576) ** 0.5
    w3 = torch.randn(256, 64, 1, 1, device=d) * (2.0 / 64) ** 0.5
    ws = torch.randn(256, cin, 1, 1, device=d) * (2.0 / cin) ** 0.5
    b1, b2, b3, bs = bn(64), bn(64), bn(256), bn(256)
    p1, p2, p3 = k.pack_conv(w1, bn=b1), k.pack_conv(w2, bn=b2, pad=1), k.pack_conv(w3, bn=b3)
    a3, asc = k.conv_affine(None, b3), k.conv_affine(None, bs)
    if proj:
        pf = k.pack_conv(torch.cat([w3 * a3[0].view(-1, 1, 1, 1), ws * asc[0].view(-1, 1, 1, 1)], 1).contiguous(), affine=(None, a3[1] + asc[1]))
        pss = k.pack_conv(ws, bn=bs)
    bk = k.pack_bottleneck(p1, p2, pf if proj else p3, proj)
    xs = [torch.randn(N, H, W, cin, device=d).relu_() for _ in range(3)]
    y = torch.empty(N, H, W, 256, device=d)
    z = torch.empty_like(y)

    def fused(i):
        return k.bottleneck_fused(xs[i % 3], bk, out=y)

    def separate(i):
        x = xs[i % 3]
        t = k.conv2d_nhwc(x, p1, relu=True)
        t = k.conv2d_nhwc(t, p2, relu=True)
        sc = k.conv2d_nhwc(x, pss) if proj else x
        return k.conv2d_nhwc(t, p3, relu=True, residual=sc, res_mode=1, out=z)

    def timeit(f):
        for i in range(4): f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): f(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20

    tf = ts = float("nan")
    if timed == "fused":
        print("%-28s fused %.4f ms" % (name, timeit(fused)), flush=True)
        return 0.0
    if timed:
        tf, ts = timeit(fused), timeit(separate)
    fused(0); separate(0)
    torch.cuda.synchronize()
    # fp64 evaluation of image 0 (whole image: every tile edge and the image border are in it)
    x0 = xs[0][:1].permute(0, 3, 1, 2).double()

    def cb(x, w, b, pad=0):
        s, t = k.conv_affine(None, b)
        return F.conv2d(x, w.double(), padding=pad) * s.double().view(1, -1, 1, 1) + t.double().view(1, -1, 1, 1)

    r = cb(cb(cb(x0, w1, b1).relu(), w2, b2, 1).relu(), w3, b3) + (cb(x0, ws, bs) if proj else x0)
    r = r.relu().permute(0, 2, 3, 1)
    scale = float(r.abs().max())
    ef = (y[:1].double() - r).abs()
    es = (z[:1].double() - r).abs()
    M = N * H * W
    gb = M * 4 * (cin + 256) / 1e9
    gf = 2.0 * M * (cin * 64 + 576 * 64 + 64 * 256 + (cin * 256 if proj else 0)) / 1e9
    print("%-28s fused %.4f ms (%.2f TB/s of %.2f GB, %.0f TF/s) | separate %.4f ms | err/scale fused max %.2e rms %.2e, separate max %.2e rms %.2e | max|fused-sep| %.1e"
          % (name, tf, gb / tf, gb, gf / tf, ts, float(ef.max()) / scale, float(ef.pow(2).mean().sqrt()) / scale, float(es.max()) / scale,
             float(es.pow(2).mean().sqrt()) / scale, float((y - z).abs().max())), flush=True)
    return float(ef.max()) / scale


if len(sys.argv) > 1 and sys.argv[1] == "small":
    for (N, H, W) in [(1, 4, 32), (1, 5, 33), (2, 13, 70), (1, 50, 84), (3, 9, 31)]:
        run("identity %dx%dx%d" % (N, H, W), N, H, W, 256, False, timed=False)
        run("projection %dx%dx%d" % (N, H, W), N, H, W, 64, True, timed=False)
elif len(sys.argv) > 1 and sys.argv[1] == "time":
    run("res2.1 / res2.2 (256->64->256)", 8, 200, 336, 256, False, timed="fused")
    run("res2.0 (64->64->256, proj)", 8, 200, 336, 64, True, timed="fused")
else:
    run("res2.1 / res2.2 (256->64->256)", 8, 200, 336, 256, False)
    run("res2.0 (64->64->256, proj)", 8, 200, 336, 64, True)
print("conv error word", k.conv_error_word(d))
