"""conv3 (+ residual + ReLU) -> next conv1 (+ ReLU) as ONE launch (csrc/conv_pw_chain.hip) against the two separate launches:
results (both vs an fp64 evaluation on a row sample) and time per block boundary at batch 8 of R50-FPN."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [("res2 c3>c1 (64,256,64)", 8 * 200 * 336, 64, 64, 256, 64), ("res2.0 [c3|proj]>c1 (128,256,64)", 8 * 200 * 336, 128, 128, 256, 64),
          ("res3 c3>c1 (128,512,128)", 8 * 100 * 168, 128, 128, 512, 128)]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if sys.argv[1] in s[0]]
for name, M, ldx, K1, N1, N2 in SHAPES:
    xs = [torch.randn(M, 1, 1, ldx, device=d).relu_() for _ in range(3)]
    res = [torch.randn(M, 1, 1, N1, device=d).relu_() for _ in range(3)] if "proj" not in name else None
    wa = torch.randn(N1, K1, 1, 1, device=d) * (2.0 / K1) ** 0.5
    wb = torch.randn(N2, N1, 1, 1, device=d) * (2.0 / N1) ** 0.5
    bna = (torch.rand(N1, device=d) + 0.5, torch.randn(N1, device=d) * 0.1, torch.randn(N1, device=d) * 0.1, torch.rand(N1, device=d) + 0.5)
    bnb = (torch.rand(N2, device=d) + 0.5, torch.randn(N2, device=d) * 0.1, torch.randn(N2, device=d) * 0.1, torch.rand(N2, device=d) + 0.5)
    pa, pb = k.pack_conv(wa, bn=bna), k.pack_conv(wb, bn=bnb)
    ch = k.pack_chain(pa, pb)
    y1 = torch.empty(M, 1, 1, N1, device=d); y2 = torch.empty(M, 1, 1, N2, device=d)
    z1 = torch.empty_like(y1); z2 = torch.empty_like(y2)

    def fused(i):
        return k.conv1x1_chain(xs[i % 3], ch, residual=res[i % 3] if res else None, out1=y1, out2=y2)

    def separate(i):
        k.conv2d_nhwc(xs[i % 3], pa, relu=True, residual=res[i % 3] if res else None, res_mode=1 if res else 0, out=z1)
        k.conv2d_nhwc(z1, pb, relu=True, out=z2)

    def timeit(f):
        for i in range(6): f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): f(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 30

    tf, ts = timeit(fused), timeit(separate)
    fused(0); separate(0)
    torch.cuda.synchronize()
    # fp64 evaluation on a row sample
    idx = torch.randint(0, M, (4096,), device=d)
    idx[0] = M - 1
    xd = xs[0].view(M, ldx)[idx, :K1].double()
    sa, ta = k.conv_affine(None, bna); sb, tb = k.conv_affine(None, bnb)
    r1 = xd @ wa.view(N1, K1).double().t() * sa.double() + ta.double()
    if res: r1 = r1 + res[0].view(M, N1)[idx].double()
    r1 = r1.relu()
    r2 = (r1 @ wb.view(N2, N1).double().t() * sb.double() + tb.double()).relu()
    def err(a, r): return float((a.double() - r).pow(2).mean().sqrt() / r.abs().max())
    gb = M * 4 * (K1 + (N1 if res else 0) + N1 + N2) / 1e9
    print("%-36s fused %.4f ms (%.2f TB/s of %.2f GB) | separate %.4f ms | rms err / scale: y1 fused %.2e sep %.2e, y2 fused %.2e sep %.2e | max |fused - sep| y1 %.1e y2 %.1e"
          % (name, tf, gb / tf, gb, ts, err(y1.view(M, N1)[idx], r1), err(z1.view(M, N1)[idx], r1), err(y2.view(M, N2)[idx], r2), err(z2.view(M, N2)[idx], r2),
             float((y1 - z1).abs().max()), float((y2 - z2).abs().max())), flush=True)
print("conv error word", k.conv_error_word(d))
