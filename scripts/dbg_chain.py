import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
for (M, K1, N1, N2, has_res) in [(8*200*336, 64, 256, 64, True), (8*100*168, 128, 512, 128, True)]:
    x = torch.randn(M, 1, 1, K1, device=d).relu_()
    res = torch.randn(M, 1, 1, N1, device=d).relu_()
    wa = torch.randn(N1, K1, 1, 1, device=d) * (2.0 / K1) ** 0.5
    wb = torch.randn(N2, N1, 1, 1, device=d) * (2.0 / N1) ** 0.5
    pa, pb = k.pack_conv(wa), k.pack_conv(wb)
    ch = k.pack_chain(pa, pb)
    z1 = k.conv2d_nhwc(x, pa, relu=True, residual=res, res_mode=1)
    z2 = k.conv2d_nhwc(z1, pb, relu=True)
    for it in range(3):
        y1, y2 = k.conv1x1_chain(x, ch, residual=res)
        torch.cuda.synchronize()
        b1 = ((y1 - z1).abs().view(M, N1) > 1e-3)
        b2 = ((y2 - z2).abs().view(M, N2) > 1e-3)
        print("shape", (K1, N1, N2), "iter", it, "bad y1 elems", int(b1.sum()), "bad y2 elems", int(b2.sum()))
        for name, b, N in (("y1", b1, N1), ("y2", b2, N2)):
            if b.any():
                rows = b.any(1).nonzero().flatten()
                cols = b.any(0).nonzero().flatten()
                print("  ", name, "bad rows", rows.numel(), "first", rows[:12].tolist(), "row%32", sorted(set((rows % 32).tolist()))[:40], "group(128)", sorted(set((rows // 128).tolist()))[:10])
                print("  ", name, "bad cols", cols.numel(), cols[:40].tolist())
                r0 = int(rows[0])
                print("   row", r0, "bad cols in row:", b[r0].nonzero().flatten().tolist()[:64])
    # what do the wrong values look like?
    if b1.any():
        rows = b1.any(1).nonzero().flatten()[:6]
        y1v, z1v, rv = y1.view(M, N1), z1.view(M, N1), res.view(M, N1)
        for r in rows.tolist():
            cs = b1[r].nonzero().flatten().tolist()[:2]
            for c in cs:
                pre = float(z1v[r, c] - rv[r, c])
                print("   y1[%d,%d] = %.5f expected %.5f (res %.5f, conv part %.5f) diff %.5f | neighbours' res: %s" % (
                    r, c, float(y1v[r, c]), float(z1v[r, c]), float(rv[r, c]), pre, float(y1v[r, c] - z1v[r, c]),
                    ["%.4f" % float(rv[r + dr, c]) for dr in (-32, -16, -4, -1, 1, 4, 16, 32) if 0 <= r + dr < M]))
