import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from lvc_amd import kernels as k
from oracle import rcnn as orc
from test_gpu_boxes import _nhwc_rpn
D = "cuda:0"
g = torch.Generator().manual_seed(5)
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
N = 2
logits = [torch.randn(N, h * w * 3, generator=g) * 2 for h, w in shapes]
logits[0][:] = 0.5
logits[1][1] = -0.0
deltas = [torch.randn(N, h * w * 3, 4, generator=g) * 0.5 for h, w in shapes]
sizes = [(800, 1333), (750, 1344)]
cell = [orc.generate_cell_anchors((s,), (0.5, 1.0, 2.0)) for s in (32, 64, 128, 256, 512)]
anchors = orc.grid_anchors(cell, shapes, [4, 8, 16, 32, 64])
ref = orc.find_top_rpn_proposals(anchors, logits, deltas, sizes, 0.7, 1000, 1000)
fused = [t.to(D) for t in _nhwc_rpn(logits, deltas, shapes)]
boxes, olog, count = k.rpn_proposals([f[..., :3] for f in fused], [f[..., 3:] for f in fused], [c.to(D) for c in cell], [4, 8, 16, 32, 64], torch.tensor(sizes, dtype=torch.int32, device=D), 1000, 1000, 0.7)
for n in range(N):
    rb, rl = ref[n]
    d = (boxes[n, : len(rb)].cpu() - rb).abs().max(dim=1)[0]
    bad = (d > 1e-4).nonzero().view(-1)
    print("image", n, "count", int(count[n]), len(rb), "mismatching rows", bad[:10].tolist(), "of", len(bad))
    for r in bad[:3].tolist():
        print("  row", r, "score", float(rl[r]), float(olog[n, r]), "ref", rb[r].tolist(), "got", boxes[n, r].cpu().tolist())
    # is it a permutation among equal scores?
    if len(bad):
        r0 = bad[0].item(); s0 = rl[r0]
        sel = (rl == s0).nonzero().view(-1)
        a = {tuple(round(v, 2) for v in rb[i].tolist()) for i in sel.tolist()}
        b = {tuple(round(v, 2) for v in boxes[n, i].cpu().tolist()) for i in sel.tolist()}
        print("  equal-score group size", len(sel), "same set:", a == b, "only in ref", len(a - b), "only in got", len(b - a))
