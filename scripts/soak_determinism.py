"""Soak: the same batch through forward() 300 times -- every run's detections must equal the first run's bit for bit (the kernels
that take work in a data-dependent or atomics-given order -- the ROIAlign work order, stream-K hand-offs, the grouped launches --
decide when a value is computed, never what it is), and the kNN sweep 100 times on one input."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd import label_verification as LV
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
model = build_model(base_rcnn_fpn()).eval()
syn.conditioned_r50_fpn_(model)
dev = torch.device("cuda:0")
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
with torch.no_grad():
    ref = None
    bad = 0
    t0 = time.perf_counter()
    for it in range(N):
        out = model(batch)
        cur = [(o["instances"].pred_boxes.tensor.clone(), o["instances"].scores.clone(), o["instances"].pred_classes.clone()) for o in out]
        if ref is None:
            ref = cur
        else:
            bad += int(not all(torch.equal(a, b) for r, c in zip(ref, cur) for a, b in zip(r, c)))
    print("forward x %d: %d runs differ from the first (%.1f img/s incl. the comparisons)" % (N, bad, 8 * N / (time.perf_counter() - t0)))
g = torch.Generator().manual_seed(0)
Q, S, D = 120000, 2400, 1024
classes = torch.arange(80).repeat_interleave(30).to(dev)
shots, q = torch.randn(S, D, generator=g).to(dev), torch.randn(Q, D, generator=g).to(dev)
det = torch.randint(0, 80, (Q,), generator=g).to(dev)
top0, keep0 = LV.knn_sweep(classes, shots, q, det, 10, True)
bad = 0
for it in range(100):
    top, keep = LV.knn_sweep(classes, shots, q, det, 10, True)
    bad += int(not (torch.equal(top, top0) and torch.equal(keep, keep0)))
print("kNN sweep x 100: %d runs differ from the first" % bad)
