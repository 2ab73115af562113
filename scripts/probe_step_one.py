"""A few detector steps (batch 8, 800x1333) and one kNN sweep, for --pmc passes over the non-conv kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.label_verification import knn_sweep
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
dev = torch.device("cuda:0")
model = build_model(base_rcnn_fpn()).eval()
syn.conditioned_r50_fpn_(model)
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]
for _ in range(3):
    with torch.no_grad():
        model.inference_batched(batch)
g = torch.Generator().manual_seed(0)
S, D, Q = 2400, 1024, 32768
shots = torch.randn(S, D, generator=g).to(dev)
q = torch.randn(Q, D, generator=g).to(dev)
cls = torch.arange(80).repeat_interleave(30).to(dev)
for _ in range(2):
    knn_sweep(cls, shots, q, None, 10, True)
torch.cuda.synchronize()
