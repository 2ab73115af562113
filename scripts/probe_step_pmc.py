"""One bench step (R50-FPN, batch 8, 800x1333) and one backbone-only pass, each between marker launches, for rocprofv3 --pmc passes
(bench.py `_live_step_pmc`): rows between markers 1-2 = the whole step, 2-3 = the backbone (stem, res2-5, FPN) alone.
The marker is lvc_gelu on four floats (kernel name contains "gelu"; no launch of the detector uses it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import _lib, kernels as K
from lvc_amd._lib import c_longlong, ptr
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = build_model(base_rcnn_fpn(depth=50, num_classes=80, device="cuda:0")).eval()
syn.conditioned_r50_fpn_(model)
batch = [{"image": syn.synthetic_image(1 + i % 16).to(dev), "height": 800, "width": 1333} for i in range(B)]
mark = torch.zeros(4, device=dev)


def marker():
    _lib.check(_lib.lib().lvc_gelu(ptr(mark), ptr(mark), c_longlong(4), K._stream(mark)), "lvc_gelu")


with torch.no_grad():
    for _ in range(2):
        model.inference_batched(batch)
    images = model.preprocess_image(batch)
    model.backbone(images.tensor)
    torch.cuda.synchronize()
    marker()
    model.inference_batched(batch)
    marker()
    model.backbone(images.tensor)
    marker()
torch.cuda.synchronize()
