"""PCIe-inclusive rate of the detector: inputs start in (pinned) host memory every step.
 a) the reference's mapper output: float32 CHW 3x800x1333 per image (12.8 MB)
 b) uint8 CHW 3x800x1333 (3.2 MB)
 c) raw file pixels uint8 HWC 480x640 (0.92 MB), ResizeShortestEdge on the device"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
model = build_model(base_rcnn_fpn()).eval()
syn.conditioned_r50_fpn_(model)
dev = torch.device("cuda:0")
f32 = [syn.synthetic_image(1 + i).pin_memory() for i in range(8)]
u8 = [t.round().clamp(0, 255).to(torch.uint8).pin_memory() for t in f32]
raw = [syn.synthetic_image(1 + i, 480, 640).permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).contiguous().pin_memory() for i in range(8)]
res = [t.to(dev) for t in f32]
def run(batch, n=20):
    for _ in range(3):
        with torch.no_grad(): model.inference_batched(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        with torch.no_grad(): model.inference_batched(batch)
    torch.cuda.synchronize()
    return 8 * n / (time.perf_counter() - t0)
print("resident float CHW           %.1f img/s" % run([{"image": t, "height": 800, "width": 1333} for t in res]))
print("host float32 CHW 800x1333    %.1f img/s" % run([{"image": t, "height": 800, "width": 1333} for t in f32]))
print("host uint8 CHW 800x1333      %.1f img/s" % run([{"image": t, "height": 800, "width": 1333} for t in u8]))
print("host raw uint8 HWC 480x640   %.1f img/s" % run([{"raw": t} for t in raw]))
