"""R101-FPN inference throughput at the benchmark's batch (8 x 3x800x1333), for the record (the bench metric is R50-FPN)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.evaluation import PipelinedInference
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
dev = torch.device("cuda:0")
model = build_model(base_rcnn_fpn(depth=101)).eval()
syn.conditioned_r50_fpn_(model, depth=101)
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]
for _ in range(3):
    with torch.no_grad(): model.inference_batched(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    with torch.no_grad(): out = model.inference_batched(batch)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
pipe = PipelinedInference(model, 2)
for _ in range(4): pipe.submit(batch, collectable=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): pipe.submit(batch, collectable=False)
torch.cuda.synchronize(); dp = time.perf_counter() - t0
pipe.synchronize()
print("R101-FPN: %.1f img/s (%.2f ms per batch of 8); two batches in flight: %.1f img/s; detections %s" % (160 / dt, dt / 20 * 1e3, 160 / dp, out[3].tolist()))
