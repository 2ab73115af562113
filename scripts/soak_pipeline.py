"""Soak: 600 batches in flight on two and three HIP streams (stream-K workers of different launches sharing the chip)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.evaluation import PipelinedInference
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
model = build_model(base_rcnn_fpn()).eval()
syn.conditioned_r50_fpn_(model)
dev = torch.device("cuda:0")
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]
for depth in (2, 3):
    pipe = PipelinedInference(model, depth)
    t0 = time.perf_counter()
    last = None
    for i in range(599):
        pipe.submit(batch, collectable=False)
    out = pipe.collect(pipe.submit(batch))
    pipe.synchronize()
    print("depth %d: 600 batches, %.1f img/s, last batch detections %s" % (depth, 4800 / (time.perf_counter() - t0), [len(o["instances"]) for o in out]), flush=True)
