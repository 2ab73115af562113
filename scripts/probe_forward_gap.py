"""Where the drop-in forward() loses time against inference_batched (bench `value` vs `value_inference_batched`): host-side pieces of one
step, timed around a synchronised GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.modeling.roi_heads.roi_heads import instances_from_batched
from lvc_amd.utils import synthetic as syn
dev = torch.device("cuda:0")
model = build_model(base_rcnn_fpn(depth=50, num_classes=80, device="cuda:0")).eval()
syn.conditioned_r50_fpn_(model)
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]
with torch.no_grad():
    for _ in range(5):
        model(batch)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        model.inference_batched(batch)
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        model(batch)
    torch.cuda.synchronize()
    tf = (time.perf_counter() - t0) / n
    # host time of issuing one step (GPU idle before): how long until all launches are queued
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.inference_batched(batch)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # the read + slicing alone (GPU idle)
    ob, osc, ocl, cnt, status = out
    sizes = [(800, 1333)] * 8
    t0 = time.perf_counter()
    for _ in range(50):
        instances_from_batched(ob, osc, ocl, cnt, sizes, status)
    t_inst = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50):
        torch.cat([cnt, status, K.range_summary(dev)]).tolist()
    t_read = (time.perf_counter() - t0) / 50
print("per step: inference_batched loop %.3f ms | forward loop %.3f ms (gap %.3f) | host issue of one step %.3f ms, step from idle %.3f ms | "
      "instances_from_batched alone %.3f ms of which the read %.3f ms" % (1e3 * tb, 1e3 * tf, 1e3 * (tf - tb), 1e3 * t_issue, 1e3 * t_all, 1e3 * t_inst, 1e3 * t_read))
