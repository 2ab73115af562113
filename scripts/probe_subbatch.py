"""Does running the early (memory-bound) ResNet stages per sub-batch keep their streams in the Infinity Cache?
Times res2 / res3 / res4 on the bench's batch of 8 against the same work in sub-batches of 4 / 2 / 1 images."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvc_amd.config import set_global_cfg
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn

cfg = base_rcnn_fpn(num_classes=80, device="cuda:0")
set_global_cfg(cfg)
model = build_model(cfg)
syn.conditioned_r50_fpn_(model)
model.eval()
bu = model.backbone.bottom_up
D = "cuda:0"
g = torch.Generator(device=D).manual_seed(0)
shapes = {"res2": (8, 200, 336, 64), "res3": (8, 200, 336, 256), "res4": (8, 100, 168, 512)}


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for stage, name in bu.stages_and_names:
        if name not in shapes:
            continue
        x = torch.randn(*shapes[name], device=D, generator=g).abs().contiguous()

        def run(sub):
            outs = []
            for i in range(0, 8, sub):
                y = x[i: i + sub]
                for blk in stage:
                    y = blk.forward_nhwc(y) if hasattr(blk, "forward_nhwc") else blk(y)
                outs.append(y)
            return outs

        ref = torch.cat(run(8))
        for sub in (8, 4, 2, 1):
            t = timeit(lambda: run(sub))
            got = torch.cat(run(sub))
            print("%s sub-batch %d: %.3f ms   max|diff| vs batch 8 %.2e" % (name, sub, t, (got - ref).abs().max().item()))
