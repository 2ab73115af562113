"""Does the trunk run faster per image when a sub-batch's activations fit the 256 MB Infinity Cache?  The bottom-up stages and the
whole backbone timed at 1, 2, 4 and 8 images per call, all scaled to 8 images."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
dev = torch.device("cuda:0")
model = build_model(base_rcnn_fpn(depth=50, num_classes=80, device="cuda:0")).eval()
syn.conditioned_r50_fpn_(model)
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


with torch.no_grad():
    images = model.preprocess_image(batch)
    from lvc_amd.modeling.backbone.resnet import _as_nhwc4
    x4 = _as_nhwc4(images.tensor)
    bu = model.backbone.bottom_up
    stem = bu.stem.forward_nhwc(x4)
    feats = {"stem": stem}
    x = stem
    for stage, name in bu.stages_and_names:
        for blk in stage:
            x = blk.forward_nhwc(x)
        feats[name] = x

    def run_stage(stage, xin):
        blocks = list(stage)
        t = None
        x = xin
        for i, blk in enumerate(blocks):
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            x, t = blk.forward_chained(x, t, nxt)
        return x

    prev = "stem"
    for stage, name in bu.stages_and_names:
        xin = feats[prev]
        row = []
        for nb in (1, 2, 4, 8):
            parts = [xin[i:i + nb] for i in range(0, 8, nb)]
            row.append(timed(lambda: [run_stage(stage, p) for p in parts]))
        print("%s (input %s, %.0f MB): 8 images as 8x1 %.3f ms | 4x2 %.3f | 2x4 %.3f | 1x8 %.3f" %
              (name, tuple(xin.shape), xin.numel() * 4 / 1e6, *row))
        prev = name
    row = []
    for nb in (1, 2, 4, 8):
        parts = [x4[i:i + nb] for i in range(0, 8, nb)]
        row.append(timed(lambda: [model.backbone.forward_nhwc(p) for p in parts]))
    print("backbone + FPN: 8x1 %.3f ms | 4x2 %.3f | 2x4 %.3f | 1x8 %.3f" % tuple(row))
