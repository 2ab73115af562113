import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import gold, r50_state_dict
from test_gpu_e2e import _model
from lvc_amd.utils import synthetic as syn
from oracle import rcnn as orc

model = _model()
inputs = [{"image": syn.synthetic_image(3, 240, 320)}, {"image": syn.synthetic_image(4, 200, 352)}]
sd = r50_state_dict()
sd64 = {k: v.double() for k, v in sd.items()}
with torch.no_grad():
    imgs, sizes = orc.preprocess([b["image"] for b in inputs], orc.RCNNSpec().pixel_mean, orc.RCNNSpec().pixel_std, 32)
    r32 = orc.resnet(sd, imgs, 50); f32 = orc.fpn(sd, r32)
    r64 = orc.resnet(sd64, imgs.double(), 50); f64 = orc.fpn(sd64, r64)
    images = model.preprocess_image(inputs)
    g32 = model.backbone.bottom_up(images.tensor); gf = model.backbone(images.tensor)
    for k in r32:
        ref = r64[k]; s = float(ref.abs().max())
        print("%s  cpu32-vs-64 %.3e   gpu32-vs-64 %.3e   cpu32-vs-gpu32 %.3e   (rel to max %.1f)" % (
            k, float((r32[k].double() - ref).abs().max()) / s, float((g32[k].cpu().double() - ref).abs().max()) / s,
            float((g32[k].cpu() - r32[k]).abs().max()) / s, s))
    for k in f32:
        ref = f64[k]; s = float(ref.abs().max())
        print("%s  cpu32-vs-64 %.3e   gpu32-vs-64 %.3e   cpu32-vs-gpu32 %.3e" % (
            k, float((f32[k].double() - ref).abs().max()) / s, float((gf[k].cpu().double() - ref).abs().max()) / s,
            float((gf[k].cpu() - f32[k]).abs().max()) / s))
