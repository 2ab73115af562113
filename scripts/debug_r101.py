"""R101 trunk: error of the CPU fp32 evaluation and of the GPU trunk against the fp64 evaluation of the same weights."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
from oracle import rcnn as orc
import lvc_amd.kernels as K
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 101
model = build_model(base_rcnn_fpn(depth=depth)).eval()
syn.conditioned_r50_fpn_(model, depth=depth)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
sd64 = {k: v.double() for k, v in sd.items()}
inputs = [{"image": syn.synthetic_image(3, 240, 320)}, {"image": syn.synthetic_image(4, 200, 352)}]
spec = orc.RCNNSpec(depth=depth)
with torch.no_grad():
    imgs, _ = orc.preprocess([b["image"] for b in inputs], spec.pixel_mean, spec.pixel_std, 32)
    f32 = orc.fpn(sd, orc.resnet(sd, imgs, depth))
    f64 = orc.fpn(sd64, orc.resnet(sd64, imgs.double(), depth))
    gf = model.backbone(model.preprocess_image(inputs).tensor)
for k in f64:
    s = float(f64[k].abs().max())
    print(depth, k, "CPU fp32 vs fp64 %.2e   GPU vs fp64 %.2e   GPU vs CPU fp32 %.2e" % (
        float((f32[k].double() - f64[k]).abs().max()) / s, float((gf[k].cpu().double() - f64[k]).abs().max()) / s,
        float((gf[k].cpu() - f32[k]).abs().max()) / s))
