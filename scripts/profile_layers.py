"""Per-launch table of the conv/GEMM kernel for one bench step (GPU only): shape, ms, TFLOP/s."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd import kernels as K
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn

model = build_model(base_rcnn_fpn()).eval()
syn.conditioned_r50_fpn_(model)
dev = torch.device("cuda:0")
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]
shapes = []
orig = K.conv2d_nhwc
def wrapped(x, pc, relu=False, residual=None, res_mode=0, out=None):
    shapes.append((tuple(x.shape), pc.K, pc.R, pc.stride))
    return orig(x, pc, relu=relu, residual=residual, res_mode=res_mode, out=out)
for _ in range(3):
    with torch.no_grad(): model.inference_batched(batch)
torch.cuda.synchronize()
import lvc_amd.layers.wrappers as W, lvc_amd.modeling.proposal_generator.rpn as R
K.conv2d_nhwc = wrapped
reps = 5
t = K.LaunchTimer(); K.CONV_TIMER = t
for _ in range(reps):
    with torch.no_grad(): model.inference_batched(batch)
torch.cuda.synchronize()
K.CONV_TIMER = None
n = len(t.records) // reps
tot = 0
rows = {}
for i in range(n):
    ms = sum(t.records[r * n + i][1].elapsed_time(t.records[r * n + i][2]) for r in range(reps)) / reps
    fl = t.records[i][0]
    key = shapes[i]
    a = rows.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl
    tot += ms
print("%-34s %5s %3s %2s %5s %9s %8s %6s" % ("input NHWC", "K", "R", "s", "calls", "ms(total)", "TF/s", "%time"))
for key, (c, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%-34s %5d %3d %2d %5d %9.3f %8.1f %6.1f" % (str(key[0]), key[1], key[2], key[3], c, ms, fl / ms / 1e9, 100 * ms / tot))
print("total conv ms/step %.3f" % tot)
# whole-step wall
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    with torch.no_grad(): model.inference_batched(batch)
e1.record(); torch.cuda.synchronize()
print("step ms (events, no timer)", e0.elapsed_time(e1) / 5)
