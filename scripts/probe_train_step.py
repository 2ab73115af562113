"""Time one training step (forward + backward + SGD) of the base detector / box-corrector configs at 800x1333.
usage: probe_train_step.py [detector|corrector] [batch] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd.config import set_global_cfg
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.structures import Boxes, Instances
from lvc_amd.utils import synthetic as syn
from lvc_amd.utils.events import EventStorage

which = sys.argv[1] if len(sys.argv) > 1 else "detector"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
loss_scale = float(os.environ.get("LOSS_SCALE", "0"))    # > 0: lvc_amd.solver.LossScaler (data gradients on the f16x2 kernels)
cfg = base_rcnn_fpn(num_classes=60)
M = cfg.MODEL
if which == "corrector":
    M.ROI_HEADS.NAME = "CascadeROIHeads"; M.ROI_HEADS.OUTPUT_LAYER = "BoxOnlyLayersCascade"
    M.ROI_HEADS.PROPOSAL_APPEND_GT = False; M.ROI_HEADS.POSITIVE_FRACTION = 1.0
    M.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 64; M.ROI_HEADS.IOU_THRESHOLDS = [0.3]
    M.ROI_BOX_HEAD.NUM_FC = 3; M.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    M.ROI_BOX_CASCADE_HEAD.IOUS = (0.3, 0.5, 0.7); M.PROPOSAL_GENERATOR.NAME = "RBG"; M.LOAD_PROPOSALS = True
set_global_cfg(cfg)
model = build_model(cfg)
syn.conditioned_r50_fpn_(model)
model.train()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
batch = []
for i in range(B):
    h, w = 800, 1333
    n = 8
    x1 = torch.rand(n, generator=g) * (w - 300); y1 = torch.rand(n, generator=g) * (h - 300)
    bw = 40 + torch.rand(n, generator=g) * 250; bh = 40 + torch.rand(n, generator=g) * 250
    boxes = torch.stack([x1, y1, x1 + bw, y1 + bh], 1)
    inst = Instances((h, w)); inst.gt_boxes = Boxes(boxes); inst.gt_classes = torch.randint(0, 60, (n,), generator=g)
    d = {"image": syn.synthetic_image(1 + i).to(dev), "instances": inst, "height": h, "width": w}
    if which == "corrector":
        props = Instances((h, w))
        props.proposal_boxes = Boxes(boxes.repeat(8, 1) + torch.randn(8 * n, 4, generator=g) * 10)
        props.objectness_logits = torch.zeros(8 * n)
        d["proposals"] = props
    batch.append(d)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9, weight_decay=1e-4)
scaler = None
if loss_scale > 0:
    from lvc_amd.solver import LossScaler
    scaler = LossScaler(init_scale=loss_scale)
phase_ms = {}
if os.environ.get("PHASES"):      # wall time of the forward's phases (synchronising hooks: use for attribution only)
    def _hook(name, mod):
        def pre(m, a, k=None):
            torch.cuda.synchronize(); m._t0 = time.perf_counter()
        def post(m, a, o):
            torch.cuda.synchronize(); phase_ms.setdefault(name, []).append((time.perf_counter() - m._t0) * 1e3)
        mod.register_forward_pre_hook(pre); mod.register_forward_hook(post)
    for nm in ("backbone", "proposal_generator", "roi_heads"):
        if getattr(model, nm, None) is not None:
            _hook(nm, getattr(model, nm))
    if hasattr(model.proposal_generator, "rpn_head"):
        _hook("rpn_head", model.proposal_generator.rpn_head)
    if hasattr(model.roi_heads, "box_pooler"):
        _hook("box_pooler", model.roi_heads.box_pooler); _hook("box_head", model.roi_heads.box_head); _hook("box_predictor", model.roi_heads.box_predictor)
times = []
with EventStorage(0):
    for it in range(steps + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        losses = model(batch)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        if scaler is not None:
            scaler.backward(sum(losses.values()))
        else:
            sum(losses.values()).backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if scaler is not None:
            scaler.step(opt)
        else:
            opt.step()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if it >= 2:
            times.append((t1 - t0, t2 - t1, t3 - t2))
        print(it, {k: round(float(v.detach()), 4) for k, v in losses.items()}, flush=True)
f = sum(t[0] for t in times) / len(times); b = sum(t[1] for t in times) / len(times); o = sum(t[2] for t in times) / len(times)
print("%s batch %d: forward %.1f ms  backward %.1f ms  sgd %.1f ms  -> %.2f img/s" % (which, B, f * 1e3, b * 1e3, o * 1e3, B / (f + b + o)))
if phase_ms:
    print({k: round(sum(v[2:]) / max(1, len(v[2:])), 2) for k, v in phase_ms.items()})
