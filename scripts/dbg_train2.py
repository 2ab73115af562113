import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import gold
import test_gpu_train as T
from lvc_amd import kernels as K
from lvc_amd.utils.events import EventStorage
torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
g = gold("train_base_steps")
grads, feats = {}, {}
for tag, chain, form in (('p1', False, 2), ('c1', True, 2), ('p2', False, 1), ('c2', False, 1)):
    K.CHAIN = chain; K.HALO_S1 = form; K.PW_S1 = form
    model = T._base_model()
    with EventStorage(0):
        batch = T._batch(g)
        images = model.preprocess_image(batch)
        with torch.no_grad():
            f = model.backbone.bottom_up(images.tensor)
        feats[tag] = {k: v.clone() for k, v in f.items()}
        losses = model(batch)
        print(tag, {k: '%.9f' % float(v) for k, v in losses.items()})
        sum(losses.values()).backward()
        grads[tag] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
def rel(a, b):
    return max(float((grads[a][n] - grads[b][n]).norm() / (grads[b][n].norm() + 1e-30)) for n in grads[a])
for a, b in (("p1", "p2"), ("c1", "c2"), ("p1", "c1"), ("p2", "c2")):
    print(a, b, "worst grad rel diff %.2e" % rel(a, b), "res2 feature diff %.2e" % float((feats[a]["res2"] - feats[b]["res2"]).abs().max()))

for n in ("roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.bbox_pred.weight", "roi_heads.box_head.fc1.weight", "proposal_generator.rpn_head.conv.weight", "proposal_generator.rpn_head.anchor_deltas.weight", "backbone.fpn_output2.weight", "backbone.fpn_output5.weight", "backbone.fpn_lateral2.weight", "backbone.bottom_up.res5.2.conv3.weight"):
    a, b = grads["p1"][n], grads["c1"][n]
    print("%-50s rel diff %.2e" % (n, float((a - b).norm() / b.norm())))
