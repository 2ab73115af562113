import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import gold, r50_state_dict
from test_gpu_e2e import _model
from lvc_amd.utils import synthetic as syn
from oracle import rcnn as orc

model = _model()
small = sys.argv[1:] and sys.argv[1] == "small"
if small:
    inputs = [{"image": syn.synthetic_image(3, 240, 320), "height": 480, "width": 640},
              {"image": syn.synthetic_image(4, 200, 352), "height": 200, "width": 352}]
    g = gold("e2e_r50_fpn_small")
else:
    inputs = [{"image": syn.synthetic_image(1), "height": 800, "width": 1333},
              {"image": syn.synthetic_image(2), "height": 800, "width": 1333}]
    g = gold("e2e_r50_fpn_800x1333")
sd = r50_state_dict()
with torch.no_grad():
    ores, mid = orc.generalized_rcnn_inference(sd, orc.RCNNSpec(), inputs, return_intermediates=True)
    images = model.preprocess_image(inputs)
    print("preprocess diff", float((images.tensor.cpu() - mid["images"]).abs().max()))
    bu = model.backbone.bottom_up(images.tensor)
    oresnet = orc.resnet(sd, mid["images"], 50)
    for k in bu:
        d = (bu[k].cpu() - oresnet[k]).abs()
        print(k, "max diff %.3e  rel-to-max %.3e  (abs max %.2f)" % (float(d.max()), float(d.max() / oresnet[k].abs().max()), float(oresnet[k].abs().max())))
    feats = model.backbone(images.tensor)
    for k in feats:
        d = (feats[k].cpu() - mid["feats"][k]).abs()
        print(k, "max diff %.3e  rel %.3e" % (float(d.max()), float(d.max() / mid["feats"][k].abs().max())))
    props, _ = model.proposal_generator(images, feats, None)
    for i in range(len(inputs)):
        pb = props[i].proposal_boxes.tensor.cpu(); ob = mid["proposals"][i][0]
        n = min(len(pb), len(ob))
        d = (pb[:n] - ob[:n]).abs().max(1)[0]
        print("img", i, "nprops", len(pb), len(ob), "ordered max diff %.3e" % float(d.max()), "n>1e-3:", int((d > 1e-3).sum()),
              "logit diff %.3e" % float((props[i].objectness_logits.cpu()[:n] - mid["proposals"][i][1][:n]).abs().max()))
    out = model(inputs)
    for i in range(len(inputs)):
        inst = out[i]["instances"].to("cpu")
        gb, gs, gc = ores[i]["pred_boxes"], ores[i]["scores"], ores[i]["pred_classes"]
        n = min(len(inst), len(gb))
        print("img", i, "ndet", len(inst), len(gb), "ordered box diff %.3e score diff %.3e cls mismatch %d" % (
            float((inst.pred_boxes.tensor[:n] - gb[:n]).abs().max()), float((inst.scores[:n] - gs[:n]).abs().max()),
            int((inst.pred_classes[:n] != gc[:n]).sum())))
