"""Box-corrector inference (GeneralizedRCNNRegOnly + CascadeROIHeads, class-agnostic 3-stage cascade) vs the golden
produced by the reference on CPU.  Boxes pass three decode stages, so the fp32-trunk noise floor applies (see
tests/test_gpu_e2e.py): tolerance 0.1 px, classes exact."""
import pytest
import torch

from helpers import gold

pytestmark = pytest.mark.gpu


def _model():
    from lvc_amd.config import set_global_cfg
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    cfg = base_rcnn_fpn(num_classes=60)
    M = cfg.MODEL
    M.META_ARCHITECTURE = "GeneralizedRCNNRegOnly"
    M.ROI_HEADS.NAME = "CascadeROIHeads"
    M.ROI_HEADS.OUTPUT_LAYER = "BoxOnlyLayersCascade"
    M.ROI_HEADS.PROPOSAL_APPEND_GT = False
    M.ROI_HEADS.POSITIVE_FRACTION = 1.0
    M.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 64
    M.ROI_HEADS.IOU_THRESHOLDS = [0.3]
    M.ROI_BOX_HEAD.NUM_FC = 3
    M.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    M.ROI_BOX_CASCADE_HEAD.IOUS = (0.3, 0.5, 0.7)
    M.PROPOSAL_GENERATOR.NAME = "RBG"
    M.LOAD_PROPOSALS = True
    set_global_cfg(cfg)
    model = build_model(cfg).eval()
    g = gold("cascade_state_dict_keys")
    mine = {k: str(tuple(v.shape)) for k, v in model.state_dict().items()}
    assert list(mine) == g["keys"].tolist() and list(mine.values()) == g["shapes"].tolist()
    syn.conditioned_r50_fpn_(model)
    return model


def test_box_corrector_matches_reference():
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn

    g = gold("box_corrector")
    model = _model()
    batch = []
    for i, (h, w, seed, oh, ow) in enumerate([(240, 320, 3, 480, 640), (200, 352, 4, 200, 352)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(g["in_boxes%d" % i].clone())
        inst.gt_classes = g["in_classes%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "height": oh, "width": ow})
    with torch.no_grad():
        out = model(batch)
    for i, o in enumerate(out):
        assert "image" not in o
        inst = o["instances"].to("cpu")
        assert inst.pred_classes.tolist() == g["out_classes%d" % i].tolist()
        err = float((inst.pred_boxes.tensor - g["out_boxes%d" % i]).abs().max())
        print("image", i, "max box error", err)
        assert err <= 0.1
