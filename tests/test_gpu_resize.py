"""Test-time input pipeline on the device: Pillow-exact bilinear resize of uint8 HWC images, fused with
the detector's normalise + pad.  Integer work: bit-exact against the reference's ResizeTransform outputs
(tests/golden/resize.npz) and against the oracle at full size."""
import numpy as np
import pytest
import torch

from helpers import gold

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_resize_matches_reference_transform_bit_exact():
    from lvc_amd.data import ResizeShortestEdge

    g = gold("resize")
    for i in range(int(g["n"])):
        img = g["in%d" % i]
        short, mx, nh, nw = [int(v) for v in g["cfg%d" % i]]
        tfm = ResizeShortestEdge([short, short], mx, "choice").get_transform(img)
        out = tfm.apply_image(img.to(DEV))
        assert out.dtype == torch.uint8 and tuple(out.shape) == (nh, nw, 3)
        assert torch.equal(out.cpu(), g["out%d" % i]), i


def test_resize_full_size_matches_oracle_and_fused_slot():
    """480x640 -> 800x1067 (up) and 1200x1600 -> 800x1067 (anti-aliased down): the uint8 result equals the oracle's; the
    fused NHWC4 slot equals preprocess_into() of that result, zero outside it."""
    from lvc_amd import kernels as K
    from lvc_amd.data import ResizeShortestEdge
    from oracle import resize as orz

    rng = np.random.default_rng(5)
    aug = ResizeShortestEdge(800, 1333, "choice")
    mean, std = [103.53, 116.28, 123.675], [1.0, 57.0, 58.4]
    for h, w in [(480, 640), (1200, 1600), (800, 1333)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        t = aug.get_transform(img)
        if (h, w) == (800, 1333):
            assert (t.new_h, t.new_w) == (800, 1333)
        ref = orz.resize_bilinear_u8(img, t.new_h, t.new_w)
        Hp, Wp = (t.new_h + 31) // 32 * 32, (t.new_w + 31) // 32 * 32 + 32
        slot = torch.full((Hp, Wp, 4), 7.0, device=DEV)
        out = t.apply_image(torch.from_numpy(img).to(DEV), out_slot=slot, mean=mean, std=std)
        assert np.array_equal(out.cpu().numpy(), ref), (h, w)
        two_step = torch.empty_like(slot)
        K.preprocess_into(out.permute(2, 0, 1).contiguous(), two_step, mean, std)
        assert torch.equal(slot, two_step)
        assert float(slot[t.new_h:].abs().sum()) == 0.0 and float(slot[:, t.new_w:].abs().sum()) == 0.0


def test_detector_on_raw_inputs_equals_two_step_pipeline():
    """inference() on {"raw": uint8 HWC} equals inference() on the CHW image the reference's DatasetMapper would have
    produced from it (resize on the host side of the test through the oracle), detections scaled back to the file."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn
    from oracle import resize as orz

    cfg = base_rcnn_fpn()
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 256, 400
    model = build_model(cfg).eval()
    syn.conditioned_r50_fpn_(model)
    raws = [syn.synthetic_image(3, 150, 200).permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).contiguous(),
            syn.synthetic_image(4, 300, 180).permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).contiguous()]
    two_step = []
    for r in raws:
        nh, nw = orz.shortest_edge_size(r.shape[0], r.shape[1], 256, 400)
        rs = torch.from_numpy(orz.resize_bilinear_u8(r.numpy(), nh, nw))
        two_step.append({"image": rs.permute(2, 0, 1).contiguous(), "height": r.shape[0], "width": r.shape[1]})
    with torch.no_grad():
        a = model([{"raw": r} for r in raws])
        b = model(two_step)
    for x, y in zip(a, b):
        ia, ib = x["instances"], y["instances"]
        assert ia.image_size == ib.image_size and len(ia) == len(ib)
        assert torch.equal(ia.pred_boxes.tensor, ib.pred_boxes.tensor) and torch.equal(ia.scores, ib.scores)
        assert torch.equal(ia.pred_classes, ib.pred_classes)
