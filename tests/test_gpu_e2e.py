"""End-to-end parity of the MI355X GeneralizedRCNN forward against golden vectors produced by the
reference's MODEL.DEVICE=cpu path (oracle/make_golden.py).

What "parity" can mean end to end.  north_star asks for boxes/scores within 1e-3 of the CPU path.  Every
discrete stage (top-k, NMS keep indices, level assignment, class ids) is bit-exact given equal inputs and
every fp32 stage matches its CPU twin to 1e-6..2e-5 relative (tests/test_gpu_kernels.py, test_gpu_boxes.py).
Chained through the 53-layer trunk, two *equally valid* fp32 evaluations (mkldnn's blocked summation on the
CPU, the MFMA k-ordered chain here) drift apart by 2e-5..7e-5 of the feature scale -- and the reference CPU
path is itself that far from the fp64 evaluation of the same weights (test_trunk_error_vs_fp64 below measures
both).  With the conditioned synthetic weights that is ~1e-3 absolute on p2..p6, ~5e-4 on RPN logits and up to
a few 1e-2 px on decoded boxes, i.e. the CPU path does not determine its own outputs to 1e-3.  The end-to-end
tests therefore assert (a) trunk error vs fp64 no larger than 1.5x the CPU reference's own error vs fp64,
(b) >= 90 % of the reference detections reproduced with identical class, |score| <= 2e-3, |box| <= 0.1 px, the
remainder being near-tie reorderings in top-k/NMS, and print the exact statistics."""

import pytest
import torch

from helpers import ROOT, gold, match_fraction, r50_state_dict

pytestmark = pytest.mark.gpu

# Identity bars (which detection is which): R50's are 0.1 px / 2e-3 = ~40 x its CPU path's own median fp32-vs-fp64 noise (2.5e-3 px / 6e-5).
# R101's conditioned weights carry ~6 x that noise at 800x1333, so its bars are DERIVED in the test from the noise run itself (oracle/noise.py:
# IDENT_K x the medians of a first, generous matching -- round 5 had 0.5 px / 1e-2 set by hand).  The ACCURACY bars are not these: median / p90
# of the matched differences <= K_NOISE x the noise, the largest <= K_MAX x its largest, found fraction >= the noise run's own - IDENT_MARGIN.


def _model(fuse_type="sum"):
    from lvc_amd.config import get_cfg
    from lvc_amd.modeling import build_model

    cfg = get_cfg()
    cfg.MODEL.FPN.FUSE_TYPE = fuse_type
    cfg.MODEL.DEVICE = "cuda"
    cfg.MODEL.BACKBONE.NAME = "build_resnet_fpn_backbone"
    cfg.MODEL.RESNETS.OUT_FEATURES = ["res2", "res3", "res4", "res5"]
    cfg.MODEL.FPN.IN_FEATURES = ["res2", "res3", "res4", "res5"]
    cfg.MODEL.ANCHOR_GENERATOR.SIZES = [[32], [64], [128], [256], [512]]
    cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS = [[0.5, 1.0, 2.0]]
    cfg.MODEL.RPN.IN_FEATURES = ["p2", "p3", "p4", "p5", "p6"]
    cfg.MODEL.RPN.PRE_NMS_TOPK_TEST = 1000
    cfg.MODEL.RPN.POST_NMS_TOPK_TEST = 1000
    cfg.MODEL.ROI_HEADS.NAME = "StandardROIHeads"
    cfg.MODEL.ROI_HEADS.IN_FEATURES = ["p2", "p3", "p4", "p5"]
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 80
    cfg.MODEL.ROI_BOX_HEAD.NAME = "FastRCNNConvFCHead"
    cfg.MODEL.ROI_BOX_HEAD.NUM_FC = 2
    cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION = 7
    cfg.MODEL.RESNETS.DEPTH = 50
    model = build_model(cfg).eval()
    missing = model.load_state_dict(r50_state_dict(), strict=True)
    return model


def _check(name, inputs, model, derive_identity=False):
    """Every bar is measured in the test (oracle/noise.py): the reference path's OWN fp32-vs-fp64 noise on these inputs gives (identity) the
    fraction of detections a valid fp32 evaluation finds within the identity bars -- the HIP path may be IDENT_MARGIN (2 points) below it --
    and (accuracy) medians / p90 (x K_NOISE = 2) and the largest matched difference (x K_MAX = 3), so that a kernel regression worth a few
    1e-2 px, or one that loses a few percent of the detections, fails."""
    from oracle import noise as onoise
    from oracle import rcnn as orc

    g = gold(name)
    with torch.no_grad():
        out = model(inputs)
    cpu_in = [dict(b, image=b["image"].cpu()) for b in inputs]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    depth = 101 if "backbone.bottom_up.res4.22.conv1.weight" in sd else 50
    nz = onoise.fp32_vs_fp64(sd, orc.RCNNSpec(depth=depth), cpu_in, derive_identity=derive_identity)
    box_tol, score_tol = nz["box_tol"], nz["score_tol"]
    hip = [(o["instances"].pred_boxes.tensor.cpu(), o["instances"].scores.cpu(), o["instances"].pred_classes.cpu()) for o in out]
    refd = [(g["det_boxes%d" % i], g["det_scores%d" % i], g["det_classes%d" % i]) for i in range(len(inputs))]
    dev = onoise.deviation(hip, refd, box_tol, score_tol)
    ok, bars, msg = onoise.gate(dev, nz)
    print("%s (identity bars %.2g px / %.2g): found %.2f%% (reference path of its own fp64 detections: %.2f%%) | |hip - reference| box median "
          "%.2e p90 %.2e max %.2e px, score median %.2e p90 %.2e max %.2e | reference fp32-vs-fp64 noise: box median %.2e p90 %.2e max %.2e, "
          "score median %.2e p90 %.2e max %.2e | bars %s"
          % (name, box_tol, score_tol, 100 * dev["matched_fraction"], 100 * nz["matched_fraction"], dev["box_median"], dev["box_p90"],
             dev["box_max"], dev["score_median"], dev["score_p90"], dev["score_max"], nz["box_median"], nz["box_p90"], nz["box_max"],
             nz["score_median"], nz["score_p90"], nz["score_max"], {k: "%.1e" % v for k, v in bars.items()}))
    assert ok, msg
    for i in range(len(inputs)):
        inst = out[i]["instances"].to("cpu")
        assert len(inst) == len(g["det_scores%d" % i])
        frac, wb, ws = match_fraction(inst.pred_boxes.tensor, inst.scores, inst.pred_classes, g["det_boxes%d" % i],
                                      g["det_scores%d" % i], g["det_classes%d" % i], box_tol=box_tol, score_tol=score_tol)
        tight, _, _ = match_fraction(inst.pred_boxes.tensor, inst.scores, inst.pred_classes, g["det_boxes%d" % i],
                                     g["det_scores%d" % i], g["det_classes%d" % i], box_tol=1e-3, score_tol=1e-3)
        print("%s image %d: matched %.0f%% (worst box %.2e px, worst score %.2e); within 1e-3: %.0f%%"
              % (name, i, 100 * frac, wb, ws, 100 * tight))
        # per image (100 detections: one is a point) the same bar with the sampling noise of that count on top
        n = max(1, len(g["det_scores%d" % i]))
        p0 = nz["matched_fraction"]
        assert frac >= p0 - onoise.IDENT_MARGIN - 2.0 * (max(p0 * (1 - p0), 1.0 / n) / n) ** 0.5, \
            "image %d: only %.0f%% of reference detections reproduced" % (i, 100 * frac)
    return g


def test_e2e_small_matches_reference_cpu():
    from lvc_amd.utils import synthetic as syn

    model = _model()
    inputs = [{"image": syn.synthetic_image(3, 240, 320), "height": 480, "width": 640},
              {"image": syn.synthetic_image(4, 200, 352), "height": 200, "width": 352}]
    _check("e2e_r50_fpn_small", inputs, model)


@pytest.mark.parametrize("merged", [True, False])
def test_fpn_avg_fusion_matches_reference(merged, monkeypatch):
    """MODEL.FPN.FUSE_TYPE "avg" (reference fpn.py:133-134; no shipped config selects it): p2..p6 of the small batch against the reference
    backbone's (tests/golden/fpn_avg_small.npz), at the bar of the sum pyramid (1e-4 of the level's largest value), on both top-down walks
    (output convs merged into one launch / level by level)."""
    from lvc_amd.modeling.backbone import fpn as F
    from lvc_amd.utils import synthetic as syn

    monkeypatch.setattr(F, "MERGE_OUTPUT_CONVS", merged)
    g = gold("fpn_avg_small")
    model = _model("avg")
    inputs = [{"image": syn.synthetic_image(3, 240, 320), "height": 480, "width": 640},
              {"image": syn.synthetic_image(4, 200, 352), "height": 200, "width": 352}]
    with torch.no_grad():
        feats = model.backbone(model.preprocess_image(inputs).tensor)
        summed = _model("sum").backbone(model.preprocess_image(inputs).tensor)
    for k in ("p2", "p3", "p4", "p5", "p6"):
        got = feats[k][:, ::16, ::2, ::2].cpu()
        assert got.shape == g["feat_" + k].shape
        err = float((got - g["feat_" + k]).abs().max()) / float(g["featstat_" + k][2])
        print(k, "relative error", err)
        assert err <= 1e-4, k
    assert not torch.equal(feats["p2"], summed["p2"]) and torch.equal(feats["p5"], summed["p5"])      # (the coarsest level has no top-down term)


def test_e2e_800x1333_matches_reference_cpu():
    from lvc_amd.utils import synthetic as syn

    model = _model()
    inputs = [{"image": syn.synthetic_image(1), "height": 800, "width": 1333},
              {"image": syn.synthetic_image(2), "height": 800, "width": 1333}]
    g = _check("e2e_r50_fpn_800x1333", inputs, model)
    # intermediate pins: features and proposals
    with torch.no_grad():
        images = model.preprocess_image(inputs)
        feats = model.backbone(images.tensor)
        props, _ = model.proposal_generator(images, feats, None)
    for k in ("p2", "p3", "p4", "p5", "p6"):
        got = feats[k][:, ::16, ::8, ::8].cpu()
        scale = float(g["featstat_" + k][2])
        assert (got - g["feat_" + k]).abs().max() <= 1e-4 * scale, k   # measured 2e-5..7e-5
    # proposals: the bar is what the oracle's own fp32 evaluation reproduces of its fp64 proposals (measured here), minus IDENT_MARGIN
    from helpers import found_bar
    from oracle import rcnn as orc

    sd = r50_state_dict()
    cpu_in = [dict(b, image=b["image"].cpu()) for b in inputs]
    with torch.no_grad():
        _, mid32 = orc.generalized_rcnn_inference(sd, orc.RCNNSpec(), cpu_in, return_intermediates=True)
        _, mid64 = orc.generalized_rcnn_inference({k: v.double() for k, v in sd.items()}, orc.RCNNSpec(), cpu_in, return_intermediates=True)
    for i in range(2):
        pb = props[i].proposal_boxes.tensor.cpu()
        assert pb.shape == g["prop_boxes%d" % i].shape
        d = (pb[:, None, :] - g["prop_boxes%d" % i][None, :, :]).abs().max(dim=2)[0].min(dim=1)[0]
        frac = float((d <= 0.1).float().mean())
        b32, b64 = mid32["proposals"][i][0].double(), mid64["proposals"][i][0]
        p0 = float(((b32[:, None, :] - b64[None, :, :]).abs().max(dim=2)[0].min(dim=1)[0] <= 0.1).double().mean())
        print("image %d: %.1f%% of reference proposals reproduced within 0.1 px (the oracle's fp32 run of its fp64 proposals: %.1f%%)"
              % (i, 100 * frac, 100 * p0))
        assert frac >= found_bar(p0, len(pb)), (frac, p0)


def test_trunk_error_vs_fp64():
    """fp32 noise floor: the GPU trunk must be as close to the fp64 evaluation as the CPU reference path is."""
    from lvc_amd.utils import synthetic as syn
    from oracle import rcnn as orc

    model = _model()
    inputs = [{"image": syn.synthetic_image(3, 240, 320)}, {"image": syn.synthetic_image(4, 200, 352)}]
    sd = r50_state_dict()
    sd64 = {k: v.double() for k, v in sd.items()}
    spec = orc.RCNNSpec()
    with torch.no_grad():
        imgs, _ = orc.preprocess([b["image"] for b in inputs], spec.pixel_mean, spec.pixel_std, 32)
        f32 = orc.fpn(sd, orc.resnet(sd, imgs, 50))
        f64 = orc.fpn(sd64, orc.resnet(sd64, imgs.double(), 50))
        gf = model.backbone(model.preprocess_image(inputs).tensor)
    for k in f64:
        s = float(f64[k].abs().max())
        e_cpu = float((f32[k].double() - f64[k]).abs().max()) / s
        e_gpu = float((gf[k].cpu().double() - f64[k]).abs().max()) / s
        print("%s: cpu-fp32 vs fp64 %.2e   gpu-fp32 vs fp64 %.2e" % (k, e_cpu, e_gpu))
        assert e_gpu <= 1.5 * e_cpu + 1e-6, k


def test_uint8_input_and_registry_surface():
    from lvc_amd.modeling import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY
    from lvc_amd.utils import synthetic as syn

    assert "GeneralizedRCNN" in META_ARCH_REGISTRY and "ProposalNetwork" in META_ARCH_REGISTRY
    assert "build_resnet_fpn_backbone" in BACKBONE_REGISTRY and "RPN" in PROPOSAL_GENERATOR_REGISTRY
    assert "StandardROIHeads" in ROI_HEADS_REGISTRY
    model = _model()
    img = syn.synthetic_image(3, 240, 320).round().clamp(0, 255)
    with torch.no_grad():
        a = model([{"image": img.to(torch.uint8)}])[0]["instances"]
        b = model([{"image": img}])[0]["instances"]
    assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)


def test_r101_trunk_matches_oracle():
    """R101-FPN (BASELINE config 5 backbone): the trunk kernels are depth-agnostic; pin res4's 23 blocks against the
    CPU oracle on a small image (uncalibrated FrozenBN -> large activations, so compare relative to the level's max)."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn
    from oracle import rcnn as orc

    model = build_model(base_rcnn_fpn(depth=101)).eval()
    cpu_sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    sd = syn.conditioned_state_dict(cpu_sd, seed=3)
    for k in sd:  # tame the residual growth without a calibration pass
        if k.endswith("conv3.norm.weight"):
            sd[k] = sd[k] * 0.3
    model.load_state_dict(sd, strict=True)
    assert sum(1 for k in sd if k.startswith("backbone.bottom_up.res4.") and k.endswith("conv1.weight")) == 23
    inputs = [{"image": syn.synthetic_image(5, 128, 160)}]
    spec = orc.RCNNSpec(depth=101)
    with torch.no_grad():
        imgs, _ = orc.preprocess([b["image"] for b in inputs], spec.pixel_mean, spec.pixel_std, 32)
        ref = orc.fpn(sd, orc.resnet(sd, imgs, 101))
        got = model.backbone(model.preprocess_image(inputs).tensor)
    for k in ref:
        s = float(ref[k].abs().max())
        err = float((got[k].cpu() - ref[k]).abs().max()) / s
        print(k, "relative error", err)
        assert err <= 2e-4, k


def test_r101_e2e_800x1333_matches_reference_cpu():
    """R101-FPN at the headline's image size against the reference's CPU run (tests/golden/e2e_r101_fpn_800x1333.npz,
    oracle/make_golden.py:gen_r101_full): detections through the noise gate (oracle/noise.py, bars = K_NOISE x the CPU path's own
    fp32-vs-fp64 noise on these inputs), pyramid features against the sampled reference features."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    model = build_model(base_rcnn_fpn(depth=101)).eval()
    syn.conditioned_r50_fpn_(model, depth=101)
    inputs = [{"image": syn.synthetic_image(1), "height": 800, "width": 1333},
              {"image": syn.synthetic_image(2), "height": 800, "width": 1333}]
    g = _check("e2e_r101_fpn_800x1333", inputs, model, derive_identity=True)
    with torch.no_grad():
        feats = model.backbone(model.preprocess_image(inputs).tensor)
    for name in ("p2", "p3", "p4", "p5", "p6"):
        got = feats[name][:, ::16, ::8, ::8].cpu()
        scale = float(g["featstat_" + name][2])
        err = float((got - g["feat_" + name]).abs().max()) / scale
        print(name, "relative error", err)
        assert err <= 8e-4, name


def test_r101_e2e_small_matches_reference_cpu():
    """R101-FPN end to end (the depth BASELINE config 5 names): detections of the 2-image small batch against the
    reference's CPU run (tests/golden/e2e_r101_fpn_small.npz), state_dict keys/shapes against the reference's, and the
    pyramid features against the sampled reference features.

    The 101-layer trunk with these conditioned random weights has a 5-6x higher fp32 noise floor than R50
    (scripts/debug_r101.py on MI355X: CPU fp32 vs fp64 1.5e-4 (p2) .. 3.9e-4 (p5) of the feature scale, this build vs
    fp64 0.9e-4 .. 2.5e-4, i.e. closer to fp64 than the reference's own CPU path), so the detection tolerances of the
    R50 test are scaled by 5: box 0.5 px, score 1e-2 as IDENTITY bars (which detection is which); the accuracy bars are not set by
    hand: median / p90 of the matched differences within K_NOISE (= 2) x the CPU path's own fp32-vs-fp64 noise for THESE weights (`_check`,
    oracle/noise.py); the features must agree to 8e-4 (2x the CPU path's own error)."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    model = build_model(base_rcnn_fpn(depth=101)).eval()
    k = gold("r101_fpn_state_dict_keys")
    mine = {n: str(tuple(v.shape)) for n, v in model.state_dict().items()}
    assert list(mine) == k["keys"].tolist() and list(mine.values()) == k["shapes"].tolist()
    syn.conditioned_r50_fpn_(model, depth=101)
    inputs = [{"image": syn.synthetic_image(3, 240, 320), "height": 480, "width": 640},
              {"image": syn.synthetic_image(4, 200, 352), "height": 200, "width": 352}]
    g = _check("e2e_r101_fpn_small", inputs, model, derive_identity=True)
    with torch.no_grad():
        feats = model.backbone(model.preprocess_image(inputs).tensor)
    for name in ("p2", "p3", "p4", "p5", "p6"):
        ref = g["feat_" + name]
        got = feats[name][:, ::8].cpu()
        assert float((got - ref).abs().max()) <= 8e-4 * float(ref.abs().max()), name
