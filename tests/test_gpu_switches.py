"""Every surviving LVC_* switch is exercised on the device: the ones without a test of their own elsewhere.
  LVC_CONV_ENGINE=f32  -> kernels.CONV_ENGINE  (every conv / GEMM on the exact fp32 MFMA kernel)
  LVC_CHAIN=0          -> kernels.CHAIN        (conv3 -> next conv1 as two launches)
  resnet.FUSE_STRIDED_PROJECTION, fpn.MERGE_OUTPUT_CONVS, rpn.MERGE_LEVELS / MERGE_LEVELS_CONV (module constants)  (the RPN head's predictor / 3x3 conv over all levels as one launch each)
and the attention's range report (kernels.mha -> the shared error word)."""
import pytest
import torch

from helpers import match_fraction, r50_state_dict

pytestmark = pytest.mark.gpu


def _inputs():
    from lvc_amd.utils import synthetic as syn

    return [{"image": syn.synthetic_image(3, 240, 320)}, {"image": syn.synthetic_image(4, 200, 352)}]


def test_conv_engine_f32_runs_the_detector(monkeypatch):
    from lvc_amd import kernels as K
    from oracle import rcnn as orc
    from test_gpu_e2e import _model

    monkeypatch.setattr(K, "CONV_ENGINE", "f32")
    model = _model()
    timer = K.LaunchTimer()
    monkeypatch.setattr(K, "CONV_TIMER", timer)
    with torch.no_grad():
        out = model(_inputs())
    monkeypatch.setattr(K, "CONV_TIMER", None)
    assert {r[3] for r in timer.records} == {"f32"}, {r[3] for r in timer.records}
    ref = orc.generalized_rcnn_inference(r50_state_dict(), orc.RCNNSpec(), _inputs())
    for o, r in zip(out, ref):
        inst = o["instances"].to("cpu")
        frac, wb, ws = match_fraction(inst.pred_boxes.tensor, inst.scores, inst.pred_classes, r["pred_boxes"], r["scores"], r["pred_classes"],
                                      box_tol=0.1, score_tol=2e-3)
        assert len(inst) == len(r["scores"]) and frac >= 0.9, (len(inst), frac)


def test_chain_switch_two_launches_equal_one(monkeypatch):
    """The trunk with the conv3 -> conv1 pairs of res2 / res3 as one launch each (default) against the two-launch form, and which
    engine tags a step launches in either case."""
    from lvc_amd import kernels as K
    from lvc_amd.modeling.backbone import resnet as R
    from test_gpu_e2e import _model

    model = _model()
    x = torch.randn(2, 3, 416, 608, generator=torch.Generator().manual_seed(1)).cuda() * 40
    res, tags = {}, {}
    monkeypatch.setattr(R, "FUSE_STRIDED_PROJECTION", False)      # (without the chain res3.0 would take that form: one launch fewer)
    for chain in (True, False):
        monkeypatch.setattr(K, "CHAIN", chain)
        timer = K.LaunchTimer()
        monkeypatch.setattr(K, "CONV_TIMER", timer)
        with torch.no_grad():
            res[chain] = {k: v.clone() for k, v in model.backbone.bottom_up(x).items()}
        monkeypatch.setattr(K, "CONV_TIMER", None)
        tags[chain] = [r[3] for r in timer.records]
    assert tags[True].count("f16s1_chain") == 5 and "f16s1_chain" not in tags[False]
    assert len(tags[False]) - len(tags[True]) == 5
    for k in res[True]:
        scale = float(res[False][k].abs().max())
        assert float((res[True][k] - res[False][k]).abs().max()) <= 2e-5 * scale, k


def test_rpn_head_over_all_levels_in_two_launches(monkeypatch):
    """StandardRPNHead: the 3x3 conv over the five levels as ONE launch (kernels.conv3x3_levels, rpn.MERGE_LEVELS_CONV) writing one
    hidden buffer, and ONE predictor launch over it (rpn.MERGE_LEVELS) -- against a launch per level and layer: eight launches fewer,
    the same products per output (a tile's contraction may be split over workers at other points, and p5 / p6 run on the fp16-split
    kernel inside the grouped launch: fp32 summation order, within the conv tolerance of 2e-5 of the output scale; the predictor
    alone 1e-6)."""
    from lvc_amd import kernels as K
    from lvc_amd.modeling.proposal_generator import rpn as R
    from test_gpu_e2e import _model

    model = _model()
    g = torch.Generator().manual_seed(2)
    feats = [torch.randn(2, h, w, 256, generator=g).cuda() for h, w in ((104, 152), (52, 76), (26, 38), (13, 19), (7, 10))]
    res, n = {}, {}
    for merge, merge_conv in ((True, True), (True, False), (False, False)):
        monkeypatch.setattr(R, "MERGE_LEVELS", merge)
        monkeypatch.setattr(R, "MERGE_LEVELS_CONV", merge_conv)
        timer = K.LaunchTimer()
        monkeypatch.setattr(K, "CONV_TIMER", timer)
        with torch.no_grad():
            res[merge, merge_conv] = [t.clone() for t in model.proposal_generator.rpn_head.forward_nhwc(feats)]
        monkeypatch.setattr(K, "CONV_TIMER", None)
        n[merge, merge_conv] = len(timer.records)
    assert n[False, False] == 10 and n[True, False] == 6 and n[True, True] == 2
    for a, b, c, f in zip(res[True, True], res[True, False], res[False, False], feats):
        assert a.shape == b.shape == c.shape == (2, f.shape[1], f.shape[2], 16) and a.is_contiguous()
        scale = max(1.0, float(c.abs().max()))
        assert float((b - c).abs().max()) <= 1e-6 * scale
        assert float((a - c).abs().max()) <= 2e-5 * scale
    assert K.conv_error_word(feats[0].device) == 0


def test_fpn_output_convs_of_all_levels_in_one_launch(monkeypatch):
    """FPN: the four 3x3 output convs (a layer per level) as ONE launch (kernels.conv3x3_levels with a list of layers,
    fpn.MERGE_OUTPUT_CONVS) against a launch per level: three launches fewer, outputs within the conv tolerance (p5 moves from the
    bf16x3 kernel to the fp16-split one inside the grouped launch)."""
    from lvc_amd import kernels as K
    from lvc_amd.modeling.backbone import fpn as F_
    from test_gpu_e2e import _model

    model = _model()
    x = torch.randn(2, 3, 416, 608, generator=torch.Generator().manual_seed(1)).cuda() * 40
    res, n = {}, {}
    for merge in (True, False):
        monkeypatch.setattr(F_, "MERGE_OUTPUT_CONVS", merge)
        timer = K.LaunchTimer()
        monkeypatch.setattr(K, "CONV_TIMER", timer)
        with torch.no_grad():
            res[merge] = {k: v.clone() for k, v in model.backbone(x).items()}
        monkeypatch.setattr(K, "CONV_TIMER", None)
        n[merge] = len(timer.records)
    assert n[False] - n[True] == 3
    assert list(res[True]) == list(res[False]) == ["p2", "p3", "p4", "p5", "p6"]
    for k in res[True]:
        scale = float(res[False][k].abs().max())
        assert res[True][k].shape == res[False][k].shape
        assert float((res[True][k] - res[False][k]).abs().max()) <= 2e-5 * scale, k
    assert K.conv_error_word(x.device) == 0


def test_strided_projection_shortcut_fused_with_conv3(monkeypatch):
    """res4.0 / res5.0 (res3.0 keeps its chained launch): conv3 and the stride-2 projection shortcut as ONE pointwise GEMM over [conv2
    output | block input sampled at the even pixels] (resnet.FUSE_STRIDED_PROJECTION) against the two launches + residual: two conv launches fewer,
    features within the conv tolerance (the FrozenBN scales go into the fused weights: one more fp32 rounding per weight)."""
    from lvc_amd import kernels as K
    from lvc_amd.modeling.backbone import resnet as R
    from test_gpu_e2e import _model

    model = _model()
    x = torch.randn(2, 3, 416, 608, generator=torch.Generator().manual_seed(1)).cuda() * 40
    res, n = {}, {}
    for fuse in (True, False):
        monkeypatch.setattr(R, "FUSE_STRIDED_PROJECTION", fuse)
        timer = K.LaunchTimer()
        monkeypatch.setattr(K, "CONV_TIMER", timer)
        with torch.no_grad():
            res[fuse] = {k: v.clone() for k, v in model.backbone.bottom_up(x).items()}
        monkeypatch.setattr(K, "CONV_TIMER", None)
        n[fuse] = len(timer.records)
    assert n[False] - n[True] == 2
    for k in res[True]:
        scale = float(res[False][k].abs().max())
        assert float((res[True][k] - res[False][k]).abs().max()) <= 2e-5 * scale, k
    assert K.conv_error_word(x.device) == 0


def test_mha_reports_operands_beyond_fp16():
    from lvc_amd import kernels as K

    d = torch.device("cuda:0")
    B, N, H = 2, 197, 6
    qkv = torch.randn(B * N, 3 * H * 64, device=d)
    y = K.mha(qkv, B, N, H, 64, 0.125)
    y2 = K.mha(qkv, B, N, H, 64, 0.125, mfma=False)
    assert float((y - y2).abs().max()) <= 2e-5 * float(y2.abs().max())
    assert K.conv_error_word(d) == 0
    qkv[5, 70] = 1.0e6          # a key / value entry fp16 cannot hold
    K.mha(qkv, B, N, H, 64, 0.125)
    with pytest.raises(K.Fp16RangeError) as e:
        K.check_conv_error_word(d)
    assert not e.value.rerouted and K.conv_error_word(d) == 0
