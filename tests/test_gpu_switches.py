"""Every surviving LVC_* switch is exercised on the device: the ones without a test of their own elsewhere.
  LVC_CONV_ENGINE=f32  -> kernels.CONV_ENGINE  (every conv / GEMM on the exact fp32 MFMA kernel)
  LVC_CHAIN=0          -> kernels.CHAIN        (conv3 -> next conv1 as two launches)
  LVC_BNECK=0          -> kernels.BNECK        (res2's bottleneck blocks as separate layers instead of one launch each)
  resnet.FUSE_STRIDED_PROJECTION, fpn.MERGE_OUTPUT_CONVS, rpn.MERGE_LEVELS / MERGE_LEVELS_CONV / FUSE_PREDICTOR (module constants)  (the RPN head's predictor / 3x3 conv over all levels as one launch each, then both as one)
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
    monkeypatch.setattr(K, "BNECK", False)                        # (res2's blocks as separate layers: the chains this test counts)
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
    monkeypatch.setattr(R, "FUSE_PREDICTOR", False)      # its own test below
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


def test_rpn_predictor_in_the_epilogue_of_the_head_conv(monkeypatch):
    """StandardRPNHead at inference as ONE launch (rpn.FUSE_PREDICTOR, kernels.conv3x3_levels_pred): every workgroup of the 3x3 kernel
    contracts its 128 hidden channels with the predictor's weights and adds the slice to the zeroed output atomically.  Against the two
    launches: one launch fewer, no hidden buffer, logits / deltas within the predictor's own summation-order tolerance (1e-6 of the
    output scale); no further from an fp64 evaluation of the head than the two launches are (x 1.25); two addends per element: the
    same bits on every run."""
    from lvc_amd import kernels as K
    from lvc_amd.modeling.proposal_generator import rpn as R
    from test_gpu_e2e import _model

    model = _model()
    head = model.proposal_generator.rpn_head
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():     # weights of a trained head's size (the default init's 0.01 leaves the logits at 1e-2)
        for l in (head.conv, head.objectness_logits, head.anchor_deltas):
            l.weight.copy_(torch.randn(l.weight.shape, generator=g) * (2.0 / (l.weight[0].numel()) ** 0.5))
            l.bias.copy_(torch.randn(l.bias.shape, generator=g) * 0.1)
    feats = [torch.randn(2, h, w, 256, generator=g).cuda() * 3 for h, w in ((104, 152), (52, 76), (26, 38), (13, 19), (7, 10))]
    res, n = {}, {}
    for fuse in (True, False):
        monkeypatch.setattr(R, "FUSE_PREDICTOR", fuse)
        timer = K.LaunchTimer()
        monkeypatch.setattr(K, "CONV_TIMER", timer)
        with torch.no_grad():
            res[fuse] = [t.clone() for t in head.forward_nhwc(feats)]
        monkeypatch.setattr(K, "CONV_TIMER", None)
        n[fuse] = len(timer.records)
    assert n[True] == 1 and n[False] == 2
    monkeypatch.setattr(R, "FUSE_PREDICTOR", True)
    with torch.no_grad():
        again = head.forward_nhwc(feats)
    wc, bc = head.conv.weight.detach().double(), head.conv.bias.detach().double()
    wp = torch.cat([head.objectness_logits.weight, head.anchor_deltas.weight], 0).detach().double()
    bp = torch.cat([head.objectness_logits.bias, head.anchor_deltas.bias], 0).detach().double()
    for a, b, a2, f in zip(res[True], res[False], again, feats):
        assert a.shape == b.shape == (2, f.shape[1], f.shape[2], 16) and a.is_contiguous()
        assert torch.equal(a, a2)
        x64 = f.double().permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(x64, wc, bc, padding=1)), wp, bp).permute(0, 2, 3, 1)
        scale = max(1.0, float(ref.abs().max()))
        assert float((a - b).abs().max()) <= 1e-6 * scale
        ea, eb = float((a[..., :15].double() - ref).abs().max()), float((b[..., :15].double() - ref).abs().max())
        assert ea <= 1.25 * eb + 1e-7 * scale, (ea, eb)
        assert float(a[..., 15].abs().max()) == 0.0          # the padding output stays zero
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


@pytest.mark.parametrize("hw", [(416, 608), (394, 602)])      # the second: odd extents in front of res3.0 (99 x 151) and res5.0 (25 x 38)
def test_strided_projection_shortcut_fused_with_conv3(monkeypatch, hw):
    """res4.0 / res5.0 (res3.0 keeps its chained launch): conv3 and the stride-2 projection shortcut as ONE pointwise GEMM over [conv2
    output | block input sampled at the even pixels] (resnet.FUSE_STRIDED_PROJECTION) against the two launches + residual: two conv launches fewer,
    features within the conv tolerance (the FrozenBN scales go into the fused weights: one more fp32 rounding per weight)."""
    from lvc_amd import kernels as K
    from lvc_amd.modeling.backbone import resnet as R
    from test_gpu_e2e import _model

    model = _model()
    x = torch.randn(2, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(1)).cuda() * 40
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


def test_bneck_switch_one_launch_per_res2_block(monkeypatch):
    """The trunk with res2's three blocks as ONE launch each (csrc/conv_bneck.hip, default) against the layer-by-layer form: the same
    features to a valid fp32 re-association, three `f16s1_bneck` launches, no res2 chain, and the block falling back (and staying
    there) once an activation leaves the fused kernel's range."""
    from lvc_amd import kernels as K
    from test_gpu_e2e import _model

    model = _model()
    x = torch.randn(2, 3, 416, 608, generator=torch.Generator().manual_seed(2)).cuda() * 40
    res, tags = {}, {}
    for on in (True, False):
        monkeypatch.setattr(K, "BNECK", on)
        timer = K.LaunchTimer()
        monkeypatch.setattr(K, "CONV_TIMER", timer)
        with torch.no_grad():
            res[on] = {k: v.clone() for k, v in model.backbone.bottom_up(x).items()}
        monkeypatch.setattr(K, "CONV_TIMER", None)
        tags[on] = [r[3] for r in timer.records]
    assert tags[True].count("f16s1_bneck") == 3 and "f16s1_bneck" not in tags[False], tags[True]
    assert tags[True].count("f16s1_chain") == 3 and tags[False].count("f16s1_chain") == 5      # res3's pairs / res2's + res3's
    assert len(tags[True]) < len(tags[False])
    for name in res[True]:
        a, b = res[True][name], res[False][name]
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 3e-5 * scale, (name, float((a - b).abs().max()) / scale)
    K.check_conv_error_word(x.device)
