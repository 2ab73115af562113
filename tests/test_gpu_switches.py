"""Every surviving LVC_* switch is exercised on the device: the ones without a test of their own elsewhere.
  LVC_CONV_ENGINE=f32  -> kernels.CONV_ENGINE  (every conv / GEMM on the exact fp32 MFMA kernel)
  LVC_CHAIN=0          -> kernels.CHAIN        (conv3 -> next conv1 as two launches)
  rpn.MERGE_LEVELS     (module constant)      (the RPN predictor of all levels as one launch)
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
    from test_gpu_e2e import _model

    model = _model()
    x = torch.randn(2, 3, 416, 608, generator=torch.Generator().manual_seed(1)).cuda() * 40
    res, tags = {}, {}
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


def test_rpn_predictor_of_all_levels_in_one_launch(monkeypatch):
    """StandardRPNHead: the five hidden maps in one buffer and ONE predictor launch over it (rpn.MERGE_LEVELS, default) against a
    predictor launch per level -- four launches fewer, the same products per pixel (the contraction of a tile may be split over
    workers at other points when the launch has other dimensions: fp32 summation order, 1e-6 of the output scale)."""
    from lvc_amd import kernels as K
    from lvc_amd.modeling.proposal_generator import rpn as R
    from test_gpu_e2e import _model

    model = _model()
    g = torch.Generator().manual_seed(2)
    feats = [torch.randn(2, h, w, 256, generator=g).cuda() for h, w in ((104, 152), (52, 76), (26, 38), (13, 19), (7, 10))]
    res, n = {}, {}
    for merge in (True, False):
        monkeypatch.setattr(R, "MERGE_LEVELS", merge)
        timer = K.LaunchTimer()
        monkeypatch.setattr(K, "CONV_TIMER", timer)
        with torch.no_grad():
            res[merge] = [t.clone() for t in model.proposal_generator.rpn_head.forward_nhwc(feats)]
        monkeypatch.setattr(K, "CONV_TIMER", None)
        n[merge] = len(timer.records)
    assert n[False] - n[True] == 4
    for a, b, f in zip(res[True], res[False], feats):
        assert a.shape == b.shape == (2, f.shape[1], f.shape[2], 16) and a.is_contiguous()
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))


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
