"""Activation range of the two-way fp16 split kernels, per layer (kernels.check_conv_error_word).

The reference computes in fp32 (detectron2/layers/wrappers.py:41-99): no activation range.  Here a layer runs on the
single-accumulator fp16 split (|a| <= 4094), and when ITS range word is raised it alone moves to the two-accumulator form
(|a| <= 65504), then to the range-free bf16x3 kernels -- and stays there; every other layer keeps its kernels; the pass is
repeated; the words are cleared per pass.  Planted here: a 5000 in one channel of res4's output (consumers: res5.0.conv1, the
res5.0 projection shortcut -- fused with conv3 --, FPN lateral 4) and a 1e5 in one channel of res5's output (consumer: FPN lateral 5); the consumers'
weights for that channel are scaled down so that everything behind them stays O(1)."""
import time

import pytest
import torch

from helpers import match_fraction, r50_state_dict

pytestmark = pytest.mark.gpu

CH4, CH5 = 37, 1001       # the planted channels of res4 / res5 outputs


def _planted_state_dict():
    sd = {k: v.clone() for k, v in r50_state_dict().items()}
    bu = "backbone.bottom_up."
    # FrozenBN bias of the last conv3 of res4 / res5: the block output (after the shortcut add and ReLU) gains +5000 / +1e5
    sd[bu + "res4.5.conv3.norm.bias"][CH4] += 5000.0
    sd[bu + "res5.2.conv3.norm.bias"][CH5] += 1.0e5
    # consumers: keep what they make of that channel O(1)
    for k in (bu + "res5.0.conv1.weight", bu + "res5.0.shortcut.weight", "backbone.fpn_lateral4.weight"):
        sd[k][:, CH4] *= 1e-4
    sd["backbone.fpn_lateral5.weight"][:, CH5] *= 1e-6
    return sd


def _model(sd):
    from test_gpu_e2e import _model as base

    m = base()
    m.load_state_dict(sd, strict=True)
    return m


def _tiers(model):
    from lvc_amd.layers.wrappers import Conv2d, Linear

    return {n: mod._range_state["tier"] for n, mod in model.named_modules() if isinstance(mod, (Conv2d, Linear)) and mod._range_state["tier"]}


def test_only_the_overflowing_layers_are_rerouted_and_results_match_the_oracle():
    from lvc_amd import kernels as K
    from lvc_amd.utils import synthetic as syn
    from oracle import rcnn as orc

    sd = _planted_state_dict()
    model = _model(sd)
    # full-size images: on small maps (fewer than 2048 output pixels) these layers run on the bf16x3 kernels, which have no range
    inputs = [{"image": syn.synthetic_image(1), "height": 800, "width": 1333}, {"image": syn.synthetic_image(2), "height": 800, "width": 1333}]
    split0, epoch0 = K.CONV_SPLIT, K.RANGE_EPOCH
    with torch.no_grad():
        out = model(inputs)
    assert K.CONV_SPLIT == split0, "the process-wide fallback must not fire for a layer-local overflow"
    assert K.RANGE_EPOCH > epoch0
    tiers = _tiers(model)
    print("re-routed layers:", tiers)
    # (res5.0's projection shortcut runs fused with conv3 -- resnet.FUSE_STRIDED_PROJECTION -- as one packed layer with a range state
    # of its own: the planted channel arrives in its sampled-input half)
    assert tiers == {"backbone.bottom_up.res5.0.conv1": 1, "backbone.fpn_lateral4": 1, "backbone.fpn_lateral5": 2}, tiers
    assert model.backbone.bottom_up.res5[0]._fused_projection().state["tier"] == 1
    assert model.backbone.bottom_up.res4[0]._fused_projection().state["tier"] == 0
    # a second pass: nothing moves, nothing is repeated
    epoch1 = K.RANGE_EPOCH
    with torch.no_grad():
        out2 = model(inputs)
    assert K.RANGE_EPOCH == epoch1 and _tiers(model) == tiers
    ref = orc.generalized_rcnn_inference({k: v.cpu() for k, v in sd.items()}, orc.RCNNSpec(), inputs)
    for o, o2, r in zip(out, out2, ref):
        inst = o["instances"].to("cpu")
        assert torch.equal(inst.pred_boxes.tensor, o2["instances"].to("cpu").pred_boxes.tensor)
        frac, wb, ws = match_fraction(inst.pred_boxes.tensor, inst.scores, inst.pred_classes, r["pred_boxes"], r["scores"],
                                      r["pred_classes"], box_tol=0.1, score_tol=2e-3)
        print("planted model: %d detections, %.0f%% matched vs the CPU oracle (worst box %.1e px, score %.1e)" % (len(inst), 100 * frac, wb, ws))
        assert len(inst) == len(r["scores"]) and frac >= 0.9


def test_steady_state_speed_with_rerouted_layers():
    """Four re-routed layers cost a few percent at most: the rest of the network stays on its kernels."""
    from lvc_amd.utils import synthetic as syn

    batch = [{"image": syn.synthetic_image(1 + i), "height": 800, "width": 1333} for i in range(2)]

    def rate(model):
        with torch.no_grad():
            for _ in range(3):
                model.inference_batched(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                model.inference_batched(batch)
            torch.cuda.synchronize()
        return 10 * len(batch) / (time.perf_counter() - t0)

    clean = _model(r50_state_dict())
    with torch.no_grad():
        clean(batch)
    r_clean = rate(clean)
    planted = _model(_planted_state_dict())
    with torch.no_grad():
        planted(batch)            # re-routes
    assert len(_tiers(planted)) == 3 and planted.backbone.bottom_up.res5[0]._fused_projection().state["tier"] == 1      # + the fused conv3 / shortcut of res5.0
    r_planted = rate(planted)
    r_clean2 = rate(clean)
    print("img/s clean %.1f / %.1f, with four re-routed layers %.1f" % (r_clean, r_clean2, r_planted))
    assert r_planted >= 0.97 * min(r_clean, r_clean2)


def test_graphed_replay_reports_an_out_of_range_activation():
    """the captured kernels write the error words of the CAPTURE stream's workspace; `instances()` must read those."""
    from lvc_amd import kernels as K
    from lvc_amd.evaluation import GraphedInference
    from lvc_amd.utils import synthetic as syn

    model = _model(r50_state_dict())
    batch = [{"image": syn.synthetic_image(3, 240, 320), "height": 240, "width": 320}]
    g = GraphedInference(model, batch)
    g.replay()
    assert len(g.instances()) == 1
    # the same graph, an image that drives the stem's successors far out of range
    hot = [{"image": batch[0]["image"] * 1.0e6, "height": 240, "width": 320}]
    g.replay(hot)
    with pytest.raises(K.Fp16RangeError):
        g.instances()
    # the words were cleared with the report: the next in-range replay is clean again
    g.replay(batch)
    assert len(g.instances()) == 1


def test_grouped_launch_overflow_moves_its_layers():
    """kernels.conv3x3_levels: an activation beyond 4094 in ONE of the maps of a grouped launch raises the launch's range word; the
    shared layer (one PackedConv over all maps), or every layer of a launch with a layer per map, moves to two accumulators, the
    repeated launch is clean and equals the per-map launches."""
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(9)
    dev = torch.device("cuda:0")
    xs = [torch.randn(2, h, w, 256, generator=g).to(dev) for h, w in ((96, 128), (48, 64), (24, 32))]
    xs[1][1, 7, 9, 33] = 5000.0
    ws = [torch.randn(256, 256, 3, 3, generator=g).to(dev) * (2.0 / (9 * 256)) ** 0.5 for _ in range(3)]
    K.clear_conv_error_word(dev)
    for pcs in (K.pack_conv(ws[0], stride=1, pad=1), [K.pack_conv(w, stride=1, pad=1) for w in ws]):
        plist = pcs if isinstance(pcs, list) else [pcs]
        assert all(q.state["tier"] == 0 for q in plist)
        K.conv3x3_levels(xs, pcs, relu=True)
        with pytest.raises(K.Fp16RangeError) as ei:
            K.check_conv_error_word(dev)
        assert ei.value.rerouted and all(q.state["tier"] == 1 for q in plist)
        outs = K.conv3x3_levels(xs, pcs, relu=True)
        assert K.conv_error_word(dev) == 0
        for l, (x, o) in enumerate(zip(xs, outs)):
            ref = K.conv2d_nhwc(x, plist[l] if isinstance(pcs, list) else pcs, relu=True)
            assert float((o - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        assert K.conv_error_word(dev) == 0


def test_hidden_overflow_of_the_fused_predictor_moves_the_predictor_only():
    """kernels.conv3x3_levels_pred: a hidden value (act(conv), never written) beyond fp16 raises the POINTWISE layer's range word, not
    the 3x3 layer's: the pointwise layer moves to the range-free kernels, the pair runs as two launches from then on (the fused launch
    declines) and equals an fp64 evaluation."""
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(11)
    dev = torch.device("cuda:0")
    xs = [torch.randn(2, h, w, 256, generator=g).to(dev) * 40 for h, w in ((96, 128), (48, 64))]
    wc = torch.randn(256, 256, 3, 3, generator=g).to(dev) * 20.0         # hidden values: sigma = 40 x 20 x sqrt(2304) = 38400, many beyond 65504
    wp = torch.randn(16, 256, 1, 1, generator=g).to(dev) * 1e-4
    wp[15] = 0
    bp = torch.randn(16, generator=g).to(dev)
    bp[15] = 0
    pc, pred = K.pack_conv(wc, stride=1, pad=1), K.pack_conv(wp, bias=bp)
    pc.two_acc = True
    K.clear_conv_error_word(dev)
    outs = K.conv3x3_levels_pred(xs, pc, pred, relu=True)
    assert outs is not None
    with pytest.raises(K.Fp16RangeError) as ei:
        K.check_conv_error_word(dev)
    assert ei.value.rerouted and pred.state["tier"] == 2 and pc.state["tier"] == 0
    assert K.conv3x3_levels_pred(xs, pc, pred, relu=True) is None
    for x in xs:
        y = K.conv2d_nhwc(K.conv2d_nhwc(x, pc, relu=True), pred)
        ref = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wc.double(), padding=1)),
                                         wp.double(), bp.double()).permute(0, 2, 3, 1)
        assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert K.conv_error_word(dev) == 0


def test_msra_scale_checkpoint_rehearsal():
    """no trained checkpoint exists here, so the |a| <= 4094 envelope of the one-accumulator forms is rehearsed on
    checkpoint-SHAPED weights (lvc_amd.utils.synthetic.msra_checkpoint_rehearsal_: MSRA-initialised trunk behind identity FrozenBN,
    loaded through the pre-v3 path without running statistics): activations reach O(1e3..1e4) in res4 / res5.  The layers whose
    operands leave their form's range re-route (and only move wider), the passes settle, the second settled pass re-routes nothing,
    and the re-routed trunk agrees with the CPU oracle (fp32) on the same weights to the usual conv tolerance."""
    from lvc_amd import kernels as K
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn
    from oracle import rcnn as orc

    model = build_model(base_rcnn_fpn()).eval()
    syn.msra_checkpoint_rehearsal_(model)
    bn = model.backbone.bottom_up.res4[0].conv1.norm
    assert float(bn.running_mean.abs().max()) == 0.0 and abs(float(bn.running_var[0]) - 1.0) < 1e-6     # supplied by the loader
    inputs = [{"image": syn.synthetic_image(1), "height": 800, "width": 1333}]
    epoch0 = K.RANGE_EPOCH
    with torch.no_grad():
        model(inputs)                  # re-routes internally (run_with_fallbacks) until the pass is clean
        tiers = _tiers(model)
        epoch1 = K.RANGE_EPOCH
        model(inputs)
    assert K.RANGE_EPOCH == epoch1 and _tiers(model) == tiers, "a settled model must not move again"
    trunk = {n: t for n, t in tiers.items() if n.startswith("backbone.bottom_up")}
    print("re-routed: %d layers (%d of the trunk) in %d re-routing passes: %s" % (len(tiers), len(trunk), epoch1 - epoch0, tiers))
    assert trunk, "MSRA-scale activations must push some trunk layer beyond |a| <= 4094"
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        images = model.preprocess_image(inputs)
        feats = model.backbone.bottom_up(images.tensor)
        imgs, _ = orc.preprocess([inputs[0]["image"]], orc.RCNNSpec().pixel_mean, orc.RCNNSpec().pixel_std, 32)
        ref = orc.resnet(sd, imgs, 50)
    for name in ("res2", "res3", "res4", "res5"):
        scale = float(ref[name].abs().max())
        err = float((feats[name].cpu() - ref[name]).abs().max()) / scale
        print("   %s: scale %.3g, max |hip - oracle| / scale %.2e" % (name, scale, err))
        assert err <= 1e-4, (name, err)
    K.check_conv_error_word(torch.device("cuda:0"))
