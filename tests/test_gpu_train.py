"""BASELINE config 3 (novel fine-tune, only box_predictor trains): one training forward/backward on the GPU vs the
golden produced by the reference on CPU (oracle/make_golden.py gen_train), with torch.randperm patched to the identity
on both sides so the sampled anchors / proposals are defined by position, not by the RNG stream."""
import pytest
import torch

from helpers import gold

pytestmark = pytest.mark.gpu


def _train_model():
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    cfg = base_rcnn_fpn(num_classes=20)
    cfg.MODEL.BACKBONE.FREEZE = True
    cfg.MODEL.PROPOSAL_GENERATOR.FREEZE = True
    cfg.MODEL.ROI_HEADS.FREEZE_FEAT = True
    model = build_model(cfg)
    syn.conditioned_r50_fpn_(model)
    return model.train()


_SMALL = ((240, 320, 3), (200, 352, 4))
_FULL = ((800, 1333, 5), (800, 1333, 6))      # the size config 3 is quoted on (oracle/make_golden.py gen_train_full)


def _batch(g, sizes=_SMALL):
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn

    batch = []
    for i, (h, w, seed) in enumerate(sizes):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(g["gt_boxes%d" % i])
        inst.gt_classes = g["gt_classes%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "height": h, "width": w})
    return batch


@pytest.mark.parametrize("fixture,sizes", [("train_novel_ft", _SMALL), ("train_novel_ft_800x1333", _FULL)])
def test_novel_finetune_step_matches_reference(monkeypatch, fixture, sizes):
    """Small images, and the full 800 x 1333 pyramid (268 569 anchors per image, every ROIAlign level): same bars."""
    from lvc_amd.utils.events import EventStorage

    g = gold(fixture)
    model = _train_model()
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 103525
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    with EventStorage(0) as storage:
        losses = model(_batch(g, sizes))
        sum(losses.values()).backward()
    for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"):
        ref = float(g["loss." + k])
        got = float(losses[k].detach())
        print(k, got, ref)
        assert abs(got - ref) <= 1e-4 * max(1.0, abs(ref)), k
    lat = storage.latest()
    assert lat["rpn/num_pos_anchors"] == float(g["scalar.rpn.num_pos_anchors"])
    assert lat["roi_head/num_fg_samples"] == float(g["scalar.roi_head.num_fg_samples"])
    for name, p in model.named_parameters():
        if p.requires_grad:
            ref = g["grad." + name]
            got = p.grad.cpu()
            rel = float((got - ref).norm() / ref.norm())
            print(name, "relative grad error", rel)
            assert rel <= 1e-3, name


def test_sgd_step_and_ddp_bucket_single_process():
    """The optimizer step the reference takes (torch SGD momentum 0.9, wd 1e-4) moves only the predictor, and the
    flattened gradient bucket is the 0.41 MB the survey measured."""
    from lvc_amd import distributed as D

    g = gold("train_novel_ft")
    model = _train_model()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.01, momentum=0.9, weight_decay=1e-4)
    before = [p.detach().clone() for p in params]
    frozen = model.backbone.fpn_output2.weight.detach().clone()
    losses = model(_batch(g))
    opt.zero_grad()
    sum(losses.values()).backward()
    assert D.allreduce_gradients_(params) == 103525 * 4
    opt.step()
    assert all(not torch.equal(a, b) for a, b in zip(before, params))
    assert torch.equal(frozen, model.backbone.fpn_output2.weight)
    with torch.no_grad():
        model.eval()
        out = model([{"image": _batch(g)[0]["image"]}])
    assert "instances" in out[0]


def _base_model():
    """faster_rcnn_R_50_FPN_base.yaml: 60 classes, FREEZE_AT 2, everything above res2 trains."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    cfg = base_rcnn_fpn(num_classes=60)
    model = build_model(cfg)
    syn.conditioned_r50_fpn_(model)
    return model.train()


@pytest.mark.parametrize("fixture,sizes", [("train_base", _SMALL), ("train_base_800x1333", _FULL)])
def test_base_detector_training_step_matches_reference(monkeypatch, fixture, sizes):
    """(_800x1333: the same step on two full-size images, oracle/make_golden.py gen_train_base_full; same bars.)
    The whole detector training (faster_rcnn_R_50_FPN_base.yaml): RPN losses -> fused predictor / 3x3 conv of the
    RPN head -> p2..p6 (p6 through LastLevelMaxPool's scatter); CE + smooth-L1 -> predictor -> 2-FC box head ->
    ROIAlign backward -> p2..p5; FPN; res5..res3.  72 trainable tensors against the reference's CPU step
    (tests/golden/train_base.npz).  Losses to 1e-4; gradients through robust metrics (direction and size): the same
    ReLU-flip noise as in test_gpu_box_corrector.py, plus proposals that the RPN's own fp32 noise reorders."""
    from lvc_amd.utils.events import EventStorage

    g = gold(fixture)
    model = _base_model()
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert frozen == g["frozen_names"].tolist()
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    with EventStorage(0) as storage:
        losses = model(_batch(g, sizes))
        sum(losses.values()).backward()
    for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"):
        ref, got = float(g["loss." + k]), float(losses[k].detach())
        print(k, got, ref)
        assert abs(got - ref) <= 2e-4 * max(1.0, abs(ref)), k
    lat = storage.latest()
    assert lat["rpn/num_pos_anchors"] == float(g["scalar.rpn.num_pos_anchors"])
    assert lat["roi_head/num_fg_samples"] == float(g["scalar.roi_head.num_fg_samples"])
    bad = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        gflat = p.grad.flatten().cpu()
        s, nrm, stride = [float(v) for v in g["grad_stats." + name]]
        sample = gflat[:: int(stride)][:2048].double()
        ref = g["grad_sample." + name].double()
        if nrm == 0.0:   # identity randperm samples the first anchors (all on p2) and no RoI is pooled from p5
            assert float(gflat.abs().max()) == 0.0, name
            continue
        cos = float((sample * ref).sum() / (sample.norm() * ref.norm()).clamp_min(1e-30))
        nerr = abs(float(gflat.double().norm()) - nrm) / max(nrm, 1e-12)
        print("%-52s cos %.6f  norm err %.2e" % (name, cos, nerr))
        if not (cos >= 0.998 and nerr <= 1e-2):
            bad[name] = (cos, nerr)
    assert not bad, bad


def test_ft_all_cosine_training_step_matches_reference(monkeypatch):
    """faster_rcnn_R_50_FPN_ft_all_30shot_aug_ftmore_dropout.yaml (frozen backbone; RPN head, box head and the
    CosineSimOutputLayers predictor train; DROPOUT 0 on both sides): 13 trainable tensors vs the reference's CPU step
    (tests/golden/train_ft_all.npz), incl. the in-place renormalised classification weight the forward leaves behind
    and the gradient through the input normalisation (lvc_rownorm_backward)."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn
    from lvc_amd.utils.events import EventStorage

    g = gold("train_ft_all")
    cfg = base_rcnn_fpn(num_classes=80)
    cfg.MODEL.ROI_HEADS.OUTPUT_LAYER = "CosineSimOutputLayers"
    cfg.MODEL.BACKBONE.FREEZE = True
    cfg.MODEL.ROI_BOX_HEAD.DROPOUT = 0.0
    model = build_model(cfg)
    syn.conditioned_r50_fpn_(model)
    model.train()
    assert [n for n, p in model.named_parameters() if p.requires_grad] == g["trainable_names"].tolist()
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    with EventStorage(0):
        losses = model(_batch(g))
        sum(losses.values()).backward()
    for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"):
        ref, got = float(g["loss." + k]), float(losses[k].detach())
        print(k, got, ref)
        assert abs(got - ref) <= 2e-4 * max(1.0, abs(ref)), k
    w = model.roi_heads.box_predictor.cls_score.weight.detach().cpu()
    assert float((w - g["cls_weight_after"]).abs().max()) <= 1e-6
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        gflat = p.grad.flatten().cpu()
        s, nrm, stride = [float(v) for v in g["grad_stats." + name]]
        sample = gflat[:: int(stride)][:2048].double()
        ref = g["grad_sample." + name].double()
        cos = float((sample * ref).sum() / (sample.norm() * ref.norm()).clamp_min(1e-30))
        nerr = abs(float(gflat.double().norm()) - nrm) / max(nrm, 1e-12)
        print("%-52s cos %.6f  norm err %.2e" % (name, cos, nerr))
        assert cos >= 0.999 and nerr <= 3e-3, (name, cos, nerr)


def test_box_head_dropout_is_applied_in_training_only():
    """ROI_BOX_HEAD.DROPOUT (the ft_all yaml's 0.5): after every FC's ReLU in training (reference box_head.py:82-91),
    identity in eval; the surviving units are scaled by 1/(1-p)."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.layers import ShapeSpec
    from lvc_amd.modeling.roi_heads.box_head import build_box_head

    cfg = base_rcnn_fpn(num_classes=80)
    cfg.MODEL.ROI_BOX_HEAD.DROPOUT = 0.5
    dev = torch.device("cuda:0")
    head = build_box_head(cfg, ShapeSpec(channels=256, height=7, width=7)).to(dev)
    x = torch.randn(64, 7, 7, 256, device=dev)
    head.eval()
    with torch.no_grad():
        e1, e2 = head.forward_nhwc(x), head.forward_nhwc(x)
    assert torch.equal(e1, e2)
    head.train()
    torch.manual_seed(3)
    t = head.forward_nhwc(x)
    zeros = float((t == 0).float().mean())
    assert zeros > 0.5 + 0.1     # ReLU zeros plus the dropped half of the rest
    t.sum().backward()
    assert head.fc1.weight.grad is not None and float(head.fc1.weight.grad.abs().sum()) > 0


def test_gradient_buckets_on_device_single_process():
    """GradientBuckets on the real training graph (world size 1: no collective, but the hooks still move every
    gradient into its bucket): gradients equal the plain backward's, and an SGD step on the bucket-resident
    gradients moves the parameters."""
    from lvc_amd import distributed as D
    from lvc_amd.utils.events import EventStorage

    g = gold("train_base")
    model = _base_model()
    params = [p for p in model.parameters() if p.requires_grad]
    torch.manual_seed(0)      # subsample_labels draws randperm: same samples in both runs
    with EventStorage(0):
        sum(model(_batch(g)).values()).backward()
    plain = [p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None
    buckets = D.GradientBuckets(params, bucket_bytes=32 << 20)
    assert len(buckets.buckets) >= 4
    torch.manual_seed(0)
    with EventStorage(0):
        sum(model(_batch(g)).values()).backward()
    nbytes = buckets.finish()
    assert nbytes == sum(p.numel() for p in params) * 4
    for p, ref in zip(params, plain):
        # atomics reorder the wgrad / ROIAlign-backward sums run to run: compare to 1e-4 of the tensor's scale
        assert float((p.grad - ref).abs().max()) <= 1e-4 * max(float(ref.abs().max()), 1e-12) + 1e-9
    opt = torch.optim.SGD(params, lr=0.01, momentum=0.9)
    before = params[0].detach().clone()
    opt.step()
    assert not torch.equal(before, params[0])
    buckets.remove()


def test_training_step_with_an_image_without_ground_truth():
    """Edge case of the training rows (the reference handles it in Matcher / label_and_sample_*: rpn.py:296-300,
    roi_heads.py:236-262): one image of the batch has no GT box.  Losses stay finite, every trainable tensor gets a
    finite gradient, and the empty image contributes only background samples."""
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn
    from lvc_amd.utils.events import EventStorage

    g = gold("train_base")
    model = _base_model()
    batch = _batch(g)
    empty = Instances((200, 352))
    empty.gt_boxes = Boxes(torch.zeros(0, 4))
    empty.gt_classes = torch.zeros(0, dtype=torch.int64)
    batch[1]["instances"] = empty
    torch.manual_seed(1)
    with EventStorage(0) as storage:
        losses = model(batch)
        sum(losses.values()).backward()
    for k, v in losses.items():
        assert torch.isfinite(v).all(), k
    for name, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
    lat = storage.latest()
    nfg = lat["roi_head/num_fg_samples"]
    nfg = nfg[0] if isinstance(nfg, tuple) else nfg
    assert 0 < float(nfg) <= 64     # only image 0 has foreground (at most 25 % of 512, halved by the empty image)


@pytest.mark.parametrize("numerics", ["two_accumulators_unchained", "default"])
def test_three_sgd_steps_follow_the_reference_trajectory(monkeypatch, numerics):
    """Optimizer step + re-packing of every trainable weight (forward operand, split planes, data-gradient operand):
    four forward passes with three torch SGD steps in between against the reference's CPU trajectory
    (tests/golden/train_base_steps.npz).  A stale packed copy anywhere would freeze that layer; the classification
    loss drops 5.43 -> 0.21 and the RPN losses by 2x per step, so following that trajectory pins the whole update path.  Later steps compound the
    fp32 noise of earlier ones (and the proposals the updated RPN emits), hence the widening tolerance."""
    from lvc_amd.utils.events import EventStorage

    g = gold("train_base_steps")
    model = _base_model()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.0002, momentum=0.9, weight_decay=1e-4)
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    ref = g["losses"]
    # ROI-head losses depend on WHICH proposals the (updated, fp32-noisy) RPN ranks first -- near-ties reorder the sampled
    # set from the second update on (measured: 0.4 % / 0.8 % after one step, 6 % / 11 % after two); the RPN losses are
    # evaluated on the fixed anchor set and stay within 2 %
    # Round 4: the tolerance after ONE update was 2 % -- what the round-3 kernels happened to give (0.4 % / 0.8 %).  It is not a property
    # of the update path but of which valid fp32 rounding the trunk runs on: a rounding-level change of the frozen res2 features (6e-7
    # of their scale: conv3 -> conv1 as one launch, csrc/conv_pw_chain.hip) or of the accumulator form of every layer (5e-6) moves the
    # gradients of res3 / res4 by 0.9 - 1.3 % through ReLU masks that flip (scripts/dbg_train2.py: identical losses to 1e-6, head
    # gradients to 1e-5, res3 weight gradients 1.3e-2 apart, run-to-run 9e-7) and the loss after one update by 0.1 - 5.4 % / 0.1 - 8.6 %.
    # What the test pins is the update path: a stale packed copy leaves loss_cls at 5.43 instead of 0.21 (2 400 %).
    # the loose bar (2x the default kernels' measured drift) is kept for the default-kernel variant only; the variant
    # that pins the trunk's arithmetic to the round-3 forms (main + cross accumulators everywhere, conv3 and conv1 as two launches)
    # holds the first update to the 2 % those kernels give (0.4 % / 0.8 % measured), so a regression of a few percent in the update
    # or re-packing path fails here.
    from lvc_amd import kernels as K

    if numerics == "two_accumulators_unchained":
        monkeypatch.setattr(K, "CHAIN", False)
        monkeypatch.setattr(K, "HALO_S1", 1)
        monkeypatch.setattr(K, "PW_S1", 1)
        tol_roi = [2e-4, 0.02, 0.25, 2.0]
    else:
        tol_roi = [2e-4, 0.15, 0.35, 2.0]    # third update: the sampled RoI set has drifted (seen 18 % .. 51 % run to run)
    tol_rpn = [2e-4, 0.005, 0.05, 0.10]
    with EventStorage(0):
        for step in range(4):
            losses = model(_batch(g))
            got = [float(losses[k].detach()) for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc")]
            dev = [abs(a - float(b)) / max(abs(float(b)), 1e-3) for a, b in zip(got, ref[step])]
            print("step", step, ["%.5f" % v for v in got], "relative deviation", ["%.1e" % d for d in dev])
            assert max(dev[:2]) <= tol_roi[step] and max(dev[2:]) <= tol_rpn[step], (step, got, ref[step].tolist())
            opt.zero_grad()
            sum(losses.values()).backward()
            opt.step()


def test_loss_scaler_backward_matches_plain_backward(monkeypatch):
    """lvc_amd.solver.LossScaler: loss x 2^10, data gradients on the two-way fp16 kernels, gradients x 2^-10 -- the
    parameter gradients must equal the plain (bf16x3 data-gradient) backward to the fp32 noise of the step, the scale
    factors being exact; an absurd scale must be caught by the kernels' range word, skip the step and halve the scale."""
    from lvc_amd import kernels as K
    from lvc_amd.solver import LossScaler
    from lvc_amd.utils.events import EventStorage

    g = gold("train_base")
    model = _base_model()
    params = [p for p in model.parameters() if p.requires_grad]
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    with EventStorage(0):
        sum(model(_batch(g)).values()).backward()
    plain = [p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None
    opt = torch.optim.SGD(params, lr=0.0)
    scaler = LossScaler(init_scale=2.0 ** 10)
    with EventStorage(0):
        scaler.backward(sum(model(_batch(g)).values()))
    assert K.DGRAD_SPLIT == "bf16x3"
    assert scaler.step(opt) is True
    worst = 0.0
    for p, ref in zip(params, plain):
        if float(ref.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0
            continue
        rel = float((p.grad - ref).norm() / ref.norm())
        worst = max(worst, rel)
    print("worst relative gradient difference, scaled f16x2 vs plain bf16x3 backward: %.2e" % worst)
    assert worst <= 2e-3
    # overflow: 2^40 x (gradients of order 1e-4..1) leaves fp16's range inside the data-gradient kernels
    for p in params:
        p.grad = None
    big = LossScaler(init_scale=2.0 ** 40, max_scale=2.0 ** 40)
    before = params[0].detach().clone()
    with EventStorage(0):
        big.backward(sum(model(_batch(g)).values()))
    assert big.step(torch.optim.SGD(params, lr=0.1)) is False
    assert big.scale_value == 2.0 ** 39 and big.skipped_steps == 1 and torch.equal(before, params[0])
    assert K.conv_error_word(params[0].device) == 0


@pytest.mark.parametrize("sizes", [[(333, 500), (320, 480)], [(97, 131)], [(224, 200), (160, 333), (250, 90)]])
def test_training_step_on_ragged_image_sizes(sizes):
    """Backward through the trunk on maps whose sizes are odd at some level (the stride-2 scatter, the 2x2 down-sum of
    the FPN and LastLevelMaxPool's scatter all depend on them) and on batches of 1 and 3 images: finite losses, a finite
    gradient for every trainable tensor, and the same gradients again from a second identical step (only the atomics'
    summation order may differ)."""
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn
    from lvc_amd.utils.events import EventStorage

    model = _base_model()
    g = torch.Generator().manual_seed(len(sizes))
    batch = []
    for i, (h, w) in enumerate(sizes):
        n = 3
        x1 = torch.rand(n, generator=g) * (w * 0.5)
        y1 = torch.rand(n, generator=g) * (h * 0.5)
        boxes = torch.stack([x1, y1, x1 + 20 + torch.rand(n, generator=g) * (w * 0.4), y1 + 20 + torch.rand(n, generator=g) * (h * 0.4)], 1)
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(boxes)
        inst.gt_classes = torch.randint(0, 60, (n,), generator=g)
        batch.append({"image": syn.synthetic_image(40 + i, h, w), "instances": inst, "height": h, "width": w})
    params = [p for p in model.parameters() if p.requires_grad]
    grads = []
    for rep in range(2):
        for p in params:
            p.grad = None
        torch.manual_seed(3)
        with EventStorage(0):
            losses = model(batch)
            sum(losses.values()).backward()
        assert all(torch.isfinite(v).all() for v in losses.values())
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
        grads.append([p.grad.detach().clone() for p in params])
    for a, b in zip(*grads):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale + 1e-12


def test_proposal_network_training_equals_rpn_part_of_the_detector(monkeypatch):
    """ProposalNetwork (reference rcnn.py:413-488) in training returns the RPN losses; with the detector's weights they and
    the gradients of the RPN head / FPN / trunk they induce equal the detector's RPN terms (golden: train_base.npz holds
    the full detector's step, whose RPN losses do not depend on the RoI heads)."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn
    from lvc_amd.utils.events import EventStorage

    g = gold("train_base")
    det = _base_model()
    cfg = base_rcnn_fpn(num_classes=60)
    cfg.MODEL.META_ARCHITECTURE = "ProposalNetwork"
    net = build_model(cfg).train()
    own = net.state_dict()
    net.load_state_dict({k: v for k, v in det.state_dict().items() if k in own}, strict=True)
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    with EventStorage(0):
        losses = net(_batch(g))
        assert set(losses) == {"loss_rpn_cls", "loss_rpn_loc"}
        sum(losses.values()).backward()
    for k in ("loss_rpn_cls", "loss_rpn_loc"):
        assert abs(float(losses[k].detach()) - float(g["loss." + k])) <= 2e-4 * max(1.0, abs(float(g["loss." + k])))
    for name in ("proposal_generator.rpn_head.conv.weight", "proposal_generator.rpn_head.objectness_logits.bias",
                 "proposal_generator.rpn_head.anchor_deltas.weight"):
        p = dict(net.named_parameters())[name]
        s, nrm, stride = [float(v) for v in g["grad_stats." + name]]
        assert abs(float(p.grad.double().norm()) - nrm) <= 1e-3 * nrm, name     # the RPN head sees only the RPN losses
    assert dict(net.named_parameters())["backbone.fpn_output2.weight"].grad is not None
    net.eval()
    with torch.no_grad():
        out = net([{"image": _batch(g)[0]["image"], "height": 240, "width": 320}])
    assert "proposals" in out[0] and len(out[0]["proposals"]) > 0


def _random_proposals_and_targets(g, B=4, num_classes=20):
    from lvc_amd.structures import Boxes, Instances

    D = "cuda:0"
    props, tgts = [], []
    for i in range(B):
        h, w = 300 + 20 * i, 400
        ng = 2 + i
        xy = torch.rand(ng, 2, generator=g) * torch.tensor([w * 0.5, h * 0.5])
        wh = 30 + torch.rand(ng, 2, generator=g) * torch.tensor([w * 0.4, h * 0.4])
        gt = torch.cat([xy, xy + wh], 1)
        n = 900 + 37 * i
        near = gt[torch.randint(0, ng, (n // 3,), generator=g)] + 8 * torch.randn(n // 3, 4, generator=g)   # high-IoU proposals
        xy2 = torch.rand(n - n // 3, 2, generator=g) * torch.tensor([w * 0.7, h * 0.7])
        far = torch.cat([xy2, xy2 + 10 + torch.rand(n - n // 3, 2, generator=g) * 120], 1)
        boxes = torch.cat([near, far])[torch.randperm(n, generator=g)]
        p = Instances((h, w))
        p.proposal_boxes = Boxes(boxes.to(D))
        p.objectness_logits = torch.randn(n, generator=g).to(D)
        t = Instances((h, w))
        t.gt_boxes = Boxes(gt.to(D))
        t.gt_classes = torch.randint(0, num_classes, (ng,), generator=g).to(D)
        props.append(p)
        tgts.append(t)
    return props, tgts


def test_batched_proposal_sampling_equals_the_per_image_loop(monkeypatch):
    """`ROIHeads._label_and_sample_batched` (one device read per batch) against the per-image loop (the reference's steps one by
    one): identical Instances when `torch.randperm` is the identity (the parity tests' convention); with the real RNG the
    sample keeps the reference's invariants (512 per image, at most 128 foreground, no proposal twice, labels untouched)."""
    from lvc_amd.utils.events import EventStorage

    model = _train_model()
    heads = model.roi_heads
    g = torch.Generator().manual_seed(12)
    props, tgts = _random_proposals_and_targets(g)
    with EventStorage(0), monkeypatch.context() as mp:
        mp.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
        a = heads._label_and_sample_batched(props, tgts, True)
        b = heads._label_and_sample_loop(props, tgts, False, None)
    for x, y in zip(a, b):
        assert len(x) == len(y) == heads.batch_size_per_image
        for f in ("objectness_logits", "gt_classes"):
            assert torch.equal(x.get(f), y.get(f)), f
        assert torch.equal(x.proposal_boxes.tensor, y.proposal_boxes.tensor) and torch.equal(x.gt_boxes.tensor, y.gt_boxes.tensor)
    with EventStorage(0):
        torch.manual_seed(3)
        c = heads._label_and_sample_batched(props, tgts, True)
        d = heads._label_and_sample_batched(props, tgts, True)
    assert not all(torch.equal(x.proposal_boxes.tensor, y.proposal_boxes.tensor) for x, y in zip(c, d)), "two draws, same sample"
    K_ = heads.num_classes
    for x, ref in zip(c, b):
        fg = x.gt_classes != K_
        assert len(x) == 512 and int(fg.sum()) <= 128 and int(fg.sum()) == int((ref.gt_classes != K_).sum())
        assert bool((x.gt_classes[: int(fg.sum())] != K_).all()), "foreground first"
        assert torch.unique(x.proposal_boxes.tensor, dim=0).shape[0] == 512


def test_batched_anchor_sampling_equals_the_per_image_loop(monkeypatch):
    """`RPN._subsample_batched` against `subsample_labels` image by image: identical labels with the identity permutation; with
    the real RNG 256 labelled anchors per image, at most 128 positive, all drawn from the matching labels."""
    from lvc_amd.modeling.sampling import subsample_labels

    model = _train_model()
    rpn = model.proposal_generator
    g = torch.Generator().manual_seed(8)
    B, R = 3, 50000
    raw = torch.full((B, R), -1, dtype=torch.int8)
    raw[torch.rand(B, R, generator=g) < 0.7] = 0
    raw[torch.rand(B, R, generator=g) < 0.004] = 1
    raw[2] = torch.where(torch.rand(R, generator=g) < 0.0005, torch.ones(R, dtype=torch.int8), raw[2].clamp(max=0))   # few positives
    raw = raw.to("cuda:0")
    with monkeypatch.context() as mp:
        mp.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
        got = rpn._subsample_batched(raw.clone())
        want = []
        for i in range(B):
            lab = raw[i].clone()
            p, n = subsample_labels(lab, rpn.batch_size_per_image, rpn.positive_fraction, 0)
            lab.fill_(-1)
            lab.scatter_(0, p, 1)
            lab.scatter_(0, n, 0)
            want.append(lab)
    assert torch.equal(got, torch.stack(want))
    torch.manual_seed(1)
    r = rpn._subsample_batched(raw.clone())
    assert ((r >= 0).sum(1) == rpn.batch_size_per_image).all() and ((r == 1).sum(1) <= 128).all()
    assert bool(((r == 1) <= (raw == 1)).all()) and bool(((r == 0) <= (raw == 0)).all())
    assert int((r[2] == 1).sum()) == int((raw[2] == 1).sum())     # fewer positives than the cap: all of them are kept


@pytest.mark.parametrize("B,N,bs,cap,p_pos,p_neg,ident", [(3, 50000, 256, 128, 0.004, 0.7, True), (3, 50000, 256, 128, 0.004, 0.7, False),
                                                           (2, 2100, 512, 128, 0.2, 0.6, False), (4, 300, 512, 128, 0.1, 0.5, True),
                                                           (2, 2200000, 256, 128, 0.0001, 0.9, False), (2, 5000, 256, 128, 0.0, 1.0, False),
                                                           (2, 5000, 64, 64, 0.5, 0.0, False)])
def test_subsample_kernel_equals_the_reference_selection(B, N, bs, cap, p_pos, p_neg, ident):
    """csrc/train_targets.hip `lvc_subsample_batched` against subsample_labels restated with sorts (reference sampling.py:10-54 with
    the permutation's keys as priorities): per row the min(#pos, cap) positives and min(#neg, bs - num_pos) negatives with the
    smallest keys, by increasing key; identity keys (the parity tests' randperm) and random ones; fewer candidates than wanted, no
    positives, no negatives, more than 2^22 keys (three radix passes)."""
    from lvc_amd import kernels as k

    g = torch.Generator().manual_seed(N + bs)
    lab = torch.full((B, N), -1, dtype=torch.int8)
    u = torch.rand(B, N, generator=g)
    lab[u < p_neg] = 0
    lab[torch.rand(B, N, generator=g) < p_pos] = 1
    keys = torch.arange(B * N).view(B, N) if ident else torch.randperm(B * N, generator=g).view(B, N)
    d = torch.device("cuda:0")
    sel, cnt = k.subsample_batched(lab.to(d), keys.to(d), cap, bs)
    sel, cnt = sel.cpu(), cnt.cpu()
    for b in range(B):
        pos, neg = (lab[b] == 1).nonzero().view(-1), (lab[b] == 0).nonzero().view(-1)
        npos = min(len(pos), cap)
        nneg = min(len(neg), bs - npos)
        want_p = pos[keys[b][pos].argsort()][:npos]
        want_n = neg[keys[b][neg].argsort()][:nneg]
        assert cnt[b].tolist() == [npos, nneg]
        assert sel[b, :npos].tolist() == want_p.tolist()
        assert sel[b, npos:npos + nneg].tolist() == want_n.tolist()
        assert bool((sel[b, npos + nneg:] == -1).all())


def test_match_boxes_batched_equals_the_per_image_matcher():
    """`lvc_match_boxes_batched` against `lvc_match_boxes` image by image: shared boxes (the anchors) and per-image tables with ragged
    row counts, an image without ground truth, both Matcher configurations of the detector (RPN: two thresholds + low-quality matches,
    ROI heads: one threshold)."""
    from lvc_amd import kernels as k
    from lvc_amd.structures import Boxes, Instances

    g = torch.Generator().manual_seed(5)
    d = torch.device("cuda:0")
    B, N = 4, 20000
    xy = torch.rand(N, 2, generator=g) * 600
    boxes = torch.cat([xy, xy + 8 + torch.rand(N, 2, generator=g) * 200], 1).to(d)
    tg = []
    for b, ng in enumerate((3, 0, 7, 1)):
        t = Instances((800, 800))
        q = torch.rand(ng, 2, generator=g) * 500
        t.gt_boxes = Boxes(torch.cat([q, q + 20 + torch.rand(ng, 2, generator=g) * 250], 1).to(d))
        tg.append(t)
    gt, gt_off, _ = k.cat_ground_truth(tg)
    for thr, labs, lq in (([0.3, 0.7], [0, -1, 1], True), ([0.5], [0, 1], False)):
        m, l = k.match_boxes_batched(gt, gt_off, B, boxes, None, thr, labs, lq)
        per = torch.stack([boxes + 3.0 * b for b in range(B)])
        nbox = torch.tensor([N, N - 17, 5, N // 2], dtype=torch.int32, device=d)
        m2, l2 = k.match_boxes_batched(gt, gt_off, B, per, nbox, thr, labs, lq)
        for b, t in enumerate(tg):
            G = len(t)
            if G:
                rm, rl, _ = k.match_boxes(t.gt_boxes.tensor, boxes, thr, labs, lq)
                assert torch.equal(m[b].long(), rm) and torch.equal(l[b], rl)
                nb = int(nbox[b])
                rm2, rl2, _ = k.match_boxes(t.gt_boxes.tensor, per[b, :nb].contiguous(), thr, labs, lq)
                assert torch.equal(m2[b, :nb].long(), rm2) and torch.equal(l2[b, :nb], rl2)
                assert bool((l2[b, nb:] == -1).all()) and bool((m2[b, nb:] == 0).all())
            else:
                assert bool((l[b] == labs[0]).all()) and bool((m[b] == 0).all())


@pytest.mark.parametrize("which", ["frozen_rpn", "trainable_rpn"])
def test_batched_targets_equal_the_per_image_path(monkeypatch, which):
    """The training forward with the batch-wide target kernels (RPN.batched_targets / ROIHeads.batched_targets: five launches each,
    no device->host read before the counts) against the round-4 path (per-image Matcher calls, torch topk / argsort sampling): with
    the identity permutation the same anchors and proposals are sampled, so the four losses agree to the fp32 summation order of
    the loss kernels and the gradients of the trainable tensors to their scale."""
    from lvc_amd.modeling.proposal_generator.rpn import RPN
    from lvc_amd.modeling.roi_heads.roi_heads import ROIHeads
    from lvc_amd.utils.events import EventStorage

    g = gold("train_novel_ft" if which == "frozen_rpn" else "train_base")
    model = _train_model() if which == "frozen_rpn" else _base_model()
    params = [p for p in model.parameters() if p.requires_grad]
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    res = {}
    for fast in (False, True):
        monkeypatch.setattr(RPN, "batched_targets", fast)
        monkeypatch.setattr(ROIHeads, "batched_targets", fast)
        for p in params:
            p.grad = None
        with EventStorage(0) as st:
            losses = model(_batch(g))
            sum(losses.values()).backward()
            scal = {k: (v[0] if isinstance(v, tuple) else v) for k, v in st.latest().items() if k.startswith(("rpn/", "roi_head/"))}
        res[fast] = ({k: float(v.detach()) for k, v in losses.items()}, [p.grad.detach().clone() for p in params], scal)
    print(res[False][0], res[True][0], res[True][2])
    for k_ in res[False][0]:
        assert abs(res[True][0][k_] - res[False][0][k_]) <= 2e-6 * max(1.0, abs(res[False][0][k_])), k_
    for k_ in ("rpn/num_pos_anchors", "rpn/num_neg_anchors", "roi_head/num_fg_samples", "roi_head/num_bg_samples"):
        assert float(res[True][2][k_]) == float(res[False][2][k_]), k_
    for a, b in zip(res[True][1], res[False][1]):
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-12) + 1e-9


def test_sampler_with_generated_keys_draws_valid_distinct_samples():
    """`lvc_subsample_batched` with keys = NULL (the production path: a four-round Feistel bijection of b N + i keyed by a seed from torch's
    CPU generator, nothing sorted or read): the counts are subsample_labels', every selected index carries the right label and appears
    once, two seeds give different samples, one seed the same sample, and over many draws every candidate is taken about equally often."""
    from lvc_amd import kernels as k

    g = torch.Generator().manual_seed(3)
    B, N, bs, cap = 3, 50000, 256, 128
    lab = torch.full((B, N), -1, dtype=torch.int8)
    lab[torch.rand(B, N, generator=g) < 0.6] = 0
    lab[torch.rand(B, N, generator=g) < 0.01] = 1
    lab[2][lab[2] == 1] = -1
    lab[2, :40] = 1                       # fewer positives than the cap
    d = torch.device("cuda:0")
    labd = lab.to(d)
    s1, c1 = k.subsample_batched(labd, None, cap, bs, seed=12345)
    s1b, _ = k.subsample_batched(labd, None, cap, bs, seed=12345)
    s2, _ = k.subsample_batched(labd, None, cap, bs, seed=99991)
    assert torch.equal(s1, s1b) and not torch.equal(s1, s2)
    s1, c1 = s1.cpu(), c1.cpu()
    for b in range(B):
        npos = min(int((lab[b] == 1).sum()), cap)
        nneg = min(int((lab[b] == 0).sum()), bs - npos)
        assert c1[b].tolist() == [npos, nneg]
        idx = s1[b, :npos + nneg].long()
        assert len(torch.unique(idx)) == npos + nneg
        assert bool((lab[b][idx[:npos]] == 1).all()) and bool((lab[b][idx[npos:]] == 0).all())
    # uniformity: 400 positives, 100 taken per draw, 600 draws -> each expected 150 times (binomial sd ~ 10.6)
    lab2 = torch.full((1, 30000), 0, dtype=torch.int8)
    pos = torch.randperm(30000, generator=g)[:400]
    lab2[0, pos] = 1
    hits = torch.zeros(30000)
    lab2d = lab2.to(d)
    for seed in range(600):
        s, _ = k.subsample_batched(lab2d, None, 100, 256, seed=seed * 7919 + 13)
        hits[s[0, :100].long().cpu()] += 1
    h = hits[pos]
    assert float(hits.sum()) == 600 * 100 and float(h.sum()) == 600 * 100
    assert 100 <= float(h.min()) and float(h.max()) <= 200 and abs(float(h.std()) - 10.6) < 3.0, (float(h.min()), float(h.max()), float(h.std()))


def test_deferred_read_training_forward_falls_back_when_an_image_misses_its_quota(monkeypatch):
    """GeneralizedRCNN._forward_train_deferred runs the heads padded and checks the row counts in its one device->host read: with
    fewer candidates than BATCH_SIZE_PER_IMAGE (here: 40 proposals kept per image by the RPN against a quota of 512) the per-image path
    recomputes the detector losses, which must then equal the `deferred_reads = False` step."""
    from lvc_amd.modeling.meta_arch.rcnn import GeneralizedRCNN
    from lvc_amd.utils.events import EventStorage

    g = gold("train_novel_ft")
    model = _train_model()
    model.proposal_generator.post_nms_topk = (model.proposal_generator.post_nms_topk[0], 40)
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    res = {}
    for fast in (False, True):
        monkeypatch.setattr(GeneralizedRCNN, "deferred_reads", fast)
        with EventStorage(0) as st:
            losses = model(_batch(g))
            res[fast] = ({k: float(v.detach()) for k, v in losses.items()}, {k: (v[0] if isinstance(v, tuple) else v) for k, v in st.latest().items()})
    print(res)
    for k_ in res[False][0]:
        assert abs(res[True][0][k_] - res[False][0][k_]) <= 2e-6 * max(1.0, abs(res[False][0][k_])), k_
    assert float(res[True][1]["roi_head/num_bg_samples"]) == float(res[False][1]["roi_head/num_bg_samples"]) < 512
