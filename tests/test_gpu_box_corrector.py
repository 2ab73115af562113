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


def _train_model(num_classes=80, freeze_backbone=True, depth=50):
    """cascade_ubbr_R_50_FPN_ft_all_30shot_aug_ftmore.yaml: 80 classes, frozen backbone, RBG + CascadeROIHeads
    (num_classes=60, freeze_backbone=False: cascade_ubbr_R_50_FPN_base.yaml, FREEZE_AT 2)."""
    from lvc_amd.config import set_global_cfg
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    cfg = base_rcnn_fpn(num_classes=num_classes, depth=depth)
    M = cfg.MODEL
    M.ROI_HEADS.NAME = "CascadeROIHeads"
    M.ROI_HEADS.OUTPUT_LAYER = "BoxOnlyLayersCascade"
    M.ROI_HEADS.PROPOSAL_APPEND_GT = False
    M.ROI_HEADS.POSITIVE_FRACTION = 1.0
    M.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 64
    M.ROI_HEADS.IOU_THRESHOLDS = [0.3]
    M.ROI_BOX_HEAD.NUM_FC = 3
    M.ROI_BOX_HEAD.DROPOUT = 0.0
    M.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
    M.ROI_BOX_CASCADE_HEAD.IOUS = (0.3, 0.5, 0.7)
    M.BACKBONE.FREEZE = freeze_backbone
    M.BACKBONE.FREEZE_AT = 2
    M.PROPOSAL_GENERATOR.NAME = "RBG"
    M.LOAD_PROPOSALS = True
    set_global_cfg(cfg)
    model = build_model(cfg)
    syn.conditioned_r50_fpn_(model, depth=depth)
    return model.train()


def _rbg_rows(monkeypatch, model, g, dev, rows):
    """rows = "lists": CascadeROIHeads.forward on per-image Instances (the reference's structure); "batched": the padded-table path
    (RBG.forward_batched -> CascadeROIHeads.forward_train_batched) fed the same recorded boxes, with a dropped junk row after every
    recorded one so that the keep mask has gaps."""
    if rows == "lists":
        monkeypatch.setattr(model.roi_heads, "batched_training", False)
        return

    def recorded_table(proposals, targets, gt, gt_off):
        per = [g["rbg_boxes%d" % i].to(dev) for i in range(len(targets))]
        P = 2 * max(len(p) for p in per) + 3
        table = torch.zeros(len(per), P, 4, device=dev)
        keep = torch.zeros(len(per), P, dtype=torch.bool, device=dev)
        for i, p in enumerate(per):
            table[i, 0: 2 * len(p): 2] = p
            table[i, 1: 2 * len(p): 2] = p.flip(0) + 7.0        # plausible boxes that RBG "dropped"
            keep[i, 0: 2 * len(p): 2] = True
        return table, keep

    monkeypatch.setattr(model.proposal_generator, "forward_batched", recorded_table)


@pytest.mark.parametrize("rows", ["lists", "batched"])
def test_box_corrector_training_step_matches_reference(monkeypatch, rows):
    """SURVEY row 20 with a frozen trunk: GeneralizedRCNN.forward (RBG branch) -> CascadeROIHeads training
    (label_and_sample_proposals, 3 x [pool -> 3 FC -> Linear(1024,4) -> decode -> clip/filter -> match]) ->
    BoxOnlyLayersCascade GIoU losses -> backward through the heads, vs the reference's CPU step
    (tests/golden/box_corrector_train.npz).  RBG's jitter comes from torch's CPU generator in the golden run, so the
    recorded RBG output is fed through a patched proposal generator; randperm is the identity on both sides."""
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn
    from lvc_amd.utils.events import EventStorage

    g = gold("box_corrector_train")
    model = _train_model()
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    assert len(trainable) == 24 and all(n.startswith("roi_heads.") for n in trainable)
    dev = torch.device("cuda:0")
    batch = []
    for i, (h, w, seed) in enumerate([(240, 320, 3), (200, 352, 4)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(g["gt_boxes%d" % i])
        inst.gt_classes = g["gt_classes%d" % i]
        props = Instances((h, w))
        props.proposal_boxes = Boxes(g["loaded_boxes%d" % i])
        props.objectness_logits = g["loaded_logits%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "proposals": props, "height": h, "width": w})

    def recorded_rbg(proposals, targets):
        out = []
        for i, t in enumerate(targets):
            p = Instances(t.image_size)
            p.proposal_boxes = Boxes(g["rbg_boxes%d" % i].to(dev))
            p.objectness_logits = g["rbg_logits%d" % i].to(dev)
            out.append(p)
        return out, {}

    monkeypatch.setattr(model.proposal_generator, "forward", recorded_rbg)
    _rbg_rows(monkeypatch, model, g, dev, rows)
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    with EventStorage(0) as storage:
        losses = model(batch)
        sum(losses.values()).backward()
    for k in ("loss_box_reg_stage0", "loss_box_reg_stage1", "loss_box_reg_stage2"):
        ref, got = float(g["loss." + k]), float(losses[k].detach())
        print(k, got, ref)
        assert abs(got - ref) <= 2e-4 * max(1.0, abs(ref)), k
    lat = storage.latest()
    for k in ("roi_head/num_fg_samples", "stage1/roi_head/num_fg_samples", "stage2/roi_head/num_fg_samples",
              "stage1/roi_head/num_bg_samples"):
        v = lat[k]
        v = v[0] if isinstance(v, tuple) else v
        assert float(v) == float(g["scalar." + k.replace("/", ".")]), k
    # Gradients.  The box predictors see no ReLU of their own and must agree entry by entry (1e-3 of the largest
    # entry; measured 1e-5).  The FC layers are compared through robust statistics because two effects of the fp32
    # noise floor of the trunk (tests/test_gpu_e2e.py) are discontinuous: a unit whose pre-activation lies within the
    # noise floor of zero falls on the other side of the ReLU than in the CPU run, which moves that unit's bias
    # gradient and its row of the weight gradient by one row's contribution (stage 0: 128 foreground rows, a few dozen
    # such units out of 131 072; stage 2: 2 foreground rows); and stage k's input boxes are stage k-1's outputs.  The
    # exact arithmetic of the backward (masking, column sums, both GEMMs, the fc1 column permutation) is pinned
    # without that noise by test_box_head_backward_matches_torch_autograd below.
    report = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        gflat = p.grad.flatten().cpu()
        s, nrm, stride = [float(v) for v in g["grad_stats." + name]]
        sample = gflat[:: int(stride)][:4096].double()
        ref = g["grad_sample." + name].double()
        scale = max(float(ref.abs().max()), 1e-12)
        ok = ((sample - ref).abs() <= 1e-3 * scale).float().mean().item()
        cos = float((sample * ref).sum() / (sample.norm() * ref.norm()).clamp_min(1e-30))
        nerr = abs(float(gflat.double().norm()) - nrm) / max(nrm, 1e-12)
        report[name] = (ok, cos, nerr)
        print("%-44s within tol %.4f  cos %.6f  norm err %.2e" % (name, ok, cos, nerr))
    for name, (ok, cos, nerr) in report.items():
        if "bbox_pred" in name:
            assert ok == 1.0 and nerr <= 1e-4, (name, ok, cos, nerr)
        else:
            # direction and size of every FC gradient; entrywise agreement is only demanded of stage 0, whose inputs
            # are one decode away from the given boxes (stages 1 / 2 see boxes that went through 1 / 2 earlier stages
            # and have 17 / 2 foreground rows: a 1e-5 change of the trunk features moves single entries by > 1e-3)
            assert cos >= 0.999 and nerr <= 3e-3, (name, ok, cos, nerr)
            if ".box_head.0." in name:
                assert ok >= 0.85, (name, ok)


@pytest.mark.parametrize("rows", ["lists", "batched"])
def test_box_corrector_training_step_with_trunk_matches_reference(monkeypatch, rows):
    """BASELINE config 5 proper (cascade_ubbr_R_50_FPN_base.yaml, FREEZE_AT 2): the same step with the trunk training
    from res3 up -- ROIAlign backward into p2..p5 (x 1/3, `_ScaleGradient`), FPN (3x3 output convs, laterals with the
    fused nearest-x2 add -> 2x2 down-sum), res5..res3 (wgrad kernel, dgrad on the forward kernels, stride-2 scatter,
    ReLU masks, FrozenBN scales) -- against the reference's CPU step (tests/golden/box_corrector_train_base.npz):
    82 trainable tensors, the 11 of stem + res2 frozen.  Robust metrics as above (ReLU units within the trunk's noise
    floor of zero flip); entrywise agreement is demanded where no such unit sits between loss and tensor."""
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn
    from lvc_amd.utils.events import EventStorage

    g = gold("box_corrector_train_base")
    model = _train_model(num_classes=60, freeze_backbone=False)
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert frozen == g["frozen_names"].tolist()
    dev = torch.device("cuda:0")
    batch = []
    for i, (h, w, seed) in enumerate([(240, 320, 3), (200, 352, 4)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(g["gt_boxes%d" % i])
        inst.gt_classes = g["gt_classes%d" % i]
        props = Instances((h, w))
        props.proposal_boxes = Boxes(g["loaded_boxes%d" % i])
        props.objectness_logits = g["loaded_logits%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "proposals": props, "height": h, "width": w})

    def recorded_rbg(proposals, targets):
        out = []
        for i, t in enumerate(targets):
            p = Instances(t.image_size)
            p.proposal_boxes = Boxes(g["rbg_boxes%d" % i].to(dev))
            p.objectness_logits = g["rbg_logits%d" % i].to(dev)
            out.append(p)
        return out, {}

    monkeypatch.setattr(model.proposal_generator, "forward", recorded_rbg)
    _rbg_rows(monkeypatch, model, g, dev, rows)
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    with EventStorage(0):
        losses = model(batch)
        sum(losses.values()).backward()
    for k in ("loss_box_reg_stage0", "loss_box_reg_stage1", "loss_box_reg_stage2"):
        ref, got = float(g["loss." + k]), float(losses[k].detach())
        print(k, got, ref)
        assert abs(got - ref) <= 2e-4 * max(1.0, abs(ref)), k
    worst = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        gflat = p.grad.flatten().cpu()
        s, nrm, stride = [float(v) for v in g["grad_stats." + name]]
        sample = gflat[:: int(stride)][:2048].double()
        ref = g["grad_sample." + name].double()
        if nrm == 0.0:   # no RoI of this batch is pooled from p4 / p5: their output convs get exactly zero
            assert float(gflat.abs().max()) == 0.0, name
            continue
        cos = float((sample * ref).sum() / (sample.norm() * ref.norm()).clamp_min(1e-30))
        nerr = abs(float(gflat.double().norm()) - nrm) / max(nrm, 1e-12)
        ok = float(((sample - ref).abs() <= 1e-3 * ref.abs().max() + 1e-2 * ref.abs()).double().mean())
        print("%-52s cos %.6f  norm err %.2e  entries within tolerance %.4f" % (name, cos, nerr, ok))
        worst[name] = (cos, nerr, ok)
    # A ReLU unit of the trunk or of a head that sits within fp32 noise of zero may flip against the reference's CPU run; one
    # flipped unit of one RoI rewrites a row / column of the FC gradients behind it.  Which tensor that hits depends on the
    # rounding of the kernels in use (the same step under LVC_CONV_SPLIT=bf16x3 reproduces box_head.2.fc3.weight to cos 1.000000,
    # under the default f16x2 split to 0.9965 with 99.95 % of its sampled entries unchanged): demand cos >= 0.998, or cos >= 0.99 with at
    # least 95 % of the sampled entries within tolerance; the norm within 1 % either way.
    bad = {n: v for n, v in worst.items() if not (v[1] <= 1e-2 and (v[0] >= 0.998 or (v[0] >= 0.99 and v[2] >= 0.95)))}
    assert not bad, bad


@pytest.mark.parametrize("fixture,sizes", [("box_corrector_train_r101", ((240, 320, 3), (200, 352, 4))),
                                           ("box_corrector_train_r101_800x1333", ((800, 1333, 5), (800, 1333, 6)))])
@pytest.mark.parametrize("rows", ["lists", "batched"])
def test_box_corrector_training_step_r101_matches_reference(monkeypatch, rows, fixture, sizes):
    """BASELINE config 5 as named: box-corrector training on R101-FPN (cascade_ubbr base yaml with RESNETS.DEPTH 101):
    133 trainable tensors (res4 has 23 blocks) against the reference's CPU step
    (tests/golden/box_corrector_train_r101.npz); same robust metrics as the R50 step.  _800x1333: the size and per-GPU batch the
    config is quoted on (two 3 x 800 x 1333 images; oracle/make_golden.py gen_box_corrector_train_r101_full), same bars."""
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn
    from lvc_amd.utils.events import EventStorage

    g = gold(fixture)
    model = _train_model(num_classes=60, freeze_backbone=False, depth=101)
    assert sum(1 for p in model.parameters() if p.requires_grad) == 133
    dev = torch.device("cuda:0")
    batch = []
    for i, (h, w, seed) in enumerate(sizes):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(g["gt_boxes%d" % i])
        inst.gt_classes = g["gt_classes%d" % i]
        props = Instances((h, w))
        props.proposal_boxes = Boxes(g["loaded_boxes%d" % i])
        props.objectness_logits = g["loaded_logits%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "proposals": props, "height": h, "width": w})

    def recorded_rbg(proposals, targets):
        out = []
        for i, t in enumerate(targets):
            p = Instances(t.image_size)
            p.proposal_boxes = Boxes(g["rbg_boxes%d" % i].to(dev))
            p.objectness_logits = g["rbg_logits%d" % i].to(dev)
            out.append(p)
        return out, {}

    monkeypatch.setattr(model.proposal_generator, "forward", recorded_rbg)
    _rbg_rows(monkeypatch, model, g, dev, rows)
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    with EventStorage(0):
        losses = model(batch)
        sum(losses.values()).backward()
    for k in ("loss_box_reg_stage0", "loss_box_reg_stage1", "loss_box_reg_stage2"):
        ref, got = float(g["loss." + k]), float(losses[k].detach())
        print(k, got, ref)
        assert abs(got - ref) <= 2e-4 * max(1.0, abs(ref)), k
    bad, worst = {}, (1.0, 0.0)
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        gflat = p.grad.flatten().cpu()
        s, nrm, stride = [float(v) for v in g["grad_stats." + name]]
        if nrm == 0.0:
            assert float(gflat.abs().max()) == 0.0, name
            continue
        sample = gflat[:: int(stride)][:1024].double()
        ref = g["grad_sample." + name].double()
        cos = float((sample * ref).sum() / (sample.norm() * ref.norm()).clamp_min(1e-30))
        nerr = abs(float(gflat.double().norm()) - nrm) / max(nrm, 1e-12)
        worst = (min(worst[0], cos), max(worst[1], nerr))
        # R101's noise floor is 5-6x R50's (tests/test_gpu_e2e.py); the last stage's head sees 2 foreground rows, so a
        # handful of hidden units on the other side of their ReLU turn single rows of its weight gradients
        lim = 0.98 if ".box_head.2." in name else 0.995
        if not (cos >= lim and nerr <= 2e-2):
            bad[name] = (cos, nerr)
    print("worst cosine %.6f, worst norm error %.2e over the trainable tensors" % worst)
    assert not bad, bad


def test_box_head_backward_matches_torch_autograd():
    """FastRCNNConvFCHead (3 FC, channels-last fc1) + Linear(1024,4): forward and every parameter gradient of
    sum(out * g) against torch's own autograd on the same fp32 inputs (no trunk in front, so no noise-floor effects):
    masks, column sums, dX / dW GEMMs and the fc1 (c,h,w) <-> (h,w,c) column permutation."""
    import torch.nn.functional as F
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.layers import ShapeSpec
    from lvc_amd.layers.wrappers import Linear
    from lvc_amd.modeling.roi_heads.box_head import build_box_head

    cfg = base_rcnn_fpn(num_classes=80)
    cfg.MODEL.ROI_BOX_HEAD.NUM_FC = 3
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(9)
    head = build_box_head(cfg, ShapeSpec(channels=256, height=7, width=7)).to(dev).train()
    pred = Linear(1024, 4).to(dev)
    with torch.no_grad():
        for p in list(head.parameters()) + list(pred.parameters()):
            p.copy_(torch.randn(p.shape, generator=gen) * (0.02 if p.dim() == 2 else 0.1))
    M = 77
    x = torch.randn(M, 256, 7, 7, generator=gen)
    gout = torch.randn(M, 4, generator=gen)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = pred(head.forward_nhwc(x_nhwc))
    (out * gout.to(dev)).sum().backward()
    # torch reference on CPU with the same parameters (reference layout: flatten (c,h,w))
    ps = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in list(head.named_parameters()) + [("pred." + n, p) for n, p in pred.named_parameters()]}
    h = x.flatten(1)
    for k in (1, 2, 3):
        h = F.relu(F.linear(h, ps["fc%d.weight" % k], ps["fc%d.bias" % k]))
    ref = F.linear(h, ps["pred.weight"], ps["pred.bias"])
    (ref * gout).sum().backward()
    assert (out.detach().cpu() - ref.detach()).abs().max() <= 2e-5 * float(ref.abs().max())
    mine = dict(list(head.named_parameters()) + [("pred." + n, p) for n, p in pred.named_parameters()])
    for n, p in ps.items():
        got = mine[n].grad.cpu()
        scale = float(p.grad.abs().max())
        bad = ((got - p.grad).abs() > 1e-4 * scale).float().mean().item()
        print(n, "max err / scale", float((got - p.grad).abs().max()) / scale, "fraction off", bad)
        assert bad <= 2e-4, n  # a pre-activation within fp32 rounding of zero may still flip its mask


def test_giou_box_loss_kernel_matches_torch_autograd():
    """lvc_giou_box_loss vs a plain torch restatement (apply_deltas + fvcore giou_loss, autograd) on random rows,
    including clamped log-sizes, disjoint boxes and background rows; both the cascade form and the `iterate` form."""
    from lvc_amd import kernels as k
    import math

    gen = torch.Generator().manual_seed(3)
    R, K = 257, 80
    p = torch.rand(R, 4, generator=gen) * 200
    p[:, 2:] = p[:, :2] + 5 + torch.rand(R, 2, generator=gen) * 150
    gt = p + torch.randn(R, 4, generator=gen) * 20
    gt[:, 2:] = torch.max(gt[:, 2:], gt[:, :2] + 2)
    gt[:20] += 500  # disjoint
    cls = torch.randint(0, K + 1, (R,), generator=gen)
    d = torch.randn(R, 4, generator=gen) * 2
    d[5, 2] = 80.0  # clamped
    w = (10.0, 10.0, 5.0, 5.0)
    clamp = math.log(1000.0 / 16)

    def ref_loss(dl, iterate, lam):
        fg = (cls >= 0) & (cls < K)
        pf, gf, df = p[fg], gt[fg], dl[fg]
        ww, hh = pf[:, 2] - pf[:, 0], pf[:, 3] - pf[:, 1]
        cx, cy = pf[:, 0] + 0.5 * ww, pf[:, 1] + 0.5 * hh
        dx, dy = df[:, 0] / w[0], df[:, 1] / w[1]
        dw, dh = torch.clamp(df[:, 2] / w[2], max=clamp), torch.clamp(df[:, 3] / w[3], max=clamp)
        pcx, pcy, pw, ph = dx * ww + cx, dy * hh + cy, torch.exp(dw) * ww, torch.exp(dh) * hh
        b = torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], 1)

        def giou(b1, b2, eps=1e-7):
            x1, y1, x2, y2 = b1.unbind(-1)
            x1g, y1g, x2g, y2g = b2.unbind(-1)
            xk1, yk1, xk2, yk2 = torch.max(x1, x1g), torch.max(y1, y1g), torch.min(x2, x2g), torch.min(y2, y2g)
            inter = torch.zeros_like(x1)
            m = (yk2 > yk1) & (xk2 > xk1)
            inter[m] = (xk2[m] - xk1[m]) * (yk2[m] - yk1[m])
            union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
            iou = inter / (union + eps)
            xc1, yc1, xc2, yc2 = torch.min(x1, x1g), torch.min(y1, y1g), torch.max(x2, x2g), torch.max(y2, y2g)
            area_c = (xc2 - xc1) * (yc2 - yc1)
            return 1 - (iou - (area_c - union) / (area_c + eps))

        la = giou(b, gf)
        if not iterate:
            return la.mean()
        return torch.maximum(la - giou(pf, gf).mul(lam), torch.zeros_like(la)).mean()

    dev = torch.device("cuda:0")
    for iterate, lam in ((False, 0.0), (True, 0.9)):
        dl = d.clone().requires_grad_(True)
        L = ref_loss(dl, iterate, lam)
        L.backward()
        out, dd = k.giou_box_loss(d.to(dev), p.to(dev), gt.to(dev), cls.to(dev), K, w, clamp, iterate=iterate, lambda_=lam)
        assert abs(float(out[0]) - float(L)) <= 1e-5 * max(1.0, abs(float(L)))
        assert (dd.cpu() - dl.grad).abs().max() <= 1e-5 * float(dl.grad.abs().max())


def test_box_corrector_evaluation_matches_reference(monkeypatch):
    """GeneralizedRCNN.inference with BoxOnlyLayersCascade (reference rcnn.py:201-230 -> CascadeROIHeads._forward_box
    reg_only evaluation): IoU with the matched GT before and after the three cascade stages, for the foreground
    proposals that detectron2's label_and_sample_proposals keeps (it subsamples with randperm even in eval: identity
    on both sides).  Same rows, same classes; IoUs within 1e-3 (three decode stages on the fp32 trunk)."""
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn

    g = gold("box_corrector_eval")
    t = gold("box_corrector_train")
    model = _train_model().eval()
    batch = []
    for i, (h, w, seed) in enumerate([(240, 320, 3), (200, 352, 4)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(t["gt_boxes%d" % i])
        inst.gt_classes = t["gt_classes%d" % i]
        props = Instances((h, w))
        props.proposal_boxes = Boxes(t["loaded_boxes%d" % i])
        props.objectness_logits = t["loaded_logits%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "proposals": props, "height": h, "width": w})
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    with torch.no_grad():
        out = model(batch)
    assert set(out) == {"input_ious", "output_ious", "gt_classes"}
    assert out["gt_classes"].cpu().tolist() == g["gt_classes"].tolist()
    assert (out["input_ious"].cpu() - g["input_ious"]).abs().max() <= 1e-5
    err = float((out["output_ious"].cpu() - g["output_ious"]).abs().max())
    print("max |IoU_out - ref|", err)
    assert err <= 1e-3


def test_standard_roi_heads_with_rbg_proposals_evaluates(monkeypatch):
    """reference roi_heads.py:561-562: StandardROIHeads built under PROPOSAL_GENERATOR.NAME = "RBG" labels and subsamples the
    given proposals against the ground truth before its usual box branch.  Equal to doing the two steps by hand."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn

    cfg = base_rcnn_fpn(num_classes=20)
    cfg.MODEL.PROPOSAL_GENERATOR.NAME = "RBG"
    cfg.MODEL.LOAD_PROPOSALS = True
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 32
    model = build_model(cfg).eval()
    syn.conditioned_r50_fpn_(model)
    heads = model.roi_heads
    assert heads.rbg
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    img = syn.synthetic_image(5, 256, 320).to(dev)
    gt = torch.tensor([[30.0, 40.0, 150.0, 200.0], [100.0, 60.0, 300.0, 220.0]])
    tgt = Instances((256, 320))
    tgt.gt_boxes = Boxes(gt.to(dev))
    tgt.gt_classes = torch.tensor([3, 7], device=dev)
    props = Instances((256, 320))
    props.proposal_boxes = Boxes((gt.repeat(12, 1) + torch.randn(24, 4, generator=g) * 6).to(dev))
    props.objectness_logits = torch.zeros(24, device=dev)
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")}))
    with torch.no_grad():
        images = model.preprocess_image([{"image": img}])
        feats = model.backbone(images.tensor)
        out, _ = heads(images, feats, [props], [tgt])
        sampled = heads.label_and_sample_proposals([props], [tgt], inference=True)
        heads.rbg = False
        ref, _ = heads(images, feats, sampled, None)
    assert len(out) == 1 and len(out[0]) == len(ref[0])
    assert torch.equal(out[0].pred_boxes.tensor, ref[0].pred_boxes.tensor) and torch.equal(out[0].scores, ref[0].scores)
    # the same through the model's own entry (reference rcnn.py:277-299: loaded proposals + GT -> RBG (identity in eval) ->
    # heads; results returned un-postprocessed because the heads are not UBBRROIHeads)
    heads.rbg = True
    with torch.no_grad():
        whole = model([{"image": img, "instances": tgt, "proposals": props, "height": 512, "width": 640}])
    assert isinstance(whole, list) and len(whole) == 1 and isinstance(whole[0], Instances)
    assert whole[0].image_size == (256, 320)          # not rescaled to height / width
    assert torch.equal(whole[0].pred_boxes.tensor, ref[0].pred_boxes.tensor)
    assert torch.equal(whole[0].scores, ref[0].scores) and torch.equal(whole[0].pred_classes, ref[0].pred_classes)
    with pytest.raises(KeyError):
        model([{"image": img, "proposals": props}])


def test_rbg_batched_table_equals_the_per_image_lists():
    """RBG.forward_batched (one padded table + a keep mask, no device->host read) holds, at the kept rows and in order, exactly the
    boxes RBG.forward returns per image for the same generator state (reference lvc/modeling/proposal_generator/rbg.py)."""
    from lvc_amd import kernels as K
    from lvc_amd.structures import Boxes, Instances

    g = gold("box_corrector_train")
    model = _train_model()
    dev = torch.device("cuda:0")
    targets, props = [], []
    for i, (h, w) in enumerate([(240, 320), (200, 352)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(g["gt_boxes%d" % i].to(dev))
        inst.gt_classes = g["gt_classes%d" % i].to(dev)
        targets.append(inst)
        p = Instances((h, w))
        p.proposal_boxes = Boxes(g["loaded_boxes%d" % i].to(dev))
        p.objectness_logits = g["loaded_logits%d" % i].to(dev)
        props.append(p)
    rbg = model.proposal_generator
    rbg.train()
    assert rbg.can_batch(props, targets)
    torch.manual_seed(123)
    lists, _ = rbg(props, targets)
    gt, gt_off, _n = K.cat_ground_truth(targets)
    torch.manual_seed(123)
    table, keep = rbg.forward_batched(props, targets, gt, gt_off)
    assert 0 < int(keep.sum()) < keep.numel()
    for i, inst in enumerate(lists):
        assert torch.equal(table[i][keep[i]], inst.proposal_boxes.tensor), i


def test_box_corrector_training_with_live_rbg_both_row_structures_and_an_image_without_ground_truth():
    """The box-corrector step with RBG's own random jitter (no recorded boxes): (1) the padded-table path and the per-image lists give
    finite losses of the same size for the same generator state (their proposals are the same boxes; the samplers draw differently,
    so only the magnitude is compared), two steps in a row (the second re-packs through the kept plan); (2) a batch in which one image
    has no ground truth cannot be batched (`can_batch` / `can_batch_train` are False) and still trains through the lists."""
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn
    from lvc_amd.utils.events import EventStorage

    g = gold("box_corrector_train")
    model = _train_model()
    batch = []
    for i, (h, w, seed) in enumerate([(240, 320, 3), (200, 352, 4)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(g["gt_boxes%d" % i])
        inst.gt_classes = g["gt_classes%d" % i]
        props = Instances((h, w))
        props.proposal_boxes = Boxes(g["loaded_boxes%d" % i])
        props.objectness_logits = g["loaded_logits%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "proposals": props, "height": h, "width": w})
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    out = {}
    for rows in (True, False):
        model.roi_heads.batched_training = rows
        vals = []
        for step in range(2):
            torch.manual_seed(77 + step)
            with EventStorage(0):
                losses = model(batch)
            assert set(losses) == {"loss_box_reg_stage0", "loss_box_reg_stage1", "loss_box_reg_stage2"}
            total = sum(losses.values())
            assert bool(torch.isfinite(total))
            opt.zero_grad()
            total.backward()
            assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.requires_grad)
            vals.append(float(total))
        out[rows] = vals
    model.roi_heads.batched_training = True
    for a, b in zip(out[True], out[False]):
        assert 0.5 < a / b < 2.0, (out[True], out[False])
    # one image without ground truth: per-image lists (the reference's structure handles it, roi_heads.py:236-262)
    empty = Instances((200, 352))
    empty.gt_boxes = Boxes(torch.zeros(0, 4))
    empty.gt_classes = torch.zeros(0, dtype=torch.int64)
    batch[1] = dict(batch[1], instances=empty)
    gts = [b["instances"].to(model.device) for b in batch]
    assert not model.roi_heads.can_batch_train(gts) and not model.proposal_generator.can_batch([b["proposals"] for b in batch], gts)
    with EventStorage(0):
        losses = model(batch)
    assert bool(torch.isfinite(sum(losses.values())))
