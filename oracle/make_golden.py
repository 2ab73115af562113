"""Generate tests/golden/*.npz from the IMPORTED REFERENCE (runs only in the build container, where
/root/reference exists; the reference has no tests or golden vectors of its own -- SURVEY.md section 4).
Each fixture holds seeded inputs and the reference's outputs; nothing of the reference's source text is
stored.  Re-run:  python -m oracle.make_golden   (about 2 minutes on 8 cores).
TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import refshim  # noqa: E402

refshim.install()

from lvc_amd.utils import synthetic as syn  # noqa: E402


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-40s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def build_ref_model(yaml, opts=()):
    from lvc.config import get_cfg, set_global_cfg
    from lvc.modeling import build_model

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(refshim.REF_ROOT, "configs", yaml))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu"] + list(opts))
    cfg.freeze()
    set_global_cfg(cfg)
    model = build_model(cfg).eval()
    return cfg, model


def gen_roi_align():
    from detectron2 import _C

    g = torch.Generator().manual_seed(0)
    feat = torch.randn(2, 8, 50, 84, generator=g)
    n = 64
    cx, cy = torch.rand(n, generator=g) * 1333, torch.rand(n, generator=g) * 800
    w, h = 2 + torch.rand(n, generator=g) * 900, 2 + torch.rand(n, generator=g) * 700
    rois = torch.stack([torch.randint(0, 2, (n,), generator=g).float(), (cx - w / 2).clamp(0, 1333),
                        (cy - h / 2).clamp(0, 800), (cx + w / 2).clamp(0, 1333), (cy + h / 2).clamp(0, 800)], 1)
    extra = torch.tensor([[0, 10, 10, 10, 10], [1, 0, 0, 1333, 800], [0, 1300, 780, 1333, 800],
                          [1, 0, 0, 0.5, 0.5], [0, 5, 5, 5, 300], [1, 1332.5, 799.5, 1333, 800]])
    rois = torch.cat([rois, extra])
    outs = {}
    for scale in (0.25, 0.125, 0.0625, 0.03125):
        for aligned in (True, False):
            for sr in (0, 2):
                key = "out_s%g_a%d_sr%d" % (scale, aligned, sr)
                outs[key] = _C.roi_align_forward(feat, rois, scale, 7, 7, sr, aligned)
    # backward (csrc/vision.cpp:97): one upstream gradient, the reference's own CPU kernel, three configurations
    grad = torch.randn(rois.shape[0], 8, 7, 7, generator=g)
    for scale, aligned, sr in ((0.25, True, 0), (0.0625, True, 2), (0.125, False, 0)):
        outs["bwd_s%g_a%d_sr%d" % (scale, aligned, sr)] = _C.roi_align_backward(
            grad, rois, scale, 7, 7, 2, 8, 50, 84, sr, aligned)
    save("roi_align", feat=feat, rois=rois, grad=grad, **outs)


def gen_nms():
    from detectron2.layers import batched_nms

    g = torch.Generator().manual_seed(1)
    n = 5000
    base = torch.rand(400, 4, generator=g)
    base = torch.stack([base[:, 0] * 1200, base[:, 1] * 700, base[:, 0] * 1200 + 8 + base[:, 2] * 300,
                        base[:, 1] * 700 + 8 + base[:, 3] * 300], 1)
    boxes = base[torch.randint(0, 400, (n,), generator=g)] + torch.randn(n, 4, generator=g) * 4
    boxes[:, 2:] = torch.max(boxes[:, 2:], boxes[:, :2] + 0.5)
    scores = torch.randn(n, generator=g)
    lvl = torch.randint(0, 5, (n,), generator=g)
    keep = batched_nms(boxes, scores, lvl, 0.7)
    keep5 = batched_nms(boxes, scores, lvl, 0.5)
    save("nms", boxes=boxes, scores=scores, idxs=lvl, keep_thr07=keep, keep_thr05=keep5)


def gen_box_ops():
    from detectron2.modeling.anchor_generator import DefaultAnchorGenerator
    from detectron2.modeling.box_regression import Box2BoxTransform
    from detectron2.modeling.poolers import assign_boxes_to_levels
    from detectron2.structures import Boxes
    from detectron2.layers import ShapeSpec

    ag = DefaultAnchorGenerator(sizes=[[32], [64], [128], [256], [512]], aspect_ratios=[[0.5, 1.0, 2.0]],
                                strides=[4, 8, 16, 32, 64], offset=0.0)
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    anchors = ag([torch.zeros(1, 1, h, w) for h, w in shapes])
    d = {}
    for i, a in enumerate(anchors):
        t = a.tensor
        d["lvl%d_first" % i] = t[:8]
        d["lvl%d_last" % i] = t[-8:]
        d["lvl%d_sum" % i] = t.double().sum(0)
        d["lvl%d_n" % i] = np.int64(len(t))
    d["cell_anchors"] = torch.stack([b for b in ag.cell_anchors])
    save("anchors", **d)

    g = torch.Generator().manual_seed(2)
    boxes = torch.rand(300, 4, generator=g) * 600
    boxes[:, 2:] = boxes[:, :2] + 1 + torch.rand(300, 2, generator=g) * 500
    deltas1 = torch.randn(300, 4, generator=g) * 0.5
    deltas1[:5, 2:] = 9.0  # exercises the exp clamp
    deltas80 = torch.randn(300, 320, generator=g) * 2.0
    out1 = Box2BoxTransform(weights=(1.0, 1.0, 1.0, 1.0)).apply_deltas(deltas1, boxes)
    out80 = Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0)).apply_deltas(deltas80, boxes)
    small = torch.rand(300, 4, generator=g) * 100
    small[:, 2:] = small[:, :2] + torch.rand(300, 2, generator=g).pow(3) * 1300
    small = torch.cat([small, torch.tensor([[0, 0, 224, 224], [0, 0, 112, 112], [0, 0, 448, 448], [5, 5, 5, 5.]])])
    lv = assign_boxes_to_levels([Boxes(small)], 2, 5, 224, 4)
    save("box_ops", boxes=boxes, deltas1=deltas1, deltas80=deltas80, out1=out1, out80=out80, lvl_boxes=small,
         levels=lv)


def gen_rpn_and_det_ops():
    from detectron2.modeling.proposal_generator.proposal_utils import find_top_rpn_proposals
    from lvc.modeling.roi_heads.fast_rcnn import fast_rcnn_inference_single_image
    from detectron2.modeling.postprocessing import detector_postprocess
    from oracle import rcnn as orc

    g = torch.Generator().manual_seed(3)
    shapes = [(40, 64), (20, 32), (10, 16), (5, 8), (3, 4)]
    strides = [4, 8, 16, 32, 64]
    N, A = 2, 3
    cell = [orc.generate_cell_anchors((s,), (0.5, 1.0, 2.0)) for s in (32, 64, 128, 256, 512)]
    anchors = orc.grid_anchors(cell, shapes, strides)
    logits = [torch.randn(N, h * w * A, generator=g) * 3 for h, w in shapes]
    deltas = [torch.randn(N, h * w * A, 4, generator=g) * 0.4 for h, w in shapes]
    props = [orc.apply_deltas(d.reshape(-1, 4), a.unsqueeze(0).expand(N, -1, -1).reshape(-1, 4), (1., 1., 1., 1.)).view(N, -1, 4)
             for a, d in zip(anchors, deltas)]
    sizes = [(160, 250), (150, 256)]
    res = find_top_rpn_proposals(props, logits, sizes, 0.7, 300, 200, 0.0, False)
    d = {}
    for i in range(5):
        d["logits%d" % i] = logits[i]
        d["deltas%d" % i] = deltas[i]
    for n in range(N):
        d["out_boxes%d" % n] = res[n].proposal_boxes.tensor
        d["out_logits%d" % n] = res[n].objectness_logits
    save("rpn_proposals", shapes=np.array(shapes), image_sizes=np.array(sizes), **d)

    R, K = 400, 20
    pb = torch.rand(R, 4, generator=g) * 300
    pb[:, 2:] = pb[:, :2] + 4 + torch.rand(R, 2, generator=g) * 200
    pd = torch.randn(R, 4 * K, generator=g) * 1.5
    cl = torch.randn(R, K + 1, generator=g) * 2.0
    boxes = orc.apply_deltas(pd.view(-1, 4), pb.unsqueeze(1).expand(-1, K, 4).reshape(-1, 4), (10., 10., 5., 5.)).view(R, 4 * K)
    probs = torch.softmax(cl, -1)
    raw = boxes.clone()  # the reference clips its `boxes` argument in place (fast_rcnn.py:110-112)
    inst, rows = fast_rcnn_inference_single_image(boxes, probs, (300, 420), 0.05, 0.5, 100)
    post = detector_postprocess(inst, 450, 640)
    save("fast_rcnn_inference", proposals=pb, deltas=pd, cls_logits=cl, image_size=np.array([300, 420]),
         out_boxes=post.pred_boxes.tensor, raw_boxes=raw, probs=probs,
         out_scores=post.scores, out_classes=post.pred_classes, kept_rows=rows, out_hw=np.array([450, 640]))


def gen_blocks():
    from detectron2.modeling.backbone.resnet import BottleneckBlock
    from detectron2.layers import FrozenBatchNorm2d

    torch.manual_seed(4)
    blk = BottleneckBlock(64, 128, bottleneck_channels=32, stride=2, norm="FrozenBN", stride_in_1x1=True).eval()
    for m in blk.modules():
        if isinstance(m, FrozenBatchNorm2d):
            m.weight.copy_(torch.rand_like(m.weight) + 0.5)
            m.bias.copy_(torch.randn_like(m.bias) * 0.1)
            m.running_mean.copy_(torch.randn_like(m.running_mean) * 0.1)
            m.running_var.copy_(torch.rand_like(m.running_var) + 0.5)
    x = torch.randn(2, 64, 20, 28)
    with torch.no_grad():
        y = blk(x)
    save("bottleneck", x=x, y=y, **{"sd." + k: v for k, v in blk.state_dict().items()})

    # the res2 shapes that run as ONE launch (lvc_amd/csrc/conv_bneck.hip): the stage head with its stride-1 projection shortcut
    # (64 -> 64 -> 256) and an identity block (256 -> 64 -> 256), on a map with ragged tile edges (13 x 37: not multiples of 8 / 32)
    for name, cin in (("bottleneck_res2_proj", 64), ("bottleneck_res2_identity", 256)):
        torch.manual_seed(40 + cin)
        blk = BottleneckBlock(cin, 256, bottleneck_channels=64, stride=1, norm="FrozenBN", stride_in_1x1=True).eval()
        for m in blk.modules():
            if isinstance(m, FrozenBatchNorm2d):
                m.weight.copy_(torch.rand_like(m.weight) + 0.5)
                m.bias.copy_(torch.randn_like(m.bias) * 0.1)
                m.running_mean.copy_(torch.randn_like(m.running_mean) * 0.1)
                m.running_var.copy_(torch.rand_like(m.running_var) + 0.5)
        x = torch.randn(2, cin, 13, 37).relu_()
        with torch.no_grad():
            y = blk(x)
        save(name, x=x, y=y, **{"sd." + k: v for k, v in blk.state_dict().items()})


def gen_e2e():
    from detectron2.layers import FrozenBatchNorm2d

    cfg, model = build_ref_model("COCO-detection/faster_rcnn_R_50_FPN_base.yaml", ["MODEL.ROI_HEADS.NUM_CLASSES", 80])
    sd0 = syn.conditioned_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd0, strict=True)
    img1 = syn.synthetic_image(1)
    inp1 = [{"image": img1, "height": 800, "width": 1333}]
    calib = syn.calibrate_frozen_bn_(model, lambda: model(inp1), FrozenBatchNorm2d)
    save("r50_bn_calibration", **calib)
    # key order + shapes of the reference state_dict: the drop-in contract for checkpoints
    sd = model.state_dict()
    save("r50_fpn_state_dict_keys", keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))

    def run(inputs):
        with torch.no_grad():
            images = model.preprocess_image(inputs)
            feats = model.backbone(images.tensor)
            props, _ = model.proposal_generator(images, feats, None)
            out = model(inputs)
        return images, feats, props, out

    img2 = syn.synthetic_image(2)
    inputs = [{"image": img1, "height": 800, "width": 1333}, {"image": img2, "height": 800, "width": 1333}]
    images, feats, props, out = run(inputs)
    d = {}
    for k, v in feats.items():
        d["feat_" + k] = v[:, ::16, ::8, ::8].contiguous()  # sparse sample, enough to localise a diff
        d["featstat_" + k] = torch.stack([v.mean(), v.std(), v.abs().max()])
    for i in range(2):
        d["prop_boxes%d" % i] = props[i].proposal_boxes.tensor
        d["prop_logits%d" % i] = props[i].objectness_logits
        inst = out[i]["instances"]
        d["det_boxes%d" % i] = inst.pred_boxes.tensor
        d["det_scores%d" % i] = inst.scores
        d["det_classes%d" % i] = inst.pred_classes
    save("e2e_r50_fpn_800x1333", **d)
    for i in range(2):
        print("  image", i, "proposals", len(props[i]), "detections", len(out[i]["instances"]),
              "score range", float(out[i]["instances"].scores.max()), float(out[i]["instances"].scores.min()))

    # small case (fast on CPU): 2 different-size images, exercises padding + output rescale
    a = syn.synthetic_image(3, 240, 320)
    b = syn.synthetic_image(4, 200, 352)
    inputs = [{"image": a, "height": 480, "width": 640}, {"image": b, "height": 200, "width": 352}]
    images, feats, props, out = run(inputs)
    d = {}
    for k, v in feats.items():
        d["feat_" + k] = v[:, ::8].contiguous()
    for i in range(2):
        d["prop_boxes%d" % i] = props[i].proposal_boxes.tensor
        d["prop_logits%d" % i] = props[i].objectness_logits
        inst = out[i]["instances"]
        d["det_boxes%d" % i] = inst.pred_boxes.tensor
        d["det_scores%d" % i] = inst.scores
        d["det_classes%d" % i] = inst.pred_classes
    save("e2e_r50_fpn_small", **d)
    for i in range(2):
        print("  small image", i, "proposals", len(props[i]), "detections", len(out[i]["instances"]))


def gen_fpn_avg():
    """MODEL.FPN.FUSE_TYPE "avg" (reference fpn.py:133-134; no shipped yaml selects it): the pyramid of the small two-image batch of
    gen_e2e through the reference backbone with averaged top-down fusion, same weights and FrozenBN calibration."""
    cfg, model = build_ref_model("COCO-detection/faster_rcnn_R_50_FPN_base.yaml", ["MODEL.ROI_HEADS.NUM_CLASSES", 80, "MODEL.FPN.FUSE_TYPE", "avg"])
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r50_bn_calibration.npz")).items()}
    model.load_state_dict(syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib), strict=True)
    inputs = [{"image": syn.synthetic_image(3, 240, 320), "height": 480, "width": 640},
              {"image": syn.synthetic_image(4, 200, 352), "height": 200, "width": 352}]
    with torch.no_grad():
        feats = model.backbone(model.preprocess_image(inputs).tensor)
    d = {}
    for k, v in feats.items():
        d["feat_" + k] = v[:, ::16, ::2, ::2].contiguous()
        d["featstat_" + k] = torch.stack([v.mean(), v.std(), v.abs().max()])
    save("fpn_avg_small", **d)


def gen_knn():
    from detectron2.structures import Instances
    from tools.run_nearest_neighbours import get_nn_class_confirmatory, run_nearest_neighbours

    g = torch.Generator().manual_seed(5)
    S, D, C = 240, 64, 12
    centers = torch.randn(C, D, generator=g)
    shot_classes = torch.arange(C).repeat_interleave(S // C)
    shots = centers[shot_classes] + 0.8 * torch.randn(S, D, generator=g) + 0.5
    counts, descs, dcls = [], [], []
    for i in range(50):
        nq = int(torch.randint(0, 6, (1,), generator=g))
        cls = torch.randint(0, C, (nq,), generator=g)
        counts.append(nq)
        descs.append(centers[cls] + 0.9 * torch.randn(nq, D, generator=g) + 0.5)
        dcls.append(cls)

    def make_queries():
        qs = []
        for d, c in zip(descs, dcls):
            inst = Instances((10, 10))
            inst.crop_feats = d.clone()
            inst.gt_classes = c.clone()
            qs.append({"instances": inst})
        return qs

    res = {}
    for tag, cosine in (("cos", True), ("l2", False)):
        qs = run_nearest_neighbours(shot_classes, shots, make_queries(), cosine)
        get_nn_class_confirmatory(qs, 10)
        res["top10_" + tag] = torch.cat([q["instances"].top10_shots.reshape(-1, 10) for q in qs])
        res["keep_" + tag] = torch.cat([q["instances"].keep for q in qs])
    save("knn", shots=shots, shot_classes=shot_classes, counts=np.array(counts), q_desc=torch.cat(descs),
         q_classes=torch.cat(dcls), **res)


def gen_train(name="train_novel_ft", sizes=((240, 320, 3), (200, 352, 4))):
    """BASELINE config 3 (novel fine-tune): loss dict + the 4 trainable gradients for a fixed 2-image batch.
    torch.randperm is patched to the identity permutation so that the sampled anchors / proposals are the FIRST
    num_pos positives and num_neg negatives (the product test applies the same patch)."""
    from detectron2.structures import Boxes, Instances
    from detectron2.utils.events import EventStorage

    cfg, model = build_ref_model("COCO-detection/faster_rcnn_R_50_FPN_ft_novel_30shot.yaml")
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r50_bn_calibration.npz")).items()}
    sd = syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib)
    model.load_state_dict(sd, strict=True)
    model.train()
    g = torch.Generator().manual_seed(11)
    batch = []
    gts = {}
    for i, (h, w, seed) in enumerate(sizes):
        n = 5 + i
        k = h / 240.0 if h > 400 else 1.0       # (the full-size batch: boxes over all pyramid levels, 100 .. 500 px)
        x1 = torch.rand(n, generator=g) * (w - 80 * k)
        y1 = torch.rand(n, generator=g) * (h - 80 * k)
        bw = (30 + torch.rand(n, generator=g) * 120) * k
        bh = (30 + torch.rand(n, generator=g) * 100) * k
        boxes = torch.stack([x1, y1, (x1 + bw).clamp(max=w), (y1 + bh).clamp(max=h)], 1)
        classes = torch.randint(0, 20, (n,), generator=g)
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(boxes)
        inst.gt_classes = classes
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "height": h, "width": w})
        gts["gt_boxes%d" % i] = boxes
        gts["gt_classes%d" % i] = classes
    real = torch.randperm
    torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
    try:
        with EventStorage(0) as storage:
            losses = model(batch)
            total = sum(losses.values())
            total.backward()
            scalars = {k: float(v[0]) if isinstance(v, tuple) else float(v) for k, v in storage.latest().items()}
    finally:
        torch.randperm = real
    grads = {"grad." + n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    assert len(grads) == 4
    print("  losses", {k: float(v) for k, v in losses.items()}, scalars)
    save(name, **gts, **{"loss." + k: v.detach() for k, v in losses.items()}, **grads,
         **{"scalar." + k.replace("/", "."): np.float64(v) for k, v in scalars.items()})


def gen_train_full():
    """Config 3 at the size it is quoted on (two 800 x 1333 images; BASELINE.json configs[2] uses eight per GPU): the same step
    as gen_train on the full pyramid -- 268 569 anchors per image, all four ROIAlign levels in use."""
    gen_train("train_novel_ft_800x1333", ((800, 1333, 5), (800, 1333, 6)))


def gen_train_base(name="train_base", source="train_novel_ft", sizes=((240, 320, 3), (200, 352, 4))):
    """faster_rcnn_R_50_FPN_base.yaml (the base detector training of BASELINE config 3's first stage: 60 classes,
    FREEZE_AT 2, RPN + box head + predictor + FPN + res3..res5 all train): one training step on the same 2-image
    batch as gen_train (read back from its fixture).  Stored: the four losses, event scalars, and per trainable
    tensor the gradient's sum, L2 norm and a strided sample; randperm = identity."""
    from detectron2.structures import Boxes, Instances
    from detectron2.utils.events import EventStorage

    cfg, model = build_ref_model("COCO-detection/faster_rcnn_R_50_FPN_base.yaml")
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r50_bn_calibration.npz")).items()}
    sd = syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib)
    model.load_state_dict(sd, strict=True)
    model.train()
    t = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, source + ".npz")).items()}
    batch, d = [], {}
    for i, (h, w, seed) in enumerate(sizes):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(t["gt_boxes%d" % i])
        inst.gt_classes = t["gt_classes%d" % i]      # < 20 < 60
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "height": h, "width": w})
        d["gt_boxes%d" % i], d["gt_classes%d" % i] = inst.gt_boxes.tensor, inst.gt_classes
    real = torch.randperm
    torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
    try:
        with EventStorage(0) as storage:
            losses = model(batch)
            sum(losses.values()).backward()
            scalars = {k: float(v[0]) if isinstance(v, tuple) else float(v) for k, v in storage.latest().items()}
    finally:
        torch.randperm = real
    ntrain, frozen = 0, []
    for n_, p_ in model.named_parameters():
        if p_.requires_grad:
            ntrain += 1
            gflat = p_.grad.flatten()
            stride = max(1, gflat.numel() // 2048) | 1   # odd: never a whole number of rows of a 2^k-wide matrix
            d["grad_sample." + n_] = gflat[::stride][:2048].clone()
            d["grad_stats." + n_] = torch.tensor([float(gflat.double().sum()), float(gflat.double().norm()), float(stride)], dtype=torch.float64)
        else:
            frozen.append(n_)
    print("  losses", {k: float(v.detach()) for k, v in losses.items()}, "trainable tensors", ntrain, "frozen", len(frozen), scalars)
    d["frozen_names"] = np.array(frozen)
    save(name, **d, **{"loss." + k: v.detach() for k, v in losses.items()},
         **{"scalar." + k.replace("/", "."): np.float64(v) for k, v in scalars.items()})


def gen_train_base_full():
    """The base detector's training step on two 3 x 800 x 1333 images (GT of gen_train_full): 72 trainable tensors under the full
    pyramid -- the size the reference trains at."""
    gen_train_base("train_base_800x1333", "train_novel_ft_800x1333", ((800, 1333, 5), (800, 1333, 6)))


def gen_train_base_steps():
    """Three consecutive SGD steps (torch.optim.SGD, lr 2e-4 -- small enough that the trajectory is smooth and one step's fp32 noise is not amplified --, momentum 0.9, weight decay 1e-4) of
    faster_rcnn_R_50_FPN_base.yaml on the gen_train batch: the four losses before every step.  Pins the optimizer step +
    re-packing of every trainable weight (stale packed copies would freeze the losses)."""
    from detectron2.structures import Boxes, Instances
    from detectron2.utils.events import EventStorage

    cfg, model = build_ref_model("COCO-detection/faster_rcnn_R_50_FPN_base.yaml")
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r50_bn_calibration.npz")).items()}
    model.load_state_dict(syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib), strict=True)
    model.train()
    t = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "train_novel_ft.npz")).items()}
    batch = []
    for i, (h, w, seed) in enumerate([(240, 320, 3), (200, 352, 4)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(t["gt_boxes%d" % i])
        inst.gt_classes = t["gt_classes%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "height": h, "width": w})
    params = [p_ for p_ in model.parameters() if p_.requires_grad]
    opt = torch.optim.SGD(params, lr=0.0002, momentum=0.9, weight_decay=1e-4)
    real = torch.randperm
    torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
    rows = []
    try:
        with EventStorage(0):
            for step in range(4):
                losses = model(batch)
                rows.append([float(losses[k].detach()) for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc")])
                print("  step", step, rows[-1])
                opt.zero_grad()
                sum(losses.values()).backward()
                opt.step()
    finally:
        torch.randperm = real
    save("train_base_steps", losses=np.array(rows, np.float64), gt_boxes0=t["gt_boxes0"], gt_boxes1=t["gt_boxes1"],
         gt_classes0=t["gt_classes0"], gt_classes1=t["gt_classes1"])


def gen_train_ft_all():
    """faster_rcnn_R_50_FPN_ft_all_30shot_aug_ftmore_dropout.yaml (the second fine-tune of BASELINE config 3: 80
    classes, BACKBONE.FREEZE only -- RPN head, 2-FC box head and the CosineSimOutputLayers predictor train), with
    ROI_BOX_HEAD.DROPOUT overridden to 0 (the yaml's 0.5 draws a mask from torch's generator, which a GPU run cannot
    reproduce).  One step on the gen_train batch; randperm = identity."""
    from detectron2.structures import Boxes, Instances
    from detectron2.utils.events import EventStorage

    cfg, model = build_ref_model("COCO-detection/faster_rcnn_R_50_FPN_ft_all_30shot_aug_ftmore_dropout.yaml",
                                 ["MODEL.ROI_BOX_HEAD.DROPOUT", 0.0])
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r50_bn_calibration.npz")).items()}
    sd = syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib)
    model.load_state_dict(sd, strict=True)
    model.train()
    t = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "train_novel_ft.npz")).items()}
    batch, d = [], {}
    for i, (h, w, seed) in enumerate([(240, 320, 3), (200, 352, 4)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(t["gt_boxes%d" % i])
        inst.gt_classes = t["gt_classes%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "height": h, "width": w})
        d["gt_boxes%d" % i], d["gt_classes%d" % i] = inst.gt_boxes.tensor, inst.gt_classes
    real = torch.randperm
    torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
    try:
        with EventStorage(0) as storage:
            losses = model(batch)
            sum(losses.values()).backward()
            scalars = {k: float(v[0]) if isinstance(v, tuple) else float(v) for k, v in storage.latest().items()}
    finally:
        torch.randperm = real
    ntrain, frozen = 0, []
    for n_, p_ in model.named_parameters():
        if p_.requires_grad:
            ntrain += 1
            gflat = p_.grad.flatten()
            stride = max(1, gflat.numel() // 2048) | 1   # odd: never a whole number of rows of a 2^k-wide matrix
            d["grad_sample." + n_] = gflat[::stride][:2048].clone()
            d["grad_stats." + n_] = torch.tensor([float(gflat.double().sum()), float(gflat.double().norm()), float(stride)], dtype=torch.float64)
        else:
            frozen.append(n_)
    print("  losses", {k: float(v.detach()) for k, v in losses.items()}, "trainable", ntrain, "frozen", len(frozen), scalars)
    d["trainable_names"] = np.array([n_ for n_, p_ in model.named_parameters() if p_.requires_grad])
    # the in-place renormalised classification weight after the forward (fast_rcnn.py:830-837)
    d["cls_weight_after"] = model.roi_heads.box_predictor.cls_score.weight.detach()
    save("train_ft_all", **d, **{"loss." + k: v.detach() for k, v in losses.items()},
         **{"scalar." + k.replace("/", "."): np.float64(v) for k, v in scalars.items()})


def gen_box_corrector():
    """Box-corrector inference over given pseudo-labels: GeneralizedRCNNRegOnly + CascadeROIHeads._forward_box_qe
    (cascade_ubbr yaml with META_ARCHITECTURE switched, as tools/train_net_reg_qe.py does)."""
    from detectron2.structures import Boxes, Instances

    cfg, model = build_ref_model("COCO-detection/cascade_ubbr_R_50_FPN_base.yaml",
                                 ["MODEL.META_ARCHITECTURE", "GeneralizedRCNNRegOnly"])
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r50_bn_calibration.npz")).items()}
    sd = syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib)
    model.load_state_dict(sd, strict=True)
    save("cascade_state_dict_keys", keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))
    g = torch.Generator().manual_seed(21)
    batch, d = [], {}
    for i, (h, w, seed, oh, ow) in enumerate([(240, 320, 3, 480, 640), (200, 352, 4, 200, 352)]):
        n = 12 + 3 * i
        x1 = torch.rand(n, generator=g) * (w - 60)
        y1 = torch.rand(n, generator=g) * (h - 60)
        bw = 8 + torch.rand(n, generator=g).pow(2) * 250
        bh = 8 + torch.rand(n, generator=g).pow(2) * 180
        boxes = torch.stack([x1, y1, (x1 + bw).clamp(max=w), (y1 + bh).clamp(max=h)], 1)
        classes = torch.randint(0, 60, (n,), generator=g)
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(boxes.clone())
        inst.gt_classes = classes
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "height": oh, "width": ow})
        d["in_boxes%d" % i] = boxes
        d["in_classes%d" % i] = classes
    with torch.no_grad():
        out = model(batch)
    for i, o in enumerate(out):
        d["out_boxes%d" % i] = o["instances"].pred_boxes.tensor
        d["out_classes%d" % i] = o["instances"].pred_classes
        print("  image", i, "in", len(d["in_boxes%d" % i]), "out", len(d["out_boxes%d" % i]),
              "mean |shift|", float((d["out_boxes%d" % i][: len(d["in_boxes%d" % i])] - 0).abs().mean()))
    save("box_corrector", **d)


def gen_box_corrector_train():
    """BASELINE config 5, fine-tune flavour (cascade_ubbr_R_50_FPN_ft_all_30shot_aug_ftmore.yaml: BACKBONE.FREEZE):
    one training step of GeneralizedRCNN + RBG + CascadeROIHeads/BoxOnlyLayersCascade on a fixed 2-image batch.
    Stored: the proposals RBG produced (its jitter draws from torch's CPU generator, which a GPU run cannot
    reproduce -- the product test feeds these boxes through a patched RBG), the three stage losses, and for every
    trainable tensor its gradient's sum, L2 norm and a strided sample of <= 4096 entries (fc1 gradients are 51 MB).
    torch.randperm is patched to the identity as in gen_train."""
    from detectron2.structures import Boxes, Instances
    from detectron2.utils.events import EventStorage

    cfg, model = build_ref_model("COCO-detection/cascade_ubbr_R_50_FPN_ft_all_30shot_aug_ftmore.yaml")
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r50_bn_calibration.npz")).items()}
    sd = syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib)
    model.load_state_dict(sd, strict=True)
    model.train()
    g = torch.Generator().manual_seed(31)
    batch, d = [], {}
    for i, (h, w, seed) in enumerate([(240, 320, 3), (200, 352, 4)]):
        n = 4 + i
        x1 = torch.rand(n, generator=g) * (w - 100)
        y1 = torch.rand(n, generator=g) * (h - 100)
        bw = 40 + torch.rand(n, generator=g) * 140
        bh = 40 + torch.rand(n, generator=g) * 110
        boxes = torch.stack([x1, y1, (x1 + bw).clamp(max=w), (y1 + bh).clamp(max=h)], 1)
        classes = torch.randint(0, 80, (n,), generator=g)
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(boxes)
        inst.gt_classes = classes
        # loaded proposals: noisy copies of the GT (some pass RBG's IoU > t filter, some do not) + far-away boxes
        noisy = boxes.repeat(3, 1) + torch.randn(3 * n, 4, generator=g) * 18
        far = torch.stack([torch.rand(4, generator=g) * 40, torch.rand(4, generator=g) * 40,
                           60 + torch.rand(4, generator=g) * 30, 60 + torch.rand(4, generator=g) * 30], 1)
        pb = torch.cat([noisy, far])
        pb[:, 2:] = torch.max(pb[:, 2:], pb[:, :2] + 4)
        props = Instances((h, w))
        props.proposal_boxes = Boxes(pb)
        props.objectness_logits = torch.randn(len(pb), generator=g)
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "proposals": props, "height": h, "width": w})
        d["gt_boxes%d" % i], d["gt_classes%d" % i] = boxes, classes
        d["loaded_boxes%d" % i], d["loaded_logits%d" % i] = pb, props.objectness_logits
    rbg_out = []
    orig = model.proposal_generator.forward

    def recording(proposals, targets):
        out, extra = orig(proposals, targets)
        rbg_out.extend(out)
        return out, extra

    model.proposal_generator.forward = recording
    real = torch.randperm
    torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
    torch.manual_seed(5)
    try:
        with EventStorage(0) as storage:
            losses = model(batch)
            sum(losses.values()).backward()
            scalars = {k: float(v[0]) if isinstance(v, tuple) else float(v) for k, v in storage.latest().items()}
    finally:
        torch.randperm = real
    for i, p in enumerate(rbg_out):
        d["rbg_boxes%d" % i] = p.proposal_boxes.tensor
        d["rbg_logits%d" % i] = p.objectness_logits
    ntrain = 0
    for n_, p_ in model.named_parameters():
        if p_.requires_grad:
            ntrain += 1
            gflat = p_.grad.flatten()
            stride = max(1, gflat.numel() // 4096) | 1   # odd: never a whole number of rows of a 2^k-wide matrix
            d["grad_sample." + n_] = gflat[::stride][:4096].clone()
            d["grad_stats." + n_] = torch.tensor([float(gflat.double().sum()), float(gflat.double().norm()), float(stride)], dtype=torch.float64)
    print("  losses", {k: float(v) for k, v in losses.items()}, "trainable tensors", ntrain,
          "rbg proposals", [len(p) for p in rbg_out], scalars)
    save("box_corrector_train", **d, **{"loss." + k: v.detach() for k, v in losses.items()},
         **{"scalar." + k.replace("/", "."): np.float64(v) for k, v in scalars.items()})


def gen_box_corrector_eval():
    """Evaluation of the same box-corrector config (GeneralizedRCNN.inference, rcnn.py:201-230): IoU with the matched GT
    before / after the cascade for the subsampled foreground proposals.  Same batch as gen_box_corrector_train (GT and
    loaded proposals are read back from its fixture), randperm = identity."""
    from detectron2.structures import Boxes, Instances

    cfg, model = build_ref_model("COCO-detection/cascade_ubbr_R_50_FPN_ft_all_30shot_aug_ftmore.yaml")
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r50_bn_calibration.npz")).items()}
    sd = syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib)
    model.load_state_dict(sd, strict=True)
    model.eval()
    t = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "box_corrector_train.npz")).items()}
    batch = []
    for i, (h, w, seed) in enumerate([(240, 320, 3), (200, 352, 4)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(t["gt_boxes%d" % i])
        inst.gt_classes = t["gt_classes%d" % i]
        props = Instances((h, w))
        props.proposal_boxes = Boxes(t["loaded_boxes%d" % i])
        props.objectness_logits = t["loaded_logits%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "proposals": props, "height": h, "width": w})
    real = torch.randperm
    torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
    try:
        with torch.no_grad():
            out = model(batch)
    finally:
        torch.randperm = real
    print("  rows", len(out["gt_classes"]), "mean IoU in %.4f out %.4f" % (float(out["input_ious"].mean()), float(out["output_ious"].mean())))
    save("box_corrector_eval", **{k: v for k, v in out.items()})


def gen_crops():
    from detectron2.structures import Boxes, Instances
    from lvc.data.utils import get_crops_qe

    g = torch.Generator().manual_seed(31)
    img = syn.synthetic_image(7, 300, 420)[None]
    boxes = []
    for _ in range(12):
        x1 = int(torch.randint(0, 380, (1,), generator=g)); y1 = int(torch.randint(0, 260, (1,), generator=g))
        w = int(torch.randint(3, 200, (1,), generator=g)); h = int(torch.randint(3, 150, (1,), generator=g))
        boxes.append([x1, y1, min(x1 + w, 419), min(y1 + h, 299)])
    boxes += [[0, 0, 419, 299], [5, 5, 5, 5], [400, 280, 419, 299]]
    insts = [Instances((300, 420), gt_boxes=Boxes(torch.tensor([b], dtype=torch.float32))) for b in boxes]
    d = {"boxes": torch.tensor(boxes)}
    for op in ("pad", "context"):
        crops = get_crops_qe(img, insts, op)
        d["crops_" + op] = crops[:, :, ::7, ::7].contiguous()     # 32x32 sample of every crop
        d["sum_" + op] = crops.double().sum(dim=(1, 2, 3))
    save("crops", **d)


def gen_box_corrector_train_base():
    """BASELINE config 5 proper (cascade_ubbr_R_50_FPN_base.yaml: BACKBONE.FREEZE_AT 2, 60 classes): one training step
    with the trunk TRAINING from res3 up -- ROIAlign backward into p2..p5, FPN, res5..res3.  Same 2-image batch as
    gen_box_corrector_train (read back from its fixture, classes folded into the 60 base classes).  Stored: RBG's
    proposals, the three stage losses, and per trainable tensor the gradient's sum, L2 norm and a strided sample."""
    from detectron2.structures import Boxes, Instances
    from detectron2.utils.events import EventStorage

    cfg, model = build_ref_model("COCO-detection/cascade_ubbr_R_50_FPN_base.yaml")
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r50_bn_calibration.npz")).items()}
    sd = syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib)
    model.load_state_dict(sd, strict=True)
    model.train()
    t = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "box_corrector_train.npz")).items()}
    batch, d = [], {}
    for i, (h, w, seed) in enumerate([(240, 320, 3), (200, 352, 4)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(t["gt_boxes%d" % i])
        inst.gt_classes = t["gt_classes%d" % i] % 60
        props = Instances((h, w))
        props.proposal_boxes = Boxes(t["loaded_boxes%d" % i])
        props.objectness_logits = t["loaded_logits%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "proposals": props, "height": h, "width": w})
        d["gt_boxes%d" % i], d["gt_classes%d" % i] = inst.gt_boxes.tensor, inst.gt_classes
        d["loaded_boxes%d" % i], d["loaded_logits%d" % i] = props.proposal_boxes.tensor, props.objectness_logits
    rbg_out = []
    orig = model.proposal_generator.forward

    def recording(proposals, targets):
        out, extra = orig(proposals, targets)
        rbg_out.extend(out)
        return out, extra

    model.proposal_generator.forward = recording
    real = torch.randperm
    torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
    torch.manual_seed(5)
    try:
        with EventStorage(0) as storage:
            losses = model(batch)
            sum(losses.values()).backward()
            scalars = {k: float(v[0]) if isinstance(v, tuple) else float(v) for k, v in storage.latest().items()}
    finally:
        torch.randperm = real
    for i, p in enumerate(rbg_out):
        d["rbg_boxes%d" % i] = p.proposal_boxes.tensor
        d["rbg_logits%d" % i] = p.objectness_logits
    ntrain, frozen = 0, []
    for n_, p_ in model.named_parameters():
        if p_.requires_grad:
            ntrain += 1
            gflat = p_.grad.flatten()
            stride = max(1, gflat.numel() // 2048) | 1   # odd: never a whole number of rows of a 2^k-wide matrix
            d["grad_sample." + n_] = gflat[::stride][:2048].clone()
            d["grad_stats." + n_] = torch.tensor([float(gflat.double().sum()), float(gflat.double().norm()), float(stride)], dtype=torch.float64)
        else:
            frozen.append(n_)
    print("  losses", {k: float(v) for k, v in losses.items()}, "trainable tensors", ntrain, "frozen", len(frozen),
          "rbg proposals", [len(p) for p in rbg_out], scalars)
    d["frozen_names"] = np.array(frozen)
    save("box_corrector_train_base", **d, **{"loss." + k: v.detach() for k, v in losses.items()},
         **{"scalar." + k.replace("/", "."): np.float64(v) for k, v in scalars.items()})


def gen_r101():
    """R101-FPN (the depth BASELINE config 5 names): FrozenBN calibration of the seed-0 conditioned weights on a small
    image, an end-to-end detection golden on the 2-image small batch (faster_rcnn base yaml, DEPTH 101, 80 classes), and
    one box-corrector training step (cascade_ubbr base yaml with DEPTH 101: res3..res5 incl. res4's 23 blocks train)."""
    from detectron2.layers import FrozenBatchNorm2d
    from detectron2.structures import Boxes, Instances
    from detectron2.utils.events import EventStorage

    cfg, model = build_ref_model("COCO-detection/faster_rcnn_R_50_FPN_base.yaml",
                                 ["MODEL.ROI_HEADS.NUM_CLASSES", 80, "MODEL.RESNETS.DEPTH", 101])
    model.load_state_dict(syn.conditioned_state_dict(model.state_dict(), seed=0), strict=True)
    a = syn.synthetic_image(3, 240, 320)
    b = syn.synthetic_image(4, 200, 352)
    calib = syn.calibrate_frozen_bn_(model, lambda: model([{"image": a, "height": 240, "width": 320}]), FrozenBatchNorm2d)
    save("r101_bn_calibration", **calib)
    sd = model.state_dict()
    save("r101_fpn_state_dict_keys", keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))
    inputs = [{"image": a, "height": 480, "width": 640}, {"image": b, "height": 200, "width": 352}]
    with torch.no_grad():
        images = model.preprocess_image(inputs)
        feats = model.backbone(images.tensor)
        props, _ = model.proposal_generator(images, feats, None)
        out = model(inputs)
    d = {}
    for k, v in feats.items():
        d["feat_" + k] = v[:, ::8].contiguous()
    for i in range(2):
        d["prop_boxes%d" % i] = props[i].proposal_boxes.tensor
        d["prop_logits%d" % i] = props[i].objectness_logits
        inst = out[i]["instances"]
        d["det_boxes%d" % i] = inst.pred_boxes.tensor
        d["det_scores%d" % i] = inst.scores
        d["det_classes%d" % i] = inst.pred_classes
        print("  small image", i, "proposals", len(props[i]), "detections", len(inst))
    save("e2e_r101_fpn_small", **d)

    # ---- box-corrector training step, R101 (BASELINE config 5)
    cfg, model = build_ref_model("COCO-detection/cascade_ubbr_R_50_FPN_base.yaml", ["MODEL.RESNETS.DEPTH", 101])
    model.load_state_dict(syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib), strict=True)
    model.train()
    t = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "box_corrector_train_base.npz")).items()
         if not k.startswith("frozen")}
    batch, d = [], {}
    for i, (h, w, seed) in enumerate([(240, 320, 3), (200, 352, 4)]):
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(t["gt_boxes%d" % i])
        inst.gt_classes = t["gt_classes%d" % i]
        props = Instances((h, w))
        props.proposal_boxes = Boxes(t["loaded_boxes%d" % i])
        props.objectness_logits = t["loaded_logits%d" % i]
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "proposals": props, "height": h, "width": w})
        for k in ("gt_boxes", "gt_classes", "loaded_boxes", "loaded_logits"):
            d["%s%d" % (k, i)] = t["%s%d" % (k, i)]
    rbg_out = []
    orig = model.proposal_generator.forward

    def recording(proposals, targets):
        out, extra = orig(proposals, targets)
        rbg_out.extend(out)
        return out, extra

    model.proposal_generator.forward = recording
    real = torch.randperm
    torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
    torch.manual_seed(5)
    try:
        with EventStorage(0):
            losses = model(batch)
            sum(losses.values()).backward()
    finally:
        torch.randperm = real
    for i, p in enumerate(rbg_out):
        d["rbg_boxes%d" % i] = p.proposal_boxes.tensor
        d["rbg_logits%d" % i] = p.objectness_logits
    ntrain = 0
    for n_, p_ in model.named_parameters():
        if p_.requires_grad:
            ntrain += 1
            gflat = p_.grad.flatten()
            stride = max(1, gflat.numel() // 1024) | 1   # odd: never a whole number of rows of a 2^k-wide matrix
            d["grad_sample." + n_] = gflat[::stride][:1024].clone()
            d["grad_stats." + n_] = torch.tensor([float(gflat.double().sum()), float(gflat.double().norm()), float(stride)], dtype=torch.float64)
    print("  R101 corrector losses", {k: float(v.detach()) for k, v in losses.items()}, "trainable tensors", ntrain)
    save("box_corrector_train_r101", **d, **{"loss." + k: v.detach() for k, v in losses.items()})


def gen_box_corrector_train_r101_full():
    """BASELINE config 5 at the size and batch it is quoted on (box-corrector training, R101-FPN, two 3 x 800 x 1333 images per
    GPU): one reference CPU step of cascade_ubbr base yaml with DEPTH 101 -- 133 trainable tensors, the full pyramid under
    ROIAlign's backward.  Stored as in gen_r101: GT, loaded proposals, RBG's output, the three stage losses, per tensor the
    gradient's sum / norm / strided sample (<= 1024 entries)."""
    from detectron2.structures import Boxes, Instances
    from detectron2.utils.events import EventStorage

    cfg, model = build_ref_model("COCO-detection/cascade_ubbr_R_50_FPN_base.yaml", ["MODEL.RESNETS.DEPTH", 101])
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r101_bn_calibration.npz")).items()}
    model.load_state_dict(syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib), strict=True)
    model.train()
    g = torch.Generator().manual_seed(47)
    batch, d = [], {}
    for i, (h, w, seed) in enumerate([(800, 1333, 5), (800, 1333, 6)]):
        n = 6 + i
        # boxes of 40 .. 600 px: every pooler level (canonical size 224) is used
        side = 40.0 * (15.0 ** torch.rand(n, generator=g))
        bw = side * (0.7 + 0.6 * torch.rand(n, generator=g))
        bh = side * (0.7 + 0.6 * torch.rand(n, generator=g))
        x1 = torch.rand(n, generator=g) * (w - bw).clamp(min=1)
        y1 = torch.rand(n, generator=g) * (h - bh).clamp(min=1)
        boxes = torch.stack([x1, y1, (x1 + bw).clamp(max=w), (y1 + bh).clamp(max=h)], 1)
        classes = torch.randint(0, 60, (n,), generator=g)
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(boxes)
        inst.gt_classes = classes
        # loaded proposals: noisy copies of the GT (noise in proportion to the box: some pass RBG's IoU filter, some do not) + far boxes
        noisy = boxes.repeat(3, 1) + torch.randn(3 * n, 4, generator=g) * 0.12 * side.repeat(3)[:, None]
        far = torch.stack([torch.rand(4, generator=g) * 100, torch.rand(4, generator=g) * 100,
                           150 + torch.rand(4, generator=g) * 80, 150 + torch.rand(4, generator=g) * 80], 1)
        pb = torch.cat([noisy, far])
        pb[:, 2:] = torch.max(pb[:, 2:], pb[:, :2] + 4)
        props = Instances((h, w))
        props.proposal_boxes = Boxes(pb)
        props.objectness_logits = torch.randn(len(pb), generator=g)
        batch.append({"image": syn.synthetic_image(seed, h, w), "instances": inst, "proposals": props, "height": h, "width": w})
        d["gt_boxes%d" % i], d["gt_classes%d" % i] = boxes, classes
        d["loaded_boxes%d" % i], d["loaded_logits%d" % i] = pb, props.objectness_logits
    rbg_out = []
    orig = model.proposal_generator.forward

    def recording(proposals, targets):
        out, extra = orig(proposals, targets)
        rbg_out.extend(out)
        return out, extra

    model.proposal_generator.forward = recording
    real = torch.randperm
    torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
    torch.manual_seed(5)
    try:
        with EventStorage(0):
            losses = model(batch)
            sum(losses.values()).backward()
    finally:
        torch.randperm = real
    for i, p in enumerate(rbg_out):
        d["rbg_boxes%d" % i] = p.proposal_boxes.tensor
        d["rbg_logits%d" % i] = p.objectness_logits
    ntrain = 0
    for n_, p_ in model.named_parameters():
        if p_.requires_grad:
            ntrain += 1
            gflat = p_.grad.flatten()
            stride = max(1, gflat.numel() // 1024) | 1
            d["grad_sample." + n_] = gflat[::stride][:1024].clone()
            d["grad_stats." + n_] = torch.tensor([float(gflat.double().sum()), float(gflat.double().norm()), float(stride)], dtype=torch.float64)
    print("  R101 corrector, 2 x 800 x 1333: losses", {k: float(v.detach()) for k, v in losses.items()}, "trainable tensors", ntrain,
          "rbg proposals", [len(p) for p in rbg_out])
    save("box_corrector_train_r101_800x1333", **d, **{"loss." + k: v.detach() for k, v in losses.items()})


def gen_r101_full():
    """R101-FPN at the headline's image size (VERDICT r4 missing #5): detections, proposals and sampled pyramid features of
    two 3x800x1333 images through the reference's CPU path, same weights as gen_r101 (seed-0 conditioned, its FrozenBN
    calibration): pins bench.py's r101 leg and tests/test_gpu_e2e.py::test_r101_e2e_800x1333_matches_reference_cpu."""
    cfg, model = build_ref_model("COCO-detection/faster_rcnn_R_50_FPN_base.yaml",
                                 ["MODEL.ROI_HEADS.NUM_CLASSES", 80, "MODEL.RESNETS.DEPTH", 101])
    calib = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "r101_bn_calibration.npz")).items()}
    model.load_state_dict(syn.conditioned_state_dict(model.state_dict(), seed=0, bn_calibration=calib), strict=True)
    inputs = [{"image": syn.synthetic_image(1), "height": 800, "width": 1333},
              {"image": syn.synthetic_image(2), "height": 800, "width": 1333}]
    with torch.no_grad():
        images = model.preprocess_image(inputs)
        feats = model.backbone(images.tensor)
        props, _ = model.proposal_generator(images, feats, None)
        out = model(inputs)
    d = {}
    for k, v in feats.items():
        d["feat_" + k] = v[:, ::16, ::8, ::8].contiguous()
        d["featstat_" + k] = torch.stack([v.mean(), v.std(), v.abs().max()])
    for i in range(2):
        d["prop_boxes%d" % i] = props[i].proposal_boxes.tensor
        d["prop_logits%d" % i] = props[i].objectness_logits
        inst = out[i]["instances"]
        d["det_boxes%d" % i] = inst.pred_boxes.tensor
        d["det_scores%d" % i] = inst.scores
        d["det_classes%d" % i] = inst.pred_classes
        print("  image", i, "proposals", len(props[i]), "detections", len(inst), "score range", float(inst.scores.max()), float(inst.scores.min()))
    save("e2e_r101_fpn_800x1333", **d)


def gen_resize():
    """Test-time input transform: the reference's ResizeShortestEdge.get_transform sizes and
    ResizeTransform.apply_image (Pillow bilinear on uint8 HWC) on small random images covering up-scaling, anti-aliased
    down-scaling, one unchanged dimension, and the max_size clamp."""
    from detectron2.data.transforms import ResizeShortestEdge

    rng = np.random.default_rng(7)
    d = {}
    cases = [(60, 80, 100, 1333), (90, 120, 64, 1333), (50, 200, 100, 333), (128, 96, 96, 1333), (37, 53, 80, 117),
             (240, 160, 50, 1333)]
    for i, (h, w, short, mx) in enumerate(cases):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        aug = ResizeShortestEdge([short, short], mx, "choice")
        tfm = aug.get_transform(img)
        out = tfm.apply_image(img)
        d["in%d" % i], d["out%d" % i] = img, out
        d["cfg%d" % i] = np.array([short, mx, tfm.new_h, tfm.new_w], np.int64)
        boxes = np.array([[1.5, 2.0, w - 3.0, h - 1.0], [0.0, 0.0, 10.0, 12.5]], np.float32)
        d["box_in%d" % i] = boxes
        d["box_out%d" % i] = tfm.apply_box(boxes.copy()).astype(np.float32)
        print("  case", i, (h, w), "->", out.shape[:2])
    d["n"] = np.int64(len(cases))
    save("resize", **d)


def gen_solver():
    """Optimizer parameter groups and learning-rate schedules (SURVEY 8 rows 19/20's host side): the reference's
    build_optimizer on a toy model holding every parameter kind the rules distinguish (conv weight, conv bias, norm
    weight/bias, a frozen parameter, a shared parameter), three SGD steps with value and norm clipping, and the
    per-iteration lr of build_lr_scheduler for the shipped schedules."""
    from detectron2.config import get_cfg
    from detectron2.solver import build_lr_scheduler, build_optimizer

    def toy():
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                                torch.nn.Conv2d(4, 4, 1, bias=False), torch.nn.GroupNorm(2, 4), torch.nn.Flatten(),
                                torch.nn.Linear(4 * 6 * 6, 5), torch.nn.Linear(5, 5),
                                torch.nn.Linear(5, 5, bias=False))
        m[8].weight = m[7].weight            # shared: listed once
        m[7].bias.requires_grad_(False)      # frozen: not listed
        return m

    d = {}
    solver_cases = [
        dict(BASE_LR=0.02, WEIGHT_DECAY=1e-4, WEIGHT_DECAY_NORM=0.0, BIAS_LR_FACTOR=2.0, WEIGHT_DECAY_BIAS=0.0,
             MOMENTUM=0.9, NESTEROV=False, clip=None),
        dict(BASE_LR=0.01, WEIGHT_DECAY=5e-4, WEIGHT_DECAY_NORM=1e-5, BIAS_LR_FACTOR=1.0, WEIGHT_DECAY_BIAS=5e-4,
             MOMENTUM=0.8, NESTEROV=True, clip=("value", 0.01, 2.0)),
        dict(BASE_LR=0.05, WEIGHT_DECAY=1e-4, WEIGHT_DECAY_NORM=0.0, BIAS_LR_FACTOR=1.0, WEIGHT_DECAY_BIAS=1e-4,
             MOMENTUM=0.9, NESTEROV=False, clip=("norm", 0.05, 2.0)),
    ]
    x = torch.randn(2, 3, 6, 6, generator=torch.Generator().manual_seed(1))
    for i, c in enumerate(solver_cases):
        cfg = get_cfg()
        for k, v in c.items():
            if k != "clip":
                setattr(cfg.SOLVER, k, v)
        if c["clip"]:
            cfg.SOLVER.CLIP_GRADIENTS.ENABLED = True
            cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE, cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE, cfg.SOLVER.CLIP_GRADIENTS.NORM_TYPE = c["clip"]
        m = toy()
        opt = build_optimizer(cfg, m)
        d["groups%d" % i] = np.array([[g["lr"], g["weight_decay"], g["momentum"], float(g["nesterov"]),
                                       g["params"][0].numel()] for g in opt.param_groups], np.float64)
        for _ in range(3):
            opt.zero_grad()
            (m(x) ** 2).sum().backward()
            opt.step()
        d["after%d" % i] = np.concatenate([p.detach().reshape(-1).numpy() for p in m.parameters()]).astype(np.float32)
        d["case%d" % i] = np.array([c[k] for k in ("BASE_LR", "WEIGHT_DECAY", "WEIGHT_DECAY_NORM", "BIAS_LR_FACTOR",
                                                   "WEIGHT_DECAY_BIAS", "MOMENTUM")] + [float(c["NESTEROV"])]
                                   + ([{"value": 1.0, "norm": 2.0}[c["clip"][0]], c["clip"][1], c["clip"][2]]
                                      if c["clip"] else [0.0, 0.0, 0.0]), np.float64)
    d["x"] = x.numpy()
    d["n_solver"] = np.int64(len(solver_cases))

    sched_cases = [
        ("WarmupMultiStepLR", 0.02, (60, 80), 0.1, 90, 0.001, 20, "linear"),
        ("WarmupMultiStepLR", 0.001, (30,), 0.5, 50, 0.1, 10, "constant"),
        ("WarmupMultiStepLR", 0.01, (5, 7), 0.1, 30, 0.001, 10, "linear"),      # a milestone inside the warmup
        ("WarmupMultiStepLR", 0.01, (20,), 0.1, 30, 0.001, 0, "linear"),        # no warmup
        ("WarmupCosineLR", 0.04, (30,), 0.1, 64, 0.01, 16, "linear"),
    ]
    for i, (name, lr, steps, gamma, max_iter, wf, wi, wm) in enumerate(sched_cases):
        cfg = get_cfg()
        cfg.SOLVER.LR_SCHEDULER_NAME, cfg.SOLVER.BASE_LR, cfg.SOLVER.STEPS, cfg.SOLVER.GAMMA = name, lr, steps, gamma
        cfg.SOLVER.MAX_ITER, cfg.SOLVER.WARMUP_FACTOR, cfg.SOLVER.WARMUP_ITERS, cfg.SOLVER.WARMUP_METHOD = max_iter, wf, wi, wm
        cfg.SOLVER.BIAS_LR_FACTOR = 2.0
        m = torch.nn.Linear(3, 2)
        opt = build_optimizer(cfg, m)
        sch = build_lr_scheduler(cfg, opt)
        lrs = []
        for _ in range(max_iter):
            lrs.append([g["lr"] for g in opt.param_groups])
            opt.step()
            sch.step()
        d["lrs%d" % i] = np.array(lrs, np.float64)
        d["sched%d" % i] = np.array([{"WarmupMultiStepLR": 0.0, "WarmupCosineLR": 1.0}[name], lr, gamma, max_iter, wf, wi,
                                     {"linear": 0.0, "constant": 1.0}[wm]], np.float64)
        d["steps%d" % i] = np.array(steps, np.int64)
    d["n_sched"] = np.int64(len(sched_cases))
    save("solver", **d)


def gen_wire():
    """SURVEY 8(f).2 wire formats: the reference's pseudo-label filter (tools/create_coco_dataset_from_dets_all.py:
    get_ret_anns + save_coco, through the restated pycocotools index of the shim) and the verified-dataset writer
    (tools/run_nearest_neighbours.py:230-249 save_coco) on a small synthetic detections file.  The fixture holds the input
    json documents and the exact bytes of every file the reference wrote."""
    import argparse
    import json
    import shutil
    import tempfile
    from collections import defaultdict

    sys.argv = [sys.argv[0]]
    import tools.create_coco_dataset_from_dets_all as T
    import tools.run_nearest_neighbours as R

    names = ["person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
             "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow"]
    cats = [{"id": i + 1 + (i // 11), "name": n, "supercategory": "x"} for i, n in enumerate(names)]   # ids with a gap, like COCO's
    g = torch.Generator().manual_seed(99)
    images = [{"id": 1000 + 7 * i, "file_name": "img_%03d.jpg" % i, "height": 400 + 16 * (i % 5), "width": 600 + 8 * (i % 3)}
              for i in range(14)]
    gt = {"info": {"description": "synthetic"}, "licenses": [], "categories": cats, "images": images, "annotations": []}
    dets = []
    for k in range(260):
        im = images[int(torch.randint(0, len(images), (1,), generator=g))]
        c = cats[int(torch.randint(0, len(cats), (1,), generator=g))]
        w = float(torch.rand(1, generator=g)) * 300 + 4
        h = float(torch.rand(1, generator=g)) * 250 + 4
        x = float(torch.rand(1, generator=g)) * (im["width"] - w)
        y = float(torch.rand(1, generator=g)) * (im["height"] - h)
        score = round(float(torch.rand(1, generator=g)) ** 0.5, 4)       # rounded: equal scores occur
        row = {"image_id": im["id"], "category_id": c["id"], "bbox": [x, y, w, h], "score": score}
        if k % 3 == 0:
            row["top2_scores"] = [score, score / 2]
            row["top2_inds"] = [c["id"], 1]
        dets.append(row)
    dets[17]["bbox"] = [0.0, 0.0, float(images[0]["width"]), float(images[0]["height"])]      # area_ratio 1.0 when it lands on image 0
    dets[17]["image_id"] = images[0]["id"]
    train_imgs = defaultdict(list)
    for j, c in enumerate(cats):
        train_imgs[c["id"]] = [images[(j + t) % len(images)]["id"] for t in range(3)]
    cases = [dict(top=False, full=False, K_min=0.8, K_max=1.0, ar=0.0), dict(top=False, full=True, K_min=0.7, K_max=0.95, ar=0.0),
             dict(top=True, full=False, K_min=6.0, K_max=1.0, ar=0.0), dict(top=True, full=True, K_min=4.0, K_max=0.0, ar=0.0),
             dict(top=False, full=False, K_min=0.5, K_max=1.0, ar=0.05)]
    out_dir = os.path.join(GOLD, "wire")
    os.makedirs(out_dir, exist_ok=True)
    json.dump(gt, open(os.path.join(out_dir, "gt.json"), "w"))
    json.dump(dets, open(os.path.join(out_dir, "dets.json"), "w"))
    json.dump({str(k): v for k, v in train_imgs.items()}, open(os.path.join(out_dir, "train_imgs.json"), "w"))
    tmp = tempfile.mkdtemp()
    manifest = []
    try:
        gt_path = os.path.join(tmp, "gt.json")
        json.dump(gt, open(gt_path, "w"))
        for ci, case in enumerate(cases):
            dt_path = os.path.join(tmp, "dets.json")
            json.dump(dets, open(dt_path, "w"))
            args = argparse.Namespace(dt_path=dt_path, full_dataset=False, all_cats=False, **case)
            coco_gt = T.COCO_PK(gt_path)
            ids = T.get_ids_names(coco_gt)
            unseen_coco_ids = ids[3]
            coco_dt = coco_gt.loadRes(dt_path, False)
            anns = T.get_ret_anns(coco_dt, train_imgs, args, unseen_coco_ids)
            img_ids = list(set(a["image_id"] for a in anns))
            name = T.save_coco(args, coco_gt, coco_dt, anns, coco_gt.loadImgs(img_ids))
            dst = "case%d__%s" % (ci, os.path.basename(name))
            shutil.copy(name, os.path.join(out_dir, dst))
            manifest.append({"case": case, "file": dst, "n_annotations": len(anns), "unseen_coco_ids": unseen_coco_ids})
            print("  case", ci, case, "->", os.path.basename(name), len(anns), "annotations")
            if ci == 0:   # the kNN step keeps a subset of those annotations: run_nearest_neighbours.save_coco
                keep_ids = [a["id"] for a in anns][::2][::-1]
                cfg = argparse.Namespace(QUERY_EXPAND=argparse.Namespace(NN_MODEL="dino_vits8/x", KNN=10, COSINE_SIM=True))
                vname = R.save_coco(cfg, keep_ids, name)
                shutil.copy(vname, os.path.join(out_dir, "verified__" + os.path.basename(vname)))
                manifest.append({"verified": "verified__" + os.path.basename(vname), "from": dst, "keep_ids": keep_ids})
    finally:
        shutil.rmtree(tmp)
    json.dump(manifest, open(os.path.join(out_dir, "manifest.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["roi_align", "nms", "box_ops", "rpn_and_det_ops", "blocks", "e2e", "knn", "train", "train_full", "train_base", "train_base_full", "train_base_steps", "train_ft_all", "box_corrector", "box_corrector_train", "box_corrector_eval", "box_corrector_train_base", "crops", "resize", "solver", "r101", "r101_full", "box_corrector_train_r101_full", "fpn_avg", "wire"]
    for w in which:
        print("== ", w)
        globals()["gen_" + w]()
