"""CPU oracle: a functional restatement of the reference's Faster-R-CNN-FPN inference path, driven
directly by a reference-layout `state_dict` (no nn.Modules, no registries).  TEST INFRASTRUCTURE
ONLY (see oracle/__init__.py).  Dense math uses torch CPU ops (ATen is what the reference itself
runs under MODEL.DEVICE=cpu); ROIAlign and NMS go through oracle.c.

Each function cites the reference code it follows (paths relative to /root/reference).  Pinned by
tests/test_oracle_golden.py against golden vectors generated from the imported reference
(oracle/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F

from . import ops as oops

SCALE_CLAMP = math.log(1000.0 / 16)  # detectron2/modeling/box_regression.py:11-12


# ------------------------------------------------------------------ preprocessing
def preprocess(images, pixel_mean, pixel_std, size_divisibility, dtype=torch.float32):
    """lvc/modeling/meta_arch/rcnn.py:324-333 + detectron2/structures/image_list.py:57-119.
    images: list of CHW float tensors.  Returns (padded [N,3,Hp,Wp], [(h,w),...]).  dtype=float64 evaluates the
    same formula in double (the "exact" run the fp32 paths are measured against)."""
    mean = torch.tensor(pixel_mean, dtype=dtype).view(-1, 1, 1)
    std = torch.tensor(pixel_std, dtype=dtype).view(-1, 1, 1)
    norm = [(im.to(dtype) - mean) / std for im in images]
    sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in norm]
    mh = max(s[0] for s in sizes)
    mw = max(s[1] for s in sizes)
    if size_divisibility > 1:
        d = size_divisibility
        mh = (mh + d - 1) // d * d
        mw = (mw + d - 1) // d * d
    out = torch.zeros(len(norm), norm[0].shape[0], mh, mw, dtype=dtype)
    for o, im in zip(out, norm):
        o[..., : im.shape[-2], : im.shape[-1]].copy_(im)
    return out, sizes


# ------------------------------------------------------------------ backbone
def _frozen_bn(sd, prefix, x, eps=1e-5):
    """detectron2/layers/batch_norm.py:45-65 (eval branch == F.batch_norm(training=False))."""
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training=False, eps=eps)


def _conv(sd, prefix, x, stride=1, padding=0, norm=True, relu=False):
    y = F.conv2d(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), stride=stride, padding=padding)
    if norm:
        y = _frozen_bn(sd, prefix + ".norm", y)
    return F.relu_(y) if relu else y


def _bottleneck(sd, prefix, x, stride):
    """detectron2/modeling/backbone/resnet.py:195-211 with STRIDE_IN_1X1=True (defaults.py:479)."""
    out = _conv(sd, prefix + ".conv1", x, stride=stride, relu=True)
    out = _conv(sd, prefix + ".conv2", out, padding=1, relu=True)
    out = _conv(sd, prefix + ".conv3", out)
    if (prefix + ".shortcut.weight") in sd:
        sc = _conv(sd, prefix + ".shortcut", x, stride=stride)
    else:
        sc = x
    out += sc
    return F.relu_(out)


def resnet(sd, x, depth=50, prefix="backbone.bottom_up"):
    """detectron2/modeling/backbone/resnet.py:588-592 (BasicStem), :708-731 (ResNet.forward),
    :845-941 (stage layout).  Returns {res2..res5}."""
    blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}[depth]
    x = _conv(sd, prefix + ".stem.conv1", x, stride=2, padding=3, relu=True)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    out = {}
    for si, nb in enumerate(blocks):
        name = "res%d" % (si + 2)
        for bi in range(nb):
            stride = 2 if (bi == 0 and si > 0) else 1
            x = _bottleneck(sd, "%s.%s.%d" % (prefix, name, bi), x, stride)
        out[name] = x
    return out


def fpn(sd, feats, prefix="backbone"):
    """detectron2/modeling/backbone/fpn.py:109-144 (sum fuse, no norm) + LastLevelMaxPool :165-177."""
    res = {}
    prev = _conv(sd, prefix + ".fpn_lateral5", feats["res5"], norm=False)
    res["p5"] = _conv(sd, prefix + ".fpn_output5", prev, padding=1, norm=False)
    for lvl in (4, 3, 2):
        top_down = F.interpolate(prev, scale_factor=2, mode="nearest")
        lat = _conv(sd, prefix + ".fpn_lateral%d" % lvl, feats["res%d" % lvl], norm=False)
        prev = lat + top_down
        res["p%d" % lvl] = _conv(sd, prefix + ".fpn_output%d" % lvl, prev, padding=1, norm=False)
    res["p6"] = F.max_pool2d(res["p5"], kernel_size=1, stride=2, padding=0)
    return {k: res[k] for k in ("p2", "p3", "p4", "p5", "p6")}


# ------------------------------------------------------------------ RPN
def rpn_head(sd, feats, prefix="proposal_generator.rpn_head"):
    """detectron2/modeling/proposal_generator/rpn.py:120-139."""
    logits, deltas = [], []
    for x in feats:
        t = F.relu(F.conv2d(x, sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"], padding=1))
        logits.append(F.conv2d(t, sd[prefix + ".objectness_logits.weight"], sd[prefix + ".objectness_logits.bias"]))
        deltas.append(F.conv2d(t, sd[prefix + ".anchor_deltas.weight"], sd[prefix + ".anchor_deltas.bias"]))
    return logits, deltas


def generate_cell_anchors(sizes, aspect_ratios):
    """detectron2/modeling/anchor_generator.py:179-208."""
    anchors = []
    for size in sizes:
        area = size ** 2.0
        for ar in aspect_ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            anchors.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(anchors).float()


def grid_anchors(cell_anchors, grid_sizes, strides, offset=0.0):
    """detectron2/modeling/anchor_generator.py:37-49, 157-178."""
    out = []
    for (gh, gw), stride, base in zip(grid_sizes, strides, cell_anchors):
        sx = torch.arange(offset * stride, gw * stride, step=stride, dtype=base.dtype)
        sy = torch.arange(offset * stride, gh * stride, step=stride, dtype=base.dtype)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        xx, yy = xx.reshape(-1), yy.reshape(-1)
        shifts = torch.stack((xx, yy, xx, yy), dim=1)
        out.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
    return out


def apply_deltas(deltas, boxes, weights, scale_clamp=SCALE_CLAMP):
    """detectron2/modeling/box_regression.py:73-110."""
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = torch.clamp(deltas[:, 2::4] / ww, max=scale_clamp)
    dh = torch.clamp(deltas[:, 3::4] / wh, max=scale_clamp)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    out = torch.zeros_like(deltas)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw
    out[:, 3::4] = pcy + 0.5 * ph
    return out


def clip_boxes_(boxes, size):
    """detectron2/structures/boxes.py:180-197 (Boxes.clip)."""
    h, w = size
    boxes[:, 0].clamp_(min=0, max=w)
    boxes[:, 1].clamp_(min=0, max=h)
    boxes[:, 2].clamp_(min=0, max=w)
    boxes[:, 3].clamp_(min=0, max=h)
    return boxes


def nonempty(boxes, threshold=0.0):
    """detectron2/structures/boxes.py:199-213."""
    return ((boxes[:, 2] - boxes[:, 0]) > threshold) & ((boxes[:, 3] - boxes[:, 1]) > threshold)


def flatten_rpn_outputs(logits, deltas):
    """rpn.py:427-438: (N,A,H,W)->(N,HWA); (N,4A,H,W)->(N,HWA,4)."""
    lg = [s.permute(0, 2, 3, 1).flatten(1) for s in logits]
    dl = [x.view(x.shape[0], -1, 4, x.shape[-2], x.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2) for x in deltas]
    return lg, dl


def find_top_rpn_proposals(anchors, logits, deltas, image_sizes, nms_thresh, pre_nms_topk, post_nms_topk,
                           min_box_size=0.0, weights=(1.0, 1.0, 1.0, 1.0)):
    """rpn.py:455-508 (_decode_proposals) + proposal_utils.py:13-118, inference branch.
    anchors: list[L] of [HWA,4]; logits: list[L] of [N,HWA]; deltas: list[L] of [N,HWA,4].
    Ties in the per-level sort are resolved lower-index-first (the reference's sort is unstable).
    Returns list of (boxes [n,4], logits [n])."""
    N = logits[0].shape[0]
    tk_scores, tk_boxes, lvl_ids = [], [], []
    for lid, (a, lg, dl) in enumerate(zip(anchors, logits, deltas)):
        props = apply_deltas(dl.reshape(-1, 4), a.unsqueeze(0).expand(N, -1, -1).reshape(-1, 4), weights)
        props = props.view(N, -1, 4)
        k = min(pre_nms_topk, lg.shape[1])
        srt, idx = lg.sort(dim=1, descending=True, stable=True)
        tk_scores.append(srt[:, :k])
        tk_boxes.append(torch.gather(props, 1, idx[:, :k, None].expand(-1, -1, 4)))
        lvl_ids.append(torch.full((k,), lid, dtype=torch.int64))
    tk_scores = torch.cat(tk_scores, 1)
    tk_boxes = torch.cat(tk_boxes, 1)
    lvl_ids = torch.cat(lvl_ids, 0)
    results = []
    for n, size in enumerate(image_sizes):
        boxes, scores, lvl = tk_boxes[n].clone(), tk_scores[n], lvl_ids
        valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores)
        boxes, scores, lvl = boxes[valid], scores[valid], lvl[valid]
        clip_boxes_(boxes, size)
        keep = nonempty(boxes, min_box_size)
        boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
        keep = oops.batched_nms(boxes, scores, lvl, nms_thresh)[:post_nms_topk]
        results.append((boxes[keep], scores[keep]))
    return results


# ------------------------------------------------------------------ ROI pooling
def assign_boxes_to_levels(boxes, min_level, max_level, canonical_box_size=224, canonical_level=4):
    """detectron2/modeling/poolers.py:23-59.  boxes [M,4]."""
    sizes = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def roi_pool(feats, scales, box_lists, output_size=7, sampling_ratio=0, aligned=True, roi_align=None):
    """detectron2/modeling/poolers.py:191-246 with ROIAlignV2 (aligned=True)."""
    roi_align = roi_align or oops.roi_align_forward
    dt = feats[0].dtype
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i), dtype=dt), b.to(dt)], 1) for i, b in enumerate(box_lists)], 0)
    min_level = int(round(-math.log2(scales[0])))
    max_level = int(round(-math.log2(scales[-1])))
    M, C = rois.shape[0], feats[0].shape[1]
    out = torch.zeros(M, C, output_size, output_size, dtype=dt)
    if len(feats) == 1:
        return roi_align(feats[0], rois, scales[0], output_size, output_size, sampling_ratio, aligned)
    lv = assign_boxes_to_levels(rois[:, 1:], min_level, max_level)
    for l, (f, s) in enumerate(zip(feats, scales)):
        inds = (lv == l).nonzero().view(-1)
        if len(inds):
            out[inds] = roi_align(f, rois[inds], s, output_size, output_size, sampling_ratio, aligned)
    return out


# ------------------------------------------------------------------ box head + predictor
def box_head(sd, x, num_fc=2, prefix="roi_heads.box_head"):
    """lvc/modeling/roi_heads/box_head.py:82-91 (no conv layers in the shipped configs; dropout = identity in eval)."""
    x = torch.flatten(x, start_dim=1)
    for i in range(1, num_fc + 1):
        x = F.relu(F.linear(x, sd["%s.fc%d.weight" % (prefix, i)], sd["%s.fc%d.bias" % (prefix, i)]))
    return x


def predictor(sd, x, cosine=False, cosine_scale=20.0, prefix="roi_heads.box_predictor"):
    """lvc/modeling/roi_heads/fast_rcnn.py:583-598 (FastRCNNOutputLayers) / :811-841 (CosineSimOutputLayers:
    x/(|x|+1e-5), W/(|W|+1e-5), scale 20, no class bias; the in-place weight renorm of :830-837 is applied to a
    copy here, i.e. this is the FIRST forward after load)."""
    if x.dim() > 2:
        x = torch.flatten(x, start_dim=1)
    if cosine:
        xn = x / (torch.norm(x, p=2, dim=1, keepdim=True) + 1e-5)
        w = sd[prefix + ".cls_score.weight"]
        wn = w / (torch.norm(w, p=2, dim=1, keepdim=True) + 1e-5)
        scores = cosine_scale * F.linear(xn, wn)
    else:
        scores = F.linear(x, sd[prefix + ".cls_score.weight"], sd[prefix + ".cls_score.bias"])
    deltas = F.linear(x, sd[prefix + ".bbox_pred.weight"], sd[prefix + ".bbox_pred.bias"])
    return scores, deltas


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk):
    """lvc/modeling/roi_heads/fast_rcnn.py:95-137.  boxes [R,4K|4], scores [R,K+1] (probabilities)."""
    scores = scores[:, :-1]
    nreg = boxes.shape[1] // 4
    boxes = clip_boxes_(boxes.reshape(-1, 4).clone(), image_shape).view(-1, nreg, 4)
    mask = scores > score_thresh
    inds = mask.nonzero(as_tuple=False)
    boxes = boxes[inds[:, 0], 0] if nreg == 1 else boxes[mask]
    scores = scores[mask]
    keep = oops.batched_nms(boxes, scores, inds[:, 1], nms_thresh)
    if topk >= 0:
        keep = keep[:topk]
    return boxes[keep], scores[keep], inds[keep, 1], inds[keep, 0]


def detector_postprocess(boxes, scores, classes, image_size, out_h, out_w, return_keep=False):
    """detectron2/modeling/postprocessing.py:10-79 (box fields only)."""
    sx, sy = out_w / image_size[1], out_h / image_size[0]
    boxes = boxes.clone()
    boxes[:, 0::2] *= sx
    boxes[:, 1::2] *= sy
    clip_boxes_(boxes, (out_h, out_w))
    keep = nonempty(boxes)
    if return_keep:
        return boxes[keep], scores[keep], classes[keep], keep
    return boxes[keep], scores[keep], classes[keep]


# ------------------------------------------------------------------ whole model
class RCNNSpec:
    """The few config values the inference path reads (Appendix B of SURVEY.md)."""

    def __init__(self, depth=50, num_classes=80, pixel_mean=(103.53, 116.28, 123.675), pixel_std=(1.0, 1.0, 1.0),
                 anchor_sizes=((32,), (64,), (128,), (256,), (512,)), aspect_ratios=((0.5, 1.0, 2.0),),
                 rpn_nms=0.7, pre_topk=1000, post_topk=1000, score_thresh=0.05, det_nms=0.5, dets_per_image=100,
                 box_weights=(10.0, 10.0, 5.0, 5.0), cosine=False, cosine_scale=20.0, num_fc=2,
                 cls_agnostic=False):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def generalized_rcnn_inference(sd, spec, batched_inputs, return_intermediates=False, roi_align=None, feats=None):
    """lvc/modeling/meta_arch/rcnn.py:177-322 (inference, do_postprocess=True), StandardROIHeads
    (lvc/modeling/roi_heads/roi_heads.py:554-629).  batched_inputs: list of {"image","height","width"}.
    The arithmetic type is the state_dict's (float32 = the reference's CPU path; a .double() state_dict gives the
    fp64 evaluation; NMS decisions are taken on the float32-rounded boxes in both).  feats: precomputed {p2..p6}
    (the post-trunk chain tests hand the SAME features to this function and to the HIP path)."""
    dt = sd["backbone.fpn_lateral2.weight"].dtype
    images, sizes = preprocess([b["image"] for b in batched_inputs], spec.pixel_mean, spec.pixel_std, 32, dtype=dt)
    if feats is None:
        feats = fpn(sd, resnet(sd, images, spec.depth))
    names = ["p2", "p3", "p4", "p5", "p6"]
    strides = [4, 8, 16, 32, 64]
    flist = [feats[n] for n in names]
    logits, deltas = rpn_head(sd, flist)
    ar = spec.aspect_ratios * len(names) if len(spec.aspect_ratios) == 1 else spec.aspect_ratios
    cell = [sd.get("proposal_generator.anchor_generator.cell_anchors.%d" % i, generate_cell_anchors(s, a)).to(dt)
            for i, (s, a) in enumerate(zip(spec.anchor_sizes, ar))]
    anchors = grid_anchors(cell, [f.shape[-2:] for f in flist], strides)
    lg, dl = flatten_rpn_outputs(logits, deltas)
    proposals = find_top_rpn_proposals(anchors, lg, dl, sizes, spec.rpn_nms, spec.pre_topk, spec.post_topk)
    box_lists = [p[0] for p in proposals]
    pooled = roi_pool(flist[:4], [1.0 / s for s in strides[:4]], box_lists, roi_align=roi_align)
    hfeat = box_head(sd, pooled, spec.num_fc)
    scores, pdeltas = predictor(sd, hfeat, spec.cosine, spec.cosine_scale)
    all_boxes = torch.cat(box_lists, 0)
    K = pdeltas.shape[1] // 4
    pred = apply_deltas(pdeltas.view(-1, 4), all_boxes.unsqueeze(1).expand(-1, K, 4).reshape(-1, 4), spec.box_weights)
    pred = pred.view(len(all_boxes), K * 4)
    probs = F.softmax(scores, dim=-1)
    counts = [len(b) for b in box_lists]
    results = []
    for bi, (pb, pp) in enumerate(zip(pred.split(counts), probs.split(counts))):
        b, s, c, rows = fast_rcnn_inference_single_image(pb, pp, sizes[bi], spec.score_thresh, spec.det_nms,
                                                         spec.dets_per_image)
        h = batched_inputs[bi].get("height", sizes[bi][0])
        w = batched_inputs[bi].get("width", sizes[bi][1])
        b, s, c, kept = detector_postprocess(b, s, c, sizes[bi], h, w, return_keep=True)
        # "rows": the proposal index each detection came from (fast_rcnn.py:131-137 returns it as the second value)
        results.append({"pred_boxes": b, "scores": s, "pred_classes": c, "rows": rows[kept]})
    if return_intermediates:
        return results, {"images": images, "feats": feats, "rpn_logits": logits, "rpn_deltas": deltas,
                         "proposals": proposals, "pooled": pooled, "head": hfeat, "cls_logits": scores,
                         "box_deltas": pdeltas}
    return results
