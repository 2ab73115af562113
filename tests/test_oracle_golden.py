"""The oracle restatement (oracle/) against golden vectors produced by the imported reference
(oracle/make_golden.py).  CPU-only; this is what pins the oracle."""
import math

import pytest
import torch

from helpers import gold, match_detections, r50_state_dict, r101_state_dict
from oracle import ops as oops
from oracle import rcnn as orc


def test_roi_align_matches_reference_kernel():
    g = gold("roi_align")
    for key in [k for k in g if k.startswith("out_")]:
        _, s, a, sr = key.split("_")
        out = oops.roi_align_forward(g["feat"], g["rois"], float(s[1:]), 7, 7, int(sr[2:]), bool(int(a[1:])))
        assert torch.equal(out, g[key]), key  # same arithmetic, same order: bit-exact


def test_roi_align_backward_matches_reference_kernel():
    """orc_roi_align_backward against the reference's own `_C.roi_align_backward` (CPU, csrc/vision.cpp:97)."""
    g = gold("roi_align")
    keys = [k for k in g if k.startswith("bwd_")]
    assert len(keys) == 3
    for key in keys:
        _, s, a, sr = key.split("_")
        gin = oops.roi_align_backward(g["grad"], g["rois"], float(s[1:]), 7, 7, 2, 8, 50, 84, int(sr[2:]), bool(int(a[1:])))
        assert torch.equal(gin, g[key]), key  # same single-threaded accumulation order: bit-exact


def test_roi_align_negative_size_raises():
    feat = torch.zeros(1, 1, 8, 8)
    with pytest.raises(RuntimeError):
        oops.roi_align_forward(feat, torch.tensor([[0, 5, 5, 2, 2.0]]), 1.0, 7, 7, 0, True)


def test_nms_matches_reference():
    g = gold("nms")
    assert oops.batched_nms(g["boxes"], g["scores"], g["idxs"], 0.7).tolist() == g["keep_thr07"].tolist()
    assert oops.batched_nms(g["boxes"], g["scores"], g["idxs"], 0.5).tolist() == g["keep_thr05"].tolist()


def test_anchors():
    g = gold("anchors")
    cell = [orc.generate_cell_anchors((s,), (0.5, 1.0, 2.0)) for s in (32, 64, 128, 256, 512)]
    assert torch.equal(torch.stack(cell), g["cell_anchors"])
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    anchors = orc.grid_anchors(cell, shapes, [4, 8, 16, 32, 64])
    assert [len(a) for a in anchors] == [201600, 50400, 12600, 3150, 819]
    for i, a in enumerate(anchors):
        assert torch.equal(a[:8], g["lvl%d_first" % i]) and torch.equal(a[-8:], g["lvl%d_last" % i])
        assert torch.equal(a.double().sum(0), g["lvl%d_sum" % i])


def test_apply_deltas_and_levels():
    g = gold("box_ops")
    assert torch.equal(orc.apply_deltas(g["deltas1"], g["boxes"], (1.0, 1.0, 1.0, 1.0)), g["out1"])
    assert torch.equal(orc.apply_deltas(g["deltas80"], g["boxes"], (10.0, 10.0, 5.0, 5.0)), g["out80"])
    assert torch.equal(orc.assign_boxes_to_levels(g["lvl_boxes"], 2, 5), g["levels"])


def test_find_top_rpn_proposals():
    g = gold("rpn_proposals")
    shapes = [tuple(s) for s in g["shapes"].tolist()]
    cell = [orc.generate_cell_anchors((s,), (0.5, 1.0, 2.0)) for s in (32, 64, 128, 256, 512)]
    anchors = orc.grid_anchors(cell, shapes, [4, 8, 16, 32, 64])
    res = orc.find_top_rpn_proposals(anchors, [g["logits%d" % i] for i in range(5)],
                                     [g["deltas%d" % i] for i in range(5)],
                                     [tuple(s) for s in g["image_sizes"].tolist()], 0.7, 300, 200)
    for n in range(2):
        assert torch.equal(res[n][0], g["out_boxes%d" % n])
        assert torch.equal(res[n][1], g["out_logits%d" % n])


def test_fast_rcnn_inference_and_postprocess():
    g = gold("fast_rcnn_inference")
    K = g["deltas"].shape[1] // 4
    boxes = orc.apply_deltas(g["deltas"].view(-1, 4), g["proposals"].unsqueeze(1).expand(-1, K, 4).reshape(-1, 4),
                             (10.0, 10.0, 5.0, 5.0)).view(-1, 4 * K)
    assert torch.equal(boxes, g["raw_boxes"])
    probs = torch.softmax(g["cls_logits"], -1)
    size = tuple(g["image_size"].tolist())
    b, s, c, rows = orc.fast_rcnn_inference_single_image(boxes, probs, size, 0.05, 0.5, 100)
    assert torch.equal(rows, g["kept_rows"])
    oh, ow = g["out_hw"].tolist()
    b, s, c = orc.detector_postprocess(b, s, c, size, oh, ow)
    assert torch.equal(b, g["out_boxes"]) and torch.equal(s, g["out_scores"]) and torch.equal(c, g["out_classes"])


def test_bottleneck_block():
    g = gold("bottleneck")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    sd = {"b." + k: v for k, v in sd.items()}
    y = orc._bottleneck(sd, "b", g["x"], stride=2)
    assert torch.allclose(y, g["y"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", ["bottleneck_res2_proj", "bottleneck_res2_identity"])
def test_bottleneck_block_res2_shapes(name):
    """The res2 blocks the HIP path runs as one launch (lvc_amd/csrc/conv_bneck.hip), reference outputs on a 13 x 37 map."""
    g = gold(name)
    sd = {"b." + k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    y = orc._bottleneck(sd, "b", g["x"], stride=1)
    assert torch.allclose(y, g["y"], rtol=0, atol=2e-6)


def _check_e2e(name, inputs, depth=50):
    g = gold(name)
    sd = r50_state_dict() if depth == 50 else r101_state_dict()
    with torch.no_grad():
        res, mid = orc.generalized_rcnn_inference(sd, orc.RCNNSpec(depth=depth), inputs, return_intermediates=True)
    for i in range(len(inputs)):
        pb, pl = mid["proposals"][i]
        assert pb.shape == g["prop_boxes%d" % i].shape
        assert (pb - g["prop_boxes%d" % i]).abs().max() <= 1e-3
        assert (pl - g["prop_logits%d" % i]).abs().max() <= 1e-4
        ok, msg = match_detections(res[i]["pred_boxes"], res[i]["scores"], res[i]["pred_classes"],
                                   g["det_boxes%d" % i], g["det_scores%d" % i], g["det_classes%d" % i])
        assert ok, msg
    return g, mid


def test_e2e_small_two_images():
    from lvc_amd.utils import synthetic as syn

    inputs = [{"image": syn.synthetic_image(3, 240, 320), "height": 480, "width": 640},
              {"image": syn.synthetic_image(4, 200, 352), "height": 200, "width": 352}]
    g, mid = _check_e2e("e2e_r50_fpn_small", inputs)
    for k in ("p2", "p3", "p4", "p5", "p6"):
        assert torch.allclose(mid["feats"][k][:, ::8], g["feat_" + k], rtol=0, atol=2e-4)


def test_e2e_800x1333():
    from lvc_amd.utils import synthetic as syn

    torch.set_num_threads(max(1, torch.get_num_threads()))
    inputs = [{"image": syn.synthetic_image(1), "height": 800, "width": 1333},
              {"image": syn.synthetic_image(2), "height": 800, "width": 1333}]
    g, mid = _check_e2e("e2e_r50_fpn_800x1333", inputs)
    for k in ("p2", "p3", "p4", "p5", "p6"):
        assert torch.allclose(mid["feats"][k][:, ::16, ::8, ::8], g["feat_" + k], rtol=0, atol=2e-4)


def test_e2e_r101_800x1333():
    """The oracle at depth 101 against the reference's CPU run of R101-FPN at the headline's image size
    (oracle/make_golden.py:gen_r101_full): what bench.py's r101 leg and the GPU tests are then held to."""
    from lvc_amd.utils import synthetic as syn

    inputs = [{"image": syn.synthetic_image(1), "height": 800, "width": 1333},
              {"image": syn.synthetic_image(2), "height": 800, "width": 1333}]
    g, mid = _check_e2e("e2e_r101_fpn_800x1333", inputs, depth=101)
    for k in ("p2", "p3", "p4", "p5", "p6"):
        assert torch.allclose(mid["feats"][k][:, ::16, ::8, ::8], g["feat_" + k], rtol=0, atol=2e-4)


def test_resize_oracle_matches_reference_transform_and_pillow():
    """the numpy restatement of Pillow's fixed-point bilinear resample (oracle/resize.py) against the
    reference's ResizeShortestEdge / ResizeTransform outputs (tests/golden/resize.npz) -- bit-exact -- and, where
    Pillow is importable, against Pillow itself on a fresh image."""
    import numpy as np
    from oracle import resize as orz

    g = gold("resize")
    for i in range(int(g["n"])):
        img = g["in%d" % i].numpy()
        short, mx, nh, nw = [int(v) for v in g["cfg%d" % i]]
        assert orz.shortest_edge_size(img.shape[0], img.shape[1], short, mx) == (nh, nw)
        out = orz.resize_bilinear_u8(img, nh, nw)
        assert np.array_equal(out, g["out%d" % i].numpy()), i
    try:
        from PIL import Image
    except ImportError:
        return
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    for nh, nw in [(160, 216), (40, 54), (97, 200), (30, 131)]:
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        assert np.array_equal(orz.resize_bilinear_u8(img, nh, nw), ref), (nh, nw)


def test_nms_restatement_has_an_independent_witness_in_the_reference_rotated_kernel():
    """torchvision is absent, so `orc_nms` (oracle.c) cannot be pinned against torchvision's own kernel.  The reference
    does carry ONE greedy-NMS implementation of its own: nms_rotated_cpu.cpp (declared there as a structural copy of
    torchvision's nms_cpu_kernel, :12), compiled unmodified into oracle/_ref/_C.so.  On axis-aligned boxes (angle 0) it
    must keep exactly the same indices as the restatement, provided no pairwise IoU lies within 1e-3 of the threshold
    (its polygon-clipping IoU differs from the closed form in the last bits, and it suppresses on `>=`, :54) and the
    scores are distinct (its sort is unstable)."""
    from oracle import refshim

    C = refshim.ref_C()
    if C is None:
        pytest.skip("oracle/_ref/_C.so not built (needs /root/reference once)")
    g = torch.Generator().manual_seed(123)
    for thr, n in ((0.5, 700), (0.7, 700), (0.3, 400)):
        c = torch.rand(n, 2, generator=g) * 300
        wh = 20 + torch.rand(n, 2, generator=g) * 100
        boxes = torch.cat([c, c + wh], 1)
        scores = (torch.randperm(n, generator=g).float() + 0.5) / n      # distinct by construction
        iou = oops.box_iou(boxes, boxes)
        near = ((iou - thr).abs() < 1e-3).any(dim=1)
        boxes, scores = boxes[~near], scores[~near]
        assert len(boxes) > n // 2
        keep = oops.nms(boxes, scores, thr)
        dets = torch.stack([(boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2, boxes[:, 2] - boxes[:, 0],
                            boxes[:, 3] - boxes[:, 1], torch.zeros(len(boxes))], 1).contiguous()
        ref = C.nms_rotated(dets, scores.contiguous(), thr)
        assert 10 < len(ref) < len(boxes)
        assert keep.tolist() == ref.tolist(), thr


def test_vit_restatement_against_torch_modules():
    """oracle/vit.py restates a third-party network (DINO ViT-S/8, absent here).  Independent witness of its arithmetic:
    the same weights run through torch's own nn.MultiheadAttention / nn.LayerNorm / nn.GELU modules wired as a pre-norm
    transformer block must give the same class token."""
    from oracle import vit as ovit

    g = torch.Generator().manual_seed(2)
    D, H, depth, P = 384, 6, 2, 8
    sd = {"patch_embed.proj.weight": torch.randn(D, 3, P, P, generator=g) * 0.05, "patch_embed.proj.bias": torch.randn(D, generator=g) * 0.1,
          "cls_token": torch.randn(1, 1, D, generator=g) * 0.1, "pos_embed": torch.randn(1, 1 + 16, D, generator=g) * 0.1,
          "norm.weight": 1 + 0.1 * torch.randn(D, generator=g), "norm.bias": 0.1 * torch.randn(D, generator=g)}
    for i in range(depth):
        p = "blocks.%d." % i
        for n, shp, s in (("norm1.weight", (D,), None), ("norm1.bias", (D,), 0.1), ("attn.qkv.weight", (3 * D, D), 0.05),
                          ("attn.qkv.bias", (3 * D,), 0.1), ("attn.proj.weight", (D, D), 0.05), ("attn.proj.bias", (D,), 0.1),
                          ("norm2.weight", (D,), None), ("norm2.bias", (D,), 0.1), ("mlp.fc1.weight", (4 * D, D), 0.05),
                          ("mlp.fc1.bias", (4 * D,), 0.1), ("mlp.fc2.weight", (D, 4 * D), 0.03), ("mlp.fc2.bias", (D,), 0.1)):
            sd[p + n] = (1 + 0.1 * torch.randn(shp, generator=g)) if s is None else torch.randn(shp, generator=g) * s
    x = torch.randn(2, 3, 32, 32, generator=g)
    got = ovit.vit_forward(sd, x, patch_size=P, num_heads=H)
    with torch.no_grad():
        t = torch.nn.functional.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=P).flatten(2).transpose(1, 2)
        t = torch.cat([sd["cls_token"].expand(2, -1, -1), t], 1) + sd["pos_embed"]
        for i in range(depth):
            p = "blocks.%d." % i
            mha = torch.nn.MultiheadAttention(D, H, bias=True, batch_first=True)
            mha.in_proj_weight.copy_(sd[p + "attn.qkv.weight"]); mha.in_proj_bias.copy_(sd[p + "attn.qkv.bias"])
            mha.out_proj.weight.copy_(sd[p + "attn.proj.weight"]); mha.out_proj.bias.copy_(sd[p + "attn.proj.bias"])
            ln1 = torch.nn.LayerNorm(D, eps=1e-6); ln1.weight.copy_(sd[p + "norm1.weight"]); ln1.bias.copy_(sd[p + "norm1.bias"])
            ln2 = torch.nn.LayerNorm(D, eps=1e-6); ln2.weight.copy_(sd[p + "norm2.weight"]); ln2.bias.copy_(sd[p + "norm2.bias"])
            y = ln1(t)
            t = t + mha(y, y, y, need_weights=False)[0]
            y = torch.nn.functional.linear(torch.nn.GELU()(torch.nn.functional.linear(ln2(t), sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])),
                                           sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
            t = t + y
        lnf = torch.nn.LayerNorm(D, eps=1e-6); lnf.weight.copy_(sd["norm.weight"]); lnf.bias.copy_(sd["norm.bias"])
        ref = lnf(t)[:, 0]
    assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
