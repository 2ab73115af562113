"""Limits the reference does not have must degrade, not fail:
  * batched NMS on more than 16 384 boxes per image (reference: per-class loop from 40 000 boxes, nms.py:22-29) --
    global sort + blocked greedy pass, keep indices still bit-exact;
  * SCORE_THRESH_TEST = 0: every (roi, class) pair is a candidate (1000 x 80 per image) -- the detector re-runs the
    batch with the candidate list sized for R*K and keeps going;
  * an activation beyond fp16's range -- the pass is repeated on the range-free bf16x3 kernels."""
import logging

import pytest
import torch

pytestmark = pytest.mark.gpu


def _boxes(g, n, span=2000.0):
    c = torch.rand(n, 2, generator=g) * span
    wh = 8 + torch.rand(n, 2, generator=g) * 150
    return torch.cat([c, c + wh], 1)


@pytest.mark.parametrize("n,classes,thr", [(20000, 1, 0.5), (40000, 7, 0.7), (70001, 80, 0.5), (16385, 3, 0.3)])
def test_large_nms_keep_indices_exact(n, classes, thr):
    from lvc_amd import kernels as K
    from oracle import ops as oops

    g = torch.Generator().manual_seed(n)
    boxes = _boxes(g, n)
    scores = torch.rand(n, generator=g)
    scores[500:520] = scores[500]                       # exact ties -> lower index first
    idxs = torch.randint(0, classes, (n,), generator=g)
    ref = oops.batched_nms(boxes, scores, idxs, thr)
    keep, nk = K.batched_nms_batch(boxes[None].cuda(), scores[None].cuda(), idxs.to(torch.int32)[None].cuda(), None, thr)
    got = keep[0, : int(nk.item())].cpu().long()
    assert len(ref) > 16384 or n < 30000
    assert torch.equal(got, ref)
    # the drop-in op
    got2 = torch.ops.lvc_amd.batched_nms(boxes.cuda(), scores.cuda(), idxs.cuda(), thr).cpu()
    assert torch.equal(got2, ref)


def test_large_nms_ragged_counts_and_max_keep():
    """Two images in one call, capacity 50 000, real counts 33 000 and 9 000 (the second fits the first block and skips
    every later stage), max_keep cuts the list where the reference's `keep[:k]` does."""
    from lvc_amd import kernels as K
    from oracle import ops as oops

    g = torch.Generator().manual_seed(77)
    Nmax, counts = 50000, [33000, 9000]
    boxes = torch.zeros(2, Nmax, 4)
    scores = torch.zeros(2, Nmax)
    idxs = torch.zeros(2, Nmax, dtype=torch.int32)
    refs = []
    for b, n in enumerate(counts):
        boxes[b, :n] = _boxes(g, n, 1200.0)
        scores[b, :n] = torch.rand(n, generator=g)
        idxs[b, :n] = torch.randint(0, 5, (n,), generator=g).int()
        refs.append(oops.batched_nms(boxes[b, :n], scores[b, :n], idxs[b, :n], 0.6))
    cnt = torch.tensor(counts, dtype=torch.int32).cuda()
    keep, nk = K.batched_nms_batch(boxes.cuda(), scores.cuda(), idxs.cuda(), cnt, 0.6)
    for b in range(2):
        assert torch.equal(keep[b, : int(nk[b])].cpu().long(), refs[b]), b
    keep, nk = K.batched_nms_batch(boxes.cuda(), scores.cuda(), idxs.cuda(), cnt, 0.6, max_keep=2500)
    for b in range(2):
        k = min(2500, len(refs[b]))
        assert int(nk[b]) == k and torch.equal(keep[b, :k].cpu().long(), refs[b][:k]), b


def test_score_thresh_zero_degrades_to_full_capacity(caplog):
    """SCORE_THRESH_TEST = 0.0 (AP-style dumps): 1000 x 80 = 80 000 candidates per image.  The first pass overflows the
    16 384-row list, the model re-runs the batch with R*K rows (logged once) and the detections equal the oracle's."""
    from helpers import match_fraction, r50_state_dict
    from lvc_amd.utils import synthetic as syn
    from oracle import rcnn as orc
    from test_gpu_e2e import _model

    model = _model()
    model.roi_heads.test_score_thresh = 0.0
    assert model.roi_heads.det_max_candidates == 16384
    inputs = [{"image": syn.synthetic_image(3, 240, 320), "height": 240, "width": 320},
              {"image": syn.synthetic_image(4, 200, 352), "height": 200, "width": 352}]
    with caplog.at_level(logging.WARNING, logger="lvc_amd"), torch.no_grad():
        out = model(inputs)
    assert model.roi_heads.det_max_candidates is None
    assert any("candidates" in r.getMessage() for r in caplog.records)
    spec = orc.RCNNSpec(score_thresh=0.0)
    with torch.no_grad():
        ref = orc.generalized_rcnn_inference(r50_state_dict(), spec, inputs)
    for o, r in zip(out, ref):
        inst = o["instances"].to("cpu")
        assert len(inst) == len(r["scores"]) == 100
        frac, wb, ws = match_fraction(inst.pred_boxes.tensor, inst.scores, inst.pred_classes, r["pred_boxes"], r["scores"],
                                      r["pred_classes"], box_tol=0.1, score_tol=2e-3)
        print("thresh 0: matched %.0f%% (worst box %.1e px, score %.1e)" % (100 * frac, wb, ws))
        assert frac >= 0.9
    with torch.no_grad():     # second call: no further overflow, same answer
        again = model(inputs)
    assert torch.equal(again[0]["instances"].pred_boxes.tensor, out[0]["instances"].pred_boxes.tensor)


def test_activation_beyond_fp16_range_falls_back_to_bf16x3(monkeypatch, caplog):
    """An image scaled so that the normalised pixels exceed 65504: the fp16x2 stem raises the range word, the model
    repeats the pass on the bf16x3 kernels (logged) and the trunk equals the fp32 oracle on that input."""
    from helpers import r50_state_dict
    from lvc_amd import kernels as K
    from lvc_amd.utils import synthetic as syn
    from oracle import rcnn as orc
    from test_gpu_e2e import _model

    monkeypatch.setattr(K, "CONV_SPLIT", "f16x2")
    monkeypatch.setattr(K, "_RANGE_FALLBACK_LOGGED", False)
    model = _model()
    img = syn.synthetic_image(3, 160, 192)
    img[:, 40:60, 50:70] = 1.0e5
    inputs = [{"image": img}]
    with caplog.at_level(logging.WARNING, logger="lvc_amd"), torch.no_grad():
        out = model(inputs)
    assert K.CONV_SPLIT == "bf16x3"
    assert any("fp16" in r.getMessage() for r in caplog.records)
    sd = r50_state_dict()
    spec = orc.RCNNSpec()
    with torch.no_grad():
        imgs, _ = orc.preprocess([img], spec.pixel_mean, spec.pixel_std, 32)
        ref = orc.fpn(sd, orc.resnet(sd, imgs, 50))
        got = model.backbone(model.preprocess_image(inputs).tensor)
    for k in ref:
        err = float((got[k].cpu() - ref[k]).abs().max()) / float(ref[k].abs().max())
        print(k, "relative error on the bf16x3 fallback", err)
        assert err <= 2e-4, k
    assert len(out[0]["instances"]) >= 0


def test_pipelined_two_consecutive_overflows(caplog, monkeypatch):
    """Batches in flight (`PipelinedInference`, depth 2 and 3): every batch submitted under the narrow limits overflows
    on its own when it is collected -- the SECOND one after the limit is already widened -- and each is re-run; an
    fp16-range batch on one stream is neither blamed on, nor hidden by, the batch on the other stream."""
    from lvc_amd import kernels as K
    from lvc_amd.evaluation import inference_on_dataset
    from lvc_amd.utils import synthetic as syn
    from test_gpu_e2e import _model

    dev = torch.device("cuda:0")
    loader = [[{"image": syn.synthetic_image(30 + 2 * b + i, 200, 320).to(dev), "height": 200, "width": 320}
               for i in range(2)] for b in range(4)]
    for depth in (2, 3):
        model = _model()
        model.roi_heads.test_score_thresh = 0.0
        with torch.no_grad():
            ref = [model(batch) for batch in loader]          # sequential: first call widens, rest run wide
        assert model.roi_heads.det_max_candidates is None
        model.roi_heads.det_max_candidates = 16384            # narrow again: `depth` batches launch under it
        got = list(inference_on_dataset(model, loader, depth=depth))
        assert model.roi_heads.det_max_candidates is None and len(got) == len(loader)
        for (_, outs), r in zip(got, ref):
            for o, q in zip(outs, r):
                a, b = o["instances"], q["instances"]
                assert len(a) == len(b) == 100
                assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)
    # range overflow: batches 1 and 2 (consecutive, different streams) carry an out-of-range image
    monkeypatch.setattr(K, "CONV_SPLIT", "f16x2")
    monkeypatch.setattr(K, "_RANGE_FALLBACK_LOGGED", False)
    model = _model()
    hot = []
    for b in range(4):
        batch = []
        for i in range(2):
            img = syn.synthetic_image(50 + 2 * b + i, 160, 192)
            if b in (1, 2) and i == 0:
                img[:, 40:60, 50:70] = 1.0e5
            batch.append({"image": img.to(dev)})
        hot.append(batch)
    got = list(inference_on_dataset(model, hot, depth=2))
    assert K.CONV_SPLIT == "bf16x3"
    with torch.no_grad():
        ref = [model(batch) for batch in hot]                 # all on the range-free kernels now
    for b, ((_, outs), r) in enumerate(zip(got, ref)):
        for o, q in zip(outs, r):
            a, q = o["instances"], q["instances"]
            assert len(a) == len(q)
            if b >= 1:   # batch 0 legitimately ran on the fp16x2 kernels; from the first re-run on everything is bf16x3
                assert torch.equal(a.pred_boxes.tensor, q.pred_boxes.tensor) and torch.equal(a.scores, q.scores)


def test_layers_beyond_2_29_elements_run_in_image_groups(caplog, monkeypatch):
    """A layer whose tensors reach 2^29 elements (batch 32 on the p2 map: 2.2 GB of fp32) is outside the 32-bit buffer
    descriptors of the split-precision kernels: it must run in image groups that stay inside them -- same kernels, same
    values -- and say so once (pointwise with residual, and the 3x3 kernel)."""
    from lvc_amd import kernels as K

    monkeypatch.setattr(K, "_LOGGED_ONCE", set())
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(32, 200, 336, 256, device=dev, generator=g)
    res = torch.randn(32, 200, 336, 256, device=dev, generator=g)
    w = torch.randn(256, 256, 1, 1, device=dev, generator=g) * 0.06
    pc = K.pack_conv(w)
    with caplog.at_level(logging.WARNING, logger="lvc_amd"):
        y = K.conv2d_nhwc(x, pc, relu=True, residual=res)
        K.conv2d_nhwc(x, pc, relu=True, residual=res, out=y)
    assert y.numel() >= 1 << 29
    msgs = [r.getMessage() for r in caplog.records if "groups of" in r.getMessage()]
    assert len(msgs) == 1, msgs                                   # logged once
    ys = K.conv2d_nhwc(x[30:32].contiguous(), pc, relu=True, residual=res[30:32].contiguous())
    torch.cuda.synchronize()
    assert K.conv_error_word(dev) == 0
    sc = float(ys.abs().max())
    assert float((y[30:32] - ys).abs().max()) <= 4e-6 * sc        # the same kernel on the same images
    ref = torch.relu(x[31, 100, :8].double() @ w[:, :, 0, 0].double().t() + res[31, 100, :8].double())
    assert float((y[31, 100, :8].double() - ref).abs().max()) <= 2e-5 * sc
    del res, ys
    w3 = torch.randn(256, 256, 3, 3, device=dev, generator=g) * 0.02
    pc3 = K.pack_conv(w3, stride=1, pad=1)
    K.conv2d_nhwc(x, pc3, relu=True, out=y)
    ys = K.conv2d_nhwc(x[5:6].contiguous(), pc3, relu=True)
    torch.cuda.synchronize()
    assert K.conv_error_word(dev) == 0
    assert float((y[5:6] - ys).abs().max()) <= 4e-6 * float(ys.abs().max())
