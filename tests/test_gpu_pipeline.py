"""Batches in flight on several HIP streams (lvc_amd/evaluation.py) give the detections of the one-stream loop, bit for
bit, in submission order -- including with the stream-K conv workers of two launches sharing the chip."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pipelined_inference_equals_sequential():
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.evaluation import inference_on_dataset
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    model = build_model(base_rcnn_fpn()).eval()
    syn.conditioned_r50_fpn_(model)
    dev = torch.device("cuda:0")
    loader = []
    for b in range(6):
        h, w = (416, 608) if b % 2 else (384, 640)
        loader.append([{"image": syn.synthetic_image(10 + 2 * b + i, h, w).to(dev), "height": 2 * h, "width": 2 * w} for i in range(2)])
    with torch.no_grad():
        seq = [model(batch) for batch in loader]
    for depth in (2, 3):
        got = list(inference_on_dataset(model, loader, depth=depth))
        assert len(got) == len(loader)
        for (inputs, outs), ref, batch in zip(got, seq, loader):
            assert inputs is batch
            for o, r in zip(outs, ref):
                a, b = o["instances"], r["instances"]
                assert a.image_size == b.image_size and len(a) == len(b) > 0
                assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor)
                assert torch.equal(a.scores, b.scores) and torch.equal(a.pred_classes, b.pred_classes)


@pytest.mark.parametrize("kind", ["image_u8", "image", "raw"])
def test_evaluation_loop_from_host_memory_equals_device_resident_inputs(kind):
    """The loop as the reference runs it (lvc/evaluation/evaluator.py:85-157: the loader hands over HOST tensors): pinned host
    inputs -- uint8 CHW `image` as the reference's DatasetMapper makes them (dataset_mapper.py:164), the same as float32, or the
    decoded uint8 HWC file under `raw` (the mapper's resize done on the device) -- copied on the batch's own stream while the other stream computes, give the
    detections of the same tensors already resident in HBM, bit for bit, batch after batch."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.evaluation import inference_on_dataset
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    model = build_model(base_rcnn_fpn()).eval()
    syn.conditioned_r50_fpn_(model)
    dev = torch.device("cuda:0")
    host, resident = [], []
    for b in range(5):
        hb, db = [], []
        for i in range(2):
            if kind in ("image", "image_u8"):
                h, w = (416, 608) if b % 2 else (384, 640)
                t = syn.synthetic_image(20 + 2 * b + i, h, w)
                if kind == "image_u8":
                    t = t.round().clamp(0, 255).to(torch.uint8)
                extra = {"height": 2 * h, "width": 2 * w}
            else:
                h, w = (150, 200) if b % 2 else (180, 160)
                t = syn.synthetic_image(20 + 2 * b + i, h, w).permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).contiguous()
                extra = {"height": h, "width": w}
            key = "raw" if kind == "raw" else "image"
            hb.append(dict({key: t.pin_memory()}, **extra))
            db.append(dict({key: t.to(dev)}, **extra))
        host.append(hb)
        resident.append(db)
    with torch.no_grad():
        ref = [model(batch) for batch in resident]
    if kind == "image_u8":      # uint8 pixels are the float32 pixels of the same values: same detections as the fp32 tensors of those values
        with torch.no_grad():
            same = model([dict(d, image=d["image"].float()) for d in resident[0]])
        for o, r in zip(same, ref[0]):
            assert torch.equal(o["instances"].pred_boxes.tensor, r["instances"].pred_boxes.tensor)
    got = list(inference_on_dataset(model, host, depth=2))
    assert len(got) == len(host)
    for (inputs, outs), want, batch in zip(got, ref, host):
        assert inputs is batch and not inputs[0]["raw" if kind == "raw" else "image"].is_cuda
        for o, r in zip(outs, want):
            a, b = o["instances"], r["instances"]
            assert a.image_size == b.image_size and len(a) == len(b) > 0
            assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor)
            assert torch.equal(a.scores, b.scores) and torch.equal(a.pred_classes, b.pred_classes)


def test_graphed_inference_is_bit_identical_to_eager():
    """`GraphedInference`: the whole forward as one hipGraph launch -- captured once, replayed on new images of the same
    shapes; outputs equal the eager call's bit for bit, replay after replay."""
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.evaluation import GraphedInference
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    model = build_model(base_rcnn_fpn()).eval()
    syn.conditioned_r50_fpn_(model)
    dev = torch.device("cuda:0")
    batches = [[{"image": syn.synthetic_image(20 + 3 * b + i, 416, 608).to(dev), "height": 832, "width": 1216} for i in range(3)]
               for b in range(3)]
    with torch.no_grad():
        eager = [[t.clone() for t in model.inference_batched(b)] for b in batches]
    g = GraphedInference(model, batches[0])
    for rnd in range(2):
        for b, ref in zip(batches, eager):
            out = g.replay(b)
            torch.cuda.synchronize()
            for a, r in zip(out, ref):
                assert torch.equal(a, r)
            inst = g.instances()
            assert len(inst) == 3 and all(len(x["instances"]) == int(c) for x, c in zip(inst, ref[3].tolist()))
