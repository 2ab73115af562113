"""Wire formats and data ops either side of the two hot paths (SURVEY.md section 8(f) "next" rows).

* `instances_to_coco_json` -- detector output -> COCO results json rows (reference
  lvc/evaluation/coco_evaluation.py:566-604: XYXY -> XYWH, python floats, optional top2 fields), the format
  `tools/create_coco_dataset_from_dets_all.py` and `tools/run_nearest_neighbours.py` consume.
* `get_crops_qe` -- the 224x224 crops fed to the descriptor network (reference lvc/data/utils.py:485-519):
  window arithmetic on the host (integers), zero-pad-to-square + nearest resize in one HIP kernel.
"""
import torch

from . import kernels as K
from .structures import BoxMode


def instances_to_coco_json(instances, img_id):
    n = len(instances)
    if n == 0:
        return []
    boxes = BoxMode.convert(instances.pred_boxes.tensor.detach().cpu().clone(), BoxMode.XYXY_ABS, BoxMode.XYWH_ABS).tolist()
    scores = instances.scores.tolist()
    classes = instances.pred_classes.tolist()
    extra = instances.has("top2_scores")
    if extra:
        t2s, t2i = instances.top2_scores.tolist(), instances.top2_inds.tolist()
    out = []
    for k in range(n):
        row = {"image_id": img_id, "category_id": classes[k], "bbox": boxes[k], "score": scores[k]}
        if extra:
            row.update({"top2_scores": t2s[k], "top2_inds": t2i[k]})
        out.append(row)
    return out


def get_padding(H, W):
    """reference lvc/data/utils.py:485-499: (left, right, top, bottom) pads that make H x W square."""
    d = max(H, W)
    hp, vp = (d - W) / 2, (d - H) / 2
    l = hp if hp % 1 == 0 else hp + 0.5
    t = vp if vp % 1 == 0 else vp + 0.5
    r = hp if hp % 1 == 0 else hp - 0.5
    b = vp if vp % 1 == 0 else vp - 0.5
    return int(l), int(r), int(t), int(b)


def crop_windows(boxes_xyxy, height, width, operation="pad"):
    """Integer window bookkeeping of get_crops_qe: rows (x1,y1,x2,y2,l_pad,t_pad,side_w,side_h)."""
    rows = []
    for x1, y1, x2, y2 in boxes_xyxy:
        if operation == "pad":
            l, r, t, b = get_padding(y2 - y1 + 1, x2 - x1 + 1)
            xa, ya, xb, yb = x1, y1, x2, y2
        elif operation == "context":
            l, r, t, b = get_padding(y2 - y1 + 1, x2 - x1 + 1)
            ya, xa = max(0, y1 - t), max(0, x1 - l)
            yb, xb = min(height, y2 + b), min(width, x2 + r)
            l, r, t, b = get_padding(yb - ya + 1, xb - xa + 1)
        else:
            raise ValueError(operation)
        # python slicing imgs[..., ya:yb+1, xa:xb+1] clamps to the image
        yb, xb = min(yb, height - 1), min(xb, width - 1)
        rows.append([xa, ya, xb, yb, l, t, (xb - xa + 1) + l + r, (yb - ya + 1) + t + b])
    return rows


def get_crops_qe(imgs, gt_boxes, operation="pad"):
    """Reference signature: imgs [1,C,H,W] device tensor, gt_boxes list of Instances each holding ONE gt box."""
    H, W = int(imgs.shape[-2]), int(imgs.shape[-1])
    boxes = [b.gt_boxes.tensor.long().squeeze(0).tolist() for b in gt_boxes]
    win = torch.tensor(crop_windows(boxes, H, W, operation), dtype=torch.int32, device=imgs.device).reshape(-1, 8)
    return K.crop_resize_nearest(imgs[0], win, 224)
