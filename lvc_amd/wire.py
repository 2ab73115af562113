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


# ------------------------------------------------------------------------------------------------------------------
# Pseudo-label files between the detector and the verification step.  Pure host code on json rows.
#   detector -> `coco_instances_results.json` rows (instances_to_coco_json above)
#            -> select_pseudo_labels / save_pseudo_label_dataset   (reference tools/create_coco_dataset_from_dets_all.py)
#            -> crops -> descriptors -> kNN -> keep ids -> save_verified_dataset (reference tools/run_nearest_neighbours.py)
AREA_RNG = (0.0, 1e5 ** 2)   # reference create_coco_dataset_from_dets_all.py:56


class DetectionIndex:
    """COCO-results rows joined with the image table: the state the reference tool works on after
    `COCO.loadRes` (pycocotools: id = position + 1, area = w * h, iscrowd = 0) and `COCO_PK.createIndex`
    (create_coco_dataset_from_dets_all.py:18-31: area_ratio = area / (H * W)).  Rows are modified in place, as there."""

    def __init__(self, images, rows, categories=None, full_dataset=False):
        self.images = list(images)
        self.by_id = {im["id"]: im for im in self.images}
        self.categories = categories
        self.rows = rows
        if not full_dataset:
            for pos, r in enumerate(rows):
                r["area"] = r["bbox"][2] * r["bbox"][3]
                r["id"] = pos + 1
                r["iscrowd"] = 0
        self.ann = {r["id"]: r for r in rows}
        self.by_image = {}
        for r in rows:
            self.by_image.setdefault(r["image_id"], []).append(r)
        for iid, lst in self.by_image.items():
            im = self.by_id[iid]
            tot = float(im["height"]) * float(im["width"])
            for r in lst:
                r["area_ratio"] = r["area"] / tot

    def image_ids(self):
        return [im["id"] for im in self.images]

    def ann_ids(self, cat_id, image_ids, ar_lo):
        """COCO_PK.getAnnIds(catIds=cid, imgIds=..., areaRng=AREA_RNG, areaRatioRng=[ar, 1.0], iscrowd=False) (:33-46)."""
        out = []
        # pycocotools: an EMPTY image list means "no image filter" (every row, in file order) -- the reference tool relies
        # on it when a category has no detection in the score window (`full`: all its detections become ignore rows)
        groups = [self.rows] if len(image_ids) == 0 else [self.by_image.get(iid, ()) for iid in image_ids]
        for grp in groups:
            for r in grp:
                if (r["category_id"] == cat_id and AREA_RNG[0] < r["area"] < AREA_RNG[1] and r["iscrowd"] == False  # noqa: E712
                        and ar_lo < r["area_ratio"] < 1.0):
                    out.append(r["id"])
        return out


def select_pseudo_labels(index, category_ids, train_imgs, K_min, K_max, top=False, full=False, ar=0.0):
    """Reference get_ret_anns (create_coco_dataset_from_dets_all.py:129-207): per novel category, the detections outside
    that category's few-shot training images, sorted by score; `top`: ranks [K_max, K_min); else scores in
    (K_min, K_max] located with searchsorted on the negated scores.  `full` also returns the other detections of the
    selected images, flagged ignore_qe = iscrowd = 1.  Returns the annotation rows in the reference's order."""
    import numpy as np

    out = []
    for cid in category_ids:
        banned = train_imgs.get(cid, ())
        valid = [i for i in index.image_ids() if i not in banned]
        anns = sorted((index.ann[a] for a in index.ann_ids(cid, valid, ar)), key=lambda r: r["score"], reverse=True)
        if top:
            sel = anns[int(K_max): int(K_min)]
            if not full:
                out.extend(sel)
                continue
        else:
            neg = -np.array([r["score"] for r in anns])
            sel = anns[int(np.searchsorted(neg, -float(K_max))): int(np.searchsorted(neg, -float(K_min)))]
        for r in sel:
            r["ignore_qe"] = 0
            r["iscrowd"] = 0
        if full:
            present = list(set(r["image_id"] for r in sel))          # set order, as the reference iterates it
            others = list(set(index.ann_ids(cid, present, ar)) - set(r["id"] for r in sel))
            rest = [index.ann[a] for a in others]
            for r in rest:
                r["ignore_qe"] = 1
                r["iscrowd"] = 1
            out.extend(rest)
        out.extend(sel)
    return out


def pseudo_label_filename(dt_path, K_min, K_max, top=False, full=False, ar=0.0, all_cats=False):
    """File-name rule of the reference's save_coco (create_coco_dataset_from_dets_all.py:224-238)."""
    s = "_ar{}".format(str(ar).replace(".", "")) if ar else ""
    if top:
        s += "_top_max{}_min{}{}_all.json".format(str(int(K_max)).zfill(4), str(int(K_min)).zfill(4), "_full" if full else "")
    else:
        s += "_score_max{}_min{}{}_all.json".format(str(K_max).replace(".", ""), str(K_min).replace(".", ""), "_full" if full else "")
    name = dt_path.replace(".json", s)
    return name.replace(".json", "_allcats.json") if all_cats else name


def save_pseudo_label_dataset(meta_dataset, annotations, images, dt_path, K_min, K_max, top=False, full=False, ar=0.0,
                              all_cats=False):
    """Reference save_coco (:210-243): every key of `meta_dataset` except annotations/images, the selected annotations
    without `segmentation` / `top2_*`, the images they live in; json with indent 4 and sorted keys.  Returns the path."""
    import json

    doc = {k: v for k, v in meta_dataset.items() if k not in ("annotations", "images")}
    for r in annotations:
        for k in ("segmentation", "top2_scores", "top2_inds"):
            r.pop(k, None)
    doc["annotations"] = annotations
    doc["images"] = images
    name = pseudo_label_filename(dt_path, K_min, K_max, top, full, ar, all_cats)
    with open(name, "w") as fp:
        fp.write(json.dumps(doc, indent=4, sort_keys=True))
    return name


def create_coco_dataset_from_dets(gt_dataset, category_dataset, results_rows, train_imgs, dt_path, K_min, K_max, top=False,
                                  full=False, ar=0.0, category_ids=None, full_dataset=False, all_cats=False):
    """The reference tool's main() (:240-263) on loaded json: `gt_dataset` supplies the image table, `category_dataset`
    the categories and every other top-level key of the output, `results_rows` the detector's rows.  With `full_dataset`
    (`--full-dataset`: the detections file is itself a COCO document, `COCO_PK.loadRes` :48-50) `results_rows` is that
    document: its `annotations` are complete rows indexed over ITS image table, and save_coco (:196-200) copies ITS
    top-level keys; the returned images still come from `gt_dataset` (`coco_gt.loadImgs`, :259).  `all_cats`
    (`--all-cats`): every category id in ascending order instead of the novel ones (:256) and the `_allcats` file suffix."""
    if full_dataset:
        if not isinstance(results_rows, dict):
            raise TypeError("full_dataset: pass the loaded detections document (a COCO dict with annotations / images)")
        dt_doc = results_rows
        index = DetectionIndex(dt_doc["images"], dt_doc["annotations"], category_dataset.get("categories"), True)
    else:
        dt_doc = None
        index = DetectionIndex(gt_dataset["images"], results_rows, category_dataset.get("categories"), False)
    if category_ids is not None:
        cats = category_ids
    elif all_cats:
        cats = sorted(c["id"] for c in category_dataset["categories"])
    else:
        cats = novel_category_ids(category_dataset["categories"])
    anns = select_pseudo_labels(index, cats, train_imgs, K_min, K_max, top, full, ar)
    img_ids = list(set(r["image_id"] for r in anns))
    gt_by_id = {im["id"]: im for im in gt_dataset["images"]}
    images = [gt_by_id[i] for i in img_ids]
    meta = dt_doc if full_dataset else category_dataset
    return save_pseudo_label_dataset(meta, anns, images, dt_path, K_min, K_max, top, full, ar, all_cats), anns


VOC_NOVEL_NAMES = ("airplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "dining table", "dog",
                   "horse", "motorcycle", "person", "potted plant", "sheep", "couch", "train", "tv")


def novel_category_ids(categories):
    """COCO ids of the 20 novel (VOC) classes, in category-table order (reference get_ids_names :82-126)."""
    return [c["id"] for c in categories if c["name"] in VOC_NOVEL_NAMES]


def save_verified_dataset(qe_dset, keep_ids, nn_model, knn, cosine):
    """Reference tools/run_nearest_neighbours.py:230-249 save_coco: the pseudo-label file reduced to the annotations the
    kNN vote kept (in keep order) and the images they live in; name suffix _{model}_{knn:02d}_{cosine|euclid}.json."""
    import json

    with open(qe_dset, "r") as fp:
        doc = json.load(fp)
    by_id = {a["id"]: a for a in doc["annotations"]}
    imgs = {im["id"]: im for im in doc["images"]}
    anns = [by_id[i] for i in keep_ids]
    doc["annotations"] = anns
    doc["images"] = [imgs[i] for i in list(set(a["image_id"] for a in anns))]
    name = qe_dset.replace(".json", "_{}_{}_{}.json".format(nn_model.replace("/", ""), str(knn).zfill(2), "cosine" if cosine else "euclid"))
    with open(name, "w") as fp:
        json.dump(doc, fp)
    return name
