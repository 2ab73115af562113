"""Import shim: makes the Python reference under /root/reference importable in THIS container
so that golden vectors can be generated from it (oracle/make_golden.py) and the oracle
restatement can be validated against it.  TEST INFRASTRUCTURE ONLY; cannot run on the GPU box
(no /root/reference there) and nothing in lvc_amd/ imports it.

The image lacks torchvision, fvcore, iopath, yacs, pycocotools, cv2, termcolor, timm, lvis,
mock, tensorboard (no network).  This module serves:
  * permissive dummy modules for those roots (data loading / visualisation / export code paths
    that the hot path never executes),
  * REAL behaviour for the few third-party pieces the hot path does execute:
      fvcore.common.registry.Registry        -> lvc_amd.utils.registry.Registry
      fvcore.common.config.CfgNode           -> lvc_amd.config.cfgnode.CfgNode
      fvcore.nn.weight_init                  -> c2_xavier_fill / c2_msra_fill (published fvcore
                                                definitions: kaiming_uniform(a=1), kaiming_normal
                                                (fan_out, relu); bias 0)
      fvcore.nn.smooth_l1_loss / giou_loss   -> published fvcore definitions
      torchvision.ops.nms / batched_nms / box_iou -> torchvision 0.8.2 algorithm via oracle.c
      detectron2._C                          -> oracle/_ref/_C.so, the reference's own csrc
  * torchvision.__version__ = "0.6.0" so detectron2/layers/roi_align.py:5-15 routes ROIAlign to
    the vendored `_C.roi_align_forward` (the compiled reference kernel) rather than torchvision.
sys.dont_write_bytecode is forced on so nothing is written into the read-only reference tree.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import math
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("DETECTRON2_DISABLE_CV2", "1")

REF_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)

_DUMMY_ROOTS = (
    "fvcore", "iopath", "yacs", "pycocotools", "cv2", "termcolor", "timm", "lvis", "mock",
    "tensorboard", "torchvision", "caffe2", "onnx", "matplotlib", "shapely",
)


class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _make_dummy(name)

    def __call__(cls, *a, **k):
        # decorator usage: @Dummy(...)(fn) or @Dummy
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return type.__call__(cls)


def _make_dummy(name):
    return _DummyMeta(name, (), {"__init__": lambda self, *a, **k: None,
                                 "__getattr__": lambda self, n: _make_dummy(n),
                                 "__call__": lambda self, *a, **k: (a[0] if a and callable(a[0]) else None)})


class _DummyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        obj = _make_dummy(name)
        setattr(self, name, obj)
        return obj


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _DUMMY_ROOTS and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _DummyModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


def _real(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _populate(module):
    fn = _REAL.get(module.__name__)
    if fn is not None:
        fn(module)


# ---------------------------------------------------------------- real pieces
def _fv_registry(m):
    from lvc_amd.utils.registry import Registry
    m.Registry = Registry


def _fv_config(m):
    from lvc_amd.config.cfgnode import CfgNode
    m.CfgNode = CfgNode


def _fv_weight_init(m):
    import torch.nn as nn

    def c2_xavier_fill(module):
        nn.init.kaiming_uniform_(module.weight, a=1)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)

    def c2_msra_fill(module):
        nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)

    m.c2_xavier_fill = c2_xavier_fill
    m.c2_msra_fill = c2_msra_fill


def _fv_nn(m):
    import torch

    def smooth_l1_loss(input, target, beta, reduction="none"):
        if beta < 1e-5:
            loss = torch.abs(input - target)
        else:
            n = torch.abs(input - target)
            cond = n < beta
            loss = torch.where(cond, 0.5 * n ** 2 / beta, n - 0.5 * beta)
        if reduction == "mean":
            loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
        elif reduction == "sum":
            loss = loss.sum()
        return loss

    def giou_loss(boxes1, boxes2, reduction="none", eps=1e-7):
        x1, y1, x2, y2 = boxes1.unbind(dim=-1)
        x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
        assert (x2 >= x1).all(), "bad box: x1 larger than x2"
        assert (y2 >= y1).all(), "bad box: y1 larger than y2"
        xkis1 = torch.max(x1, x1g)
        ykis1 = torch.max(y1, y1g)
        xkis2 = torch.min(x2, x2g)
        ykis2 = torch.min(y2, y2g)
        intsctk = torch.zeros_like(x1)
        mask = (ykis2 > ykis1) & (xkis2 > xkis1)
        intsctk[mask] = (xkis2[mask] - xkis1[mask]) * (ykis2[mask] - ykis1[mask])
        unionk = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - intsctk
        iouk = intsctk / (unionk + eps)
        xc1 = torch.min(x1, x1g)
        yc1 = torch.min(y1, y1g)
        xc2 = torch.max(x2, x2g)
        yc2 = torch.max(y2, y2g)
        area_c = (xc2 - xc1) * (yc2 - yc1)
        miouk = iouk - ((area_c - unionk) / (area_c + eps))
        loss = 1 - miouk
        if reduction == "mean":
            loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
        elif reduction == "sum":
            loss = loss.sum()
        return loss

    m.smooth_l1_loss = smooth_l1_loss
    m.giou_loss = giou_loss
    wi = sys.modules.get("fvcore.nn.weight_init") or importlib.import_module("fvcore.nn.weight_init")
    m.weight_init = wi


def _fvcore(m):
    m.__version__ = "0.1.5"


def _fv_transform(m):
    """fvcore.transforms.transform (third-party, not under /root/reference): the base-class behaviour the reference's
    ResizeTransform relies on, from fvcore's published source -- `_set_attributes` stores the constructor arguments,
    `apply_box` maps the 4 corners through `apply_coords` and takes their min/max."""
    import numpy as np

    class Transform:
        def _set_attributes(self, params=None):
            if params:
                for k, v in params.items():
                    if k != "self" and not k.startswith("_"):
                        setattr(self, k, v)

        def apply_box(self, box):
            idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
            coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
            coords = self.apply_coords(coords).reshape((-1, 4, 2))
            return np.concatenate((coords.min(axis=1), coords.max(axis=1)), axis=1)

        @classmethod
        def register_type(cls, *a, **k):
            return (lambda f: f) if not (len(a) == 2 and callable(a[1])) else None

    class NoOpTransform(Transform):
        def apply_image(self, img):
            return img

        def apply_coords(self, coords):
            return coords

    m.Transform = Transform
    m.NoOpTransform = NoOpTransform


def _termcolor(m):
    m.colored = lambda s, *a, **k: s


def _torchvision(m):
    m.__version__ = "0.6.0"


def _tv_ops(m):
    from oracle import ops as oops
    m.nms = oops.nms
    m.box_iou = oops.box_iou
    m.roi_align = None  # never reached: version gate routes to detectron2._C
    boxes = importlib.import_module("torchvision.ops.boxes")
    m.boxes = boxes


def _tv_ops_boxes(m):
    from oracle import ops as oops
    m.nms = oops.nms
    m.batched_nms = oops.batched_nms
    m.box_iou = oops.box_iou


def _pycocotools_coco(m):
    """pycocotools.coco.COCO (third-party, absent here; setup.py:186-188 asks pycocotools>=2.0.1): the index / query
    subset the reference's pseudo-label tool uses (tools/create_coco_dataset_from_dets_all.py:14-54, 129-263), restated
    from the published 2.0 source.  "parity unpinned" for these methods: nothing under /root/reference tests them."""
    import copy
    import itertools
    import json
    from collections import defaultdict

    def _is_array_like(obj):
        return hasattr(obj, "__iter__") and hasattr(obj, "__len__")

    class COCO:
        def __init__(self, annotation_file=None):
            self.dataset, self.anns, self.cats, self.imgs = dict(), dict(), dict(), dict()
            self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
            if annotation_file is not None:
                with open(annotation_file, "r") as f:
                    dataset = json.load(f)
                assert type(dataset) == dict
                self.dataset = dataset
                self.createIndex()

        def createIndex(self):
            anns, cats, imgs = {}, {}, {}
            imgToAnns, catToImgs = defaultdict(list), defaultdict(list)
            if "annotations" in self.dataset:
                for ann in self.dataset["annotations"]:
                    imgToAnns[ann["image_id"]].append(ann)
                    anns[ann["id"]] = ann
            if "images" in self.dataset:
                for img in self.dataset["images"]:
                    imgs[img["id"]] = img
            if "categories" in self.dataset:
                for cat in self.dataset["categories"]:
                    cats[cat["id"]] = cat
            if "annotations" in self.dataset and "categories" in self.dataset:
                for ann in self.dataset["annotations"]:
                    catToImgs[ann["category_id"]].append(ann["image_id"])
            self.anns, self.imgToAnns, self.catToImgs, self.imgs, self.cats = anns, imgToAnns, catToImgs, imgs, cats

        def getAnnIds(self, imgIds=[], catIds=[], areaRng=[], iscrowd=None):
            imgIds = imgIds if _is_array_like(imgIds) else [imgIds]
            catIds = catIds if _is_array_like(catIds) else [catIds]
            if len(imgIds) == len(catIds) == len(areaRng) == 0:
                anns = self.dataset["annotations"]
            else:
                if not len(imgIds) == 0:
                    lists = [self.imgToAnns[i] for i in imgIds if i in self.imgToAnns]
                    anns = list(itertools.chain.from_iterable(lists))
                else:
                    anns = self.dataset["annotations"]
                anns = anns if len(catIds) == 0 else [a for a in anns if a["category_id"] in catIds]
                anns = anns if len(areaRng) == 0 else [a for a in anns if a["area"] > areaRng[0] and a["area"] < areaRng[1]]
            if iscrowd is not None:
                return [a["id"] for a in anns if a["iscrowd"] == iscrowd]
            return [a["id"] for a in anns]

        def getImgIds(self, imgIds=[], catIds=[]):
            imgIds = imgIds if _is_array_like(imgIds) else [imgIds]
            catIds = catIds if _is_array_like(catIds) else [catIds]
            if len(imgIds) == len(catIds) == 0:
                ids = self.imgs.keys()
            else:
                ids = set(imgIds)
                for i, catId in enumerate(catIds):
                    if i == 0 and len(ids) == 0:
                        ids = set(self.catToImgs[catId])
                    else:
                        ids &= set(self.catToImgs[catId])
            return list(ids)

        def loadAnns(self, ids=[]):
            if _is_array_like(ids):
                return [self.anns[i] for i in ids]
            return [self.anns[ids]]

        def loadImgs(self, ids=[]):
            if _is_array_like(ids):
                return [self.imgs[i] for i in ids]
            return [self.imgs[ids]]

        def loadRes(self, resFile):
            res = COCO()
            res.dataset["images"] = [img for img in self.dataset["images"]]
            if isinstance(resFile, str):
                with open(resFile) as f:
                    anns = json.load(f)
            else:
                anns = resFile
            assert type(anns) == list, "results in not an array of objects"
            annsImgIds = [ann["image_id"] for ann in anns]
            assert set(annsImgIds) == (set(annsImgIds) & set(self.getImgIds()))
            assert "bbox" in anns[0] and not anns[0]["bbox"] == []
            res.dataset["categories"] = copy.deepcopy(self.dataset["categories"])
            for id, ann in enumerate(anns):
                bb = ann["bbox"]
                x1, x2, y1, y2 = [bb[0], bb[0] + bb[2], bb[1], bb[1] + bb[3]]
                if "segmentation" not in ann:
                    ann["segmentation"] = [[x1, y1, x1, y2, x2, y2, x2, y1]]
                ann["area"] = bb[2] * bb[3]
                ann["id"] = id + 1
                ann["iscrowd"] = 0
            res.dataset["annotations"] = anns
            res.createIndex()
            return res

    m.COCO = COCO


_REAL = {
    "pycocotools.coco": _pycocotools_coco,
    "fvcore": _fvcore,
    "fvcore.common.registry": _fv_registry,
    "fvcore.common.config": _fv_config,
    "fvcore.nn.weight_init": _fv_weight_init,
    "fvcore.nn": _fv_nn,
    "fvcore.transforms.transform": _fv_transform,
    "termcolor": _termcolor,
    "torchvision": _torchvision,
    "torchvision.ops": _tv_ops,
    "torchvision.ops.boxes": _tv_ops_boxes,
}

_installed = False


def install():
    """Idempotent.  After this, `import detectron2`, `import lvc`, `import tools....` work."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (this shim only works in the build container)")
    if _REPO not in sys.path:
        sys.path.insert(0, _REPO)
    sys.meta_path.insert(0, _Finder())
    # Pillow >= 10 dropped Image.LINEAR (used at detectron2/data/transforms/transform.py:42)
    try:
        from PIL import Image
        if not hasattr(Image, "LINEAR"):
            Image.LINEAR = Image.BILINEAR
    except ImportError:
        pass
    # the reference's own native extension, compiled from its sources (oracle/build_ref.py)
    so = os.path.join(_HERE, "_ref", "_C.so")
    if not os.path.exists(so):
        from oracle import build_ref
        build_ref.build()
    import torch  # noqa: F401  (libtorch must be loaded before the extension is dlopen'ed)
    spec = importlib.util.spec_from_file_location("_C", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["detectron2._C"] = mod
    sys.path.insert(0, REF_ROOT)
    _installed = True


def ref_C():
    """The compiled reference extension alone (works on the GPU box too: prebuilt .so)."""
    so = os.path.join(_HERE, "_ref", "_C.so")
    if not os.path.exists(so):
        return None
    import torch  # noqa: F401
    spec = importlib.util.spec_from_file_location("_C", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
