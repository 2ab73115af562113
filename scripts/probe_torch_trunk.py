"""Yardstick, not a product path: the same R50-FPN trunk (stem, res2-5, FPN incl. p6) evaluated by PyTorch-ROCm's own convolutions
(MIOpen through F.conv2d: what the reference's modules would run on this GPU) in fp32 and in fp16, against this repo's backbone on
the same batch of 8 x 800 x 1333; outputs compared so that the two do the same work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn

dev = torch.device("cuda:0")
model = build_model(base_rcnn_fpn(depth=50, num_classes=80, device="cuda:0")).eval()
syn.conditioned_r50_fpn_(model)
sd = {k: v.detach() for k, v in model.state_dict().items()}
B = 8
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(B)]


def fold(prefix, dtype):
    """conv weight with the FrozenBN affine folded in -> (weight, bias)"""
    w = sd[prefix + ".weight"].float()
    if prefix + ".norm.weight" in sd:
        s = sd[prefix + ".norm.weight"] * (sd[prefix + ".norm.running_var"] + 1e-5).rsqrt()
        b = sd[prefix + ".norm.bias"] - sd[prefix + ".norm.running_mean"] * s
        w = w * s[:, None, None, None]
    else:
        b = sd.get(prefix + ".bias", torch.zeros(w.shape[0], device=dev))
    return w.to(dtype).contiguous(memory_format=torch.channels_last), b.to(dtype)


def build(dtype):
    P = {}
    for k in sd:
        if k.endswith(".weight") and sd[k].dim() == 4 and k.startswith("backbone."):
            P[k[: -len(".weight")]] = fold(k[: -len(".weight")], dtype)
    return P


def trunk(x, P):
    bu = "backbone.bottom_up."
    w, b = P[bu + "stem.conv1"]
    x = F.max_pool2d(F.relu_(F.conv2d(x, w, b, stride=2, padding=3)), 3, 2, 1)
    outs = {}
    for stage, nblk in (("res2", 3), ("res3", 4), ("res4", 6), ("res5", 3)):
        for i in range(nblk):
            p = bu + "%s.%d." % (stage, i)
            stride = 2 if (i == 0 and stage != "res2") else 1
            sc = x
            if p + "shortcut" in P:
                w, b = P[p + "shortcut"]
                sc = F.conv2d(x, w, b, stride=stride)
            w, b = P[p + "conv1"]
            y = F.relu_(F.conv2d(x, w, b, stride=stride))        # STRIDE_IN_1X1
            w, b = P[p + "conv2"]
            y = F.relu_(F.conv2d(y, w, b, padding=1))
            w, b = P[p + "conv3"]
            x = F.relu_(F.conv2d(y, w, b) + sc)
        outs[stage] = x
    res = {}
    prev = None
    for lvl, stage in ((5, "res5"), (4, "res4"), (3, "res3"), (2, "res2")):
        w, b = P["backbone.fpn_lateral%d" % lvl]
        lat = F.conv2d(outs[stage], w, b)
        if prev is not None:
            lat = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = lat
        w, b = P["backbone.fpn_output%d" % lvl]
        res["p%d" % lvl] = F.conv2d(lat, w, b, padding=1)
    res["p6"] = F.max_pool2d(res["p5"], 1, 2, 0)
    return res


def timeit(f, n):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    images = model.preprocess_image(batch)
    x = images.tensor                                   # [8, 3, 800, 1344] NCHW-shaped view, the normalised, padded batch
    mine = model.backbone(x)
    t_mine = timeit(lambda: model.backbone(x), 10)
    print("lvc_amd backbone + FPN (fp32-accurate fp16-split kernels): %.2f ms per batch of %d" % (t_mine, B), flush=True)
    kinds = ((torch.float32, "fp32"), (torch.float16, "fp16 (not the reference's precision)"))
    if os.environ.get("LVC_TRUNK_YARDSTICK_FP32_ONLY") == "1":
        kinds = kinds[:1]
    for dtype, name in kinds:
        P = build(dtype)
        xin = x[:, :3].to(dtype).contiguous(memory_format=torch.channels_last)
        t0 = time.perf_counter()
        ref = trunk(xin, P)
        torch.cuda.synchronize()
        print("PyTorch-ROCm F.conv2d %s: first call %.1f s" % (name, time.perf_counter() - t0), flush=True)
        t = timeit(lambda: trunk(xin, P), 5)
        err = max(float((ref[k].float() - mine[k]).abs().max() / mine[k].abs().max()) for k in ("p2", "p3", "p4", "p5"))
        print("PyTorch-ROCm F.conv2d %s, channels_last: %.2f ms per batch  (%.2f x lvc_amd; max |diff| / scale vs lvc_amd %.1e)" % (name, t, t / t_mine, err), flush=True)
