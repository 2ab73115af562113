"""Seeded synthetic inputs and random-init weights for the hot path.

There is no network in the build/bench environment, so benchmarks and parity tests run on
synthetic 3x800x1333 images and random weights of the R50/R101-FPN architecture.  Default
random init is degenerate for a detector (activations blow up, every
proposal clips to the full image, 1 detection survives), so the recipe here produces a
*conditioned* model: order-independent per-tensor seeded weights + FrozenBN statistics
calibrated once on the synthetic image (committed as a small fixture, see
tests/golden/README.md) so that ~1000 proposals and ~100 detections flow through ROIAlign
and both NMS stages.

Everything is generated with CPU torch generators keyed by the tensor *name*, so the same
state_dict is reproduced on any host without the reference.
"""
import math
import zlib

import torch
import torch.nn.functional as F


def synthetic_image(seed=1, height=800, width=1333):
    """Smooth blobs + noise in [0,255], float32 CHW (BGR by convention)."""
    g = torch.Generator().manual_seed(seed)
    lh, lw = max(2, height // 32), max(2, width // 32)
    low = torch.rand(1, 3, lh, lw, generator=g) * 255.0
    up = F.interpolate(low, size=(height, width), mode="bilinear", align_corners=False)[0]
    noise = torch.rand(3, height, width, generator=g) * 255.0
    return (0.8 * up + 0.2 * noise).contiguous()


def _gen(name, seed):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)


# (substring of key, std) overrides applied after the generic rule; chosen so that the synthetic
# image yields ~1000 proposals spread over levels and ~100 detections (see tests/golden/README.md)
_HEAD_STD = (
    ("rpn_head.objectness_logits.weight", 0.03),
    ("rpn_head.anchor_deltas.weight", 0.015),
    ("box_predictor.cls_score.weight", 0.008),
    ("box_predictor.bbox_pred.weight", 0.004),
)


def conditioned_state_dict(template, seed=0, bn_calibration=None):
    """Return a new state_dict with the keys/shapes of `template` (a module.state_dict()).

    conv/linear weights ~ N(0, 2/fan_in) (He), biases ~ N(0, 0.01), FrozenBN weight=1, bias=0,
    running_mean/var from `bn_calibration` (dict name->tensor) when given, else (0, 1).
    `cell_anchors` buffers are left as they are.
    """
    out = {}
    for name, t in template.items():
        if "cell_anchors" in name:
            out[name] = t.clone()
            continue
        g = _gen(name, seed)
        leaf = name.rsplit(".", 1)[-1]
        if ".norm." in name or name.endswith("running_mean") or name.endswith("running_var"):
            if bn_calibration is not None and name in bn_calibration:
                out[name] = bn_calibration[name].clone().to(t.dtype)
            elif leaf == "weight":
                out[name] = torch.ones_like(t)
            elif leaf == "running_var":
                out[name] = torch.ones_like(t)
            else:
                out[name] = torch.zeros_like(t)
            continue
        if leaf == "bias":
            out[name] = torch.randn(t.shape, generator=g) * 0.01
            continue
        fan_in = t[0].numel() if t.dim() > 1 else t.numel()
        std = math.sqrt(2.0 / fan_in)
        for pat, s in _HEAD_STD:
            if pat in name:
                std = s
        out[name] = torch.randn(t.shape, generator=g) * std
    return out


def calibrate_frozen_bn_(model, run, bn_type):
    """Sequentially set every `bn_type` module's running stats to the statistics of ITS input
    under `run()` (one forward per call order is not needed: a single pass with forward
    pre-hooks that update the module before it executes).  Returns {state_dict key: tensor}."""
    calib = {}
    handles = []
    names = {m: n for n, m in model.named_modules()}

    def pre_hook(mod, inp):
        x = inp[0]
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        mod.running_mean.copy_(mean)
        mod.running_var.copy_(var.clamp_min(1e-6))
        calib[names[mod] + ".running_mean"] = mod.running_mean.clone()
        calib[names[mod] + ".running_var"] = mod.running_var.clone()

    for m in model.modules():
        if isinstance(m, bn_type):
            handles.append(m.register_forward_pre_hook(pre_hook))
    with torch.no_grad():
        run()
    for h in handles:
        h.remove()
    return calib
