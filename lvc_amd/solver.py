"""Power-of-two loss scaling for the backward pass on the two-way fp16 data-gradient kernels.

The reference trains in plain fp32 (torch SGD, detectron2/solver/build.py); nothing here changes the optimizer or the
numbers it sees: the loss is multiplied by S = 2^k before `backward()` and every parameter gradient by 1/S after it, both
exact in fp32.  What the scale buys is the operand range of the DATA-gradient convolutions: scaled by S the upstream
gradients fall inside fp16's range, so those convolutions can run on the two-way fp16 split kernels (3 fp16 MFMAs per
fp32-accurate product, as the forward does) instead of the three-way bf16 split (6 bf16 MFMAs), ~1.5x faster
(`scripts/probe_train_step.py --loss-scale`).  The weight-gradient kernel is fp32 MFMA either way.

An operand beyond 65504 raises bit 1 of the conv error word (lvc_amd/csrc/conv3x3_halo_h2.hip); `step()` then skips the
optimizer step, halves the scale and clears the word -- the AMP GradScaler protocol, with the overflow detected by the
kernels themselves instead of an inf/nan scan of the gradients.
"""
import contextlib

import torch

from . import kernels as K


class LossScaler:
    def __init__(self, init_scale=2.0 ** 10, growth_interval=2000, max_scale=2.0 ** 20):
        assert init_scale > 0 and float(init_scale) == 2.0 ** round(torch.log2(torch.tensor(float(init_scale))).item()), \
            "the scale must be a power of two (the scaling has to be exact)"
        self.scale_value = float(init_scale)
        self.growth_interval = growth_interval
        self.max_scale = max_scale
        self._good_steps = 0
        self.skipped_steps = 0

    def scale(self, loss):
        return loss * self.scale_value

    @contextlib.contextmanager
    def backward_pass(self):
        """Run `backward()` inside: data gradients on the two-way fp16 kernels."""
        prev = K.DGRAD_SPLIT
        K.DGRAD_SPLIT = "f16x2"
        try:
            yield
        finally:
            K.DGRAD_SPLIT = prev

    def backward(self, loss):
        with self.backward_pass():
            self.scale(loss).backward()

    def step(self, optimizer, params=None, device=None):
        """Unscale the gradients, take the optimizer step unless a kernel saw an out-of-range operand; returns True when
        the step was taken.  Reads the conv error word (one host sync, as an AMP step's inf check)."""
        if params is None:
            params = [p for g in optimizer.param_groups for p in g["params"]]
        params = [p for p in params if p.grad is not None]
        device = device or (params[0].device if params else None)
        word = K.conv_error_word(device) if device is not None else 0
        if word & 1:
            K.check_conv_error_word(device)        # stream-K timeout: not a scaling matter
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # data parallel: every rank must take the same decision (one rank's overflow skips the step everywhere)
            flag = torch.tensor([float(word & 6)], device=device)     # bit 1: finite operand out of range, bit 2: non-finite
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            if float(flag.item()) > 0:
                word |= 2
        if word & 6:
            K.clear_conv_error_word(device)
            for p in params:
                p.grad = None
            self.scale_value = max(self.scale_value / 2.0, 1.0)
            self._good_steps = 0
            self.skipped_steps += 1
            return False
        inv = 1.0 / self.scale_value
        if params:
            torch._foreach_mul_([p.grad for p in params], inv)
        optimizer.step()
        self._good_steps += 1
        if self._good_steps >= self.growth_interval and self.scale_value < self.max_scale:
            self.scale_value *= 2.0
            self._good_steps = 0
        return True


# ---- optimizer / learning-rate schedule (host side; reference detectron2/solver/build.py, lr_scheduler.py) -----------
_NORM_TYPES = (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm, torch.nn.modules.instancenorm._InstanceNorm,
               torch.nn.LayerNorm, torch.nn.LocalResponseNorm)


def parameter_groups(cfg, model):
    """One group per trainable parameter, as the reference builds them (detectron2/solver/build.py:110-131): weights
    of normalisation modules decay by WEIGHT_DECAY_NORM, parameters NAMED "bias" train at BASE_LR * BIAS_LR_FACTOR and
    decay by WEIGHT_DECAY_BIAS, everything else at BASE_LR / WEIGHT_DECAY.  A parameter shared by two modules is
    listed once, under the first module that owns it."""
    s = cfg.SOLVER
    groups, seen = [], set()
    for module in model.modules():
        is_norm = isinstance(module, _NORM_TYPES)
        for name, p in module.named_parameters(recurse=False):
            if not p.requires_grad or id(p) in seen:
                continue
            seen.add(id(p))
            lr, wd = s.BASE_LR, s.WEIGHT_DECAY
            if is_norm:
                wd = s.WEIGHT_DECAY_NORM
            elif name == "bias":
                lr, wd = s.BASE_LR * s.BIAS_LR_FACTOR, s.WEIGHT_DECAY_BIAS
            groups.append({"params": [p], "lr": lr, "weight_decay": wd})
    return groups


class _ClippedSGD(torch.optim.SGD):
    """SGD whose step first clips every parameter's gradient on its own (the reference clips per parameter, not over
    the whole model: detectron2/solver/build.py:46-51)."""
    clip_type, clip_value, norm_type = "value", 1.0, 2.0

    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if self.clip_type == "value":
                    p.grad.clamp_(-self.clip_value, self.clip_value)
                else:
                    torch.nn.utils.clip_grad_norm_(p, self.clip_value, self.norm_type)
        return super().step(closure)


def build_optimizer(cfg, model):
    """torch SGD over `parameter_groups` (detectron2/solver/build.py:93-136); SOLVER.CLIP_GRADIENTS.ENABLED adds the
    per-parameter clip in front of every step (ibid. 59-91)."""
    s = cfg.SOLVER
    clip = s.CLIP_GRADIENTS
    if clip.ENABLED:
        if clip.CLIP_TYPE not in ("value", "norm"):
            raise ValueError("'{}' is not a valid GradientClipType".format(clip.CLIP_TYPE))
        opt = _ClippedSGD(parameter_groups(cfg, model), s.BASE_LR, momentum=s.MOMENTUM, nesterov=s.NESTEROV)
        opt.clip_type, opt.clip_value, opt.norm_type = clip.CLIP_TYPE, float(clip.CLIP_VALUE), float(clip.NORM_TYPE)
        return opt
    return torch.optim.SGD(parameter_groups(cfg, model), s.BASE_LR, momentum=s.MOMENTUM, nesterov=s.NESTEROV)


def warmup_factor_at_iter(method, it, warmup_iters, warmup_factor):
    """detectron2/solver/lr_scheduler.py:90-116: 1 after the warmup; during it `warmup_factor` ("constant") or the
    line from `warmup_factor` at iteration 0 to 1 at `warmup_iters` ("linear")."""
    if it >= warmup_iters:
        return 1.0
    if method == "constant":
        return warmup_factor
    if method == "linear":
        a = it / warmup_iters
        return warmup_factor * (1 - a) + a
    raise ValueError("Unknown warmup method: {}".format(method))


class _WarmupLR(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, warmup_factor, warmup_iters, warmup_method, last_epoch=-1):
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def _decay(self, it):
        raise NotImplementedError

    def get_lr(self):
        f = warmup_factor_at_iter(self.warmup_method, self.last_epoch, self.warmup_iters, self.warmup_factor)
        d = self._decay(self.last_epoch)
        return [base * f * d for base in self.base_lrs]

    _compute_values = get_lr


class WarmupMultiStepLR(_WarmupLR):
    """lr = base * warmup * gamma^(number of milestones <= iteration) (detectron2/solver/lr_scheduler.py:15-48)."""

    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=0.001, warmup_iters=1000,
                 warmup_method="linear", last_epoch=-1):
        milestones = list(milestones)
        if milestones != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        self.milestones, self.gamma = milestones, gamma
        super().__init__(optimizer, warmup_factor, warmup_iters, warmup_method, last_epoch)

    def _decay(self, it):
        return self.gamma ** sum(1 for m in self.milestones if m <= it)


class WarmupCosineLR(_WarmupLR):
    """lr = base * warmup * (1 + cos(pi * iteration / max_iters)) / 2 (detectron2/solver/lr_scheduler.py:51-87)."""

    def __init__(self, optimizer, max_iters, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear",
                 last_epoch=-1):
        self.max_iters = max_iters
        super().__init__(optimizer, warmup_factor, warmup_iters, warmup_method, last_epoch)

    def _decay(self, it):
        import math
        return 0.5 * (1.0 + math.cos(math.pi * it / self.max_iters))


def build_lr_scheduler(cfg, optimizer):
    """detectron2/solver/build.py:139-165."""
    s = cfg.SOLVER
    kw = dict(warmup_factor=s.WARMUP_FACTOR, warmup_iters=s.WARMUP_ITERS, warmup_method=s.WARMUP_METHOD)
    if s.LR_SCHEDULER_NAME == "WarmupMultiStepLR":
        return WarmupMultiStepLR(optimizer, s.STEPS, s.GAMMA, **kw)
    if s.LR_SCHEDULER_NAME == "WarmupCosineLR":
        return WarmupCosineLR(optimizer, s.MAX_ITER, **kw)
    raise ValueError("Unknown LR scheduler: {}".format(s.LR_SCHEDULER_NAME))
