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
            flag = torch.tensor([float(word & 2)], device=device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            if float(flag.item()) > 0:
                word |= 2
        if word & 2:
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
