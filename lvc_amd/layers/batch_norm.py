"""FrozenBatchNorm2d: buffer-only per-channel affine (reference detectron2/layers/batch_norm.py:14-125).

y = x * (weight * rsqrt(running_var + eps)) + (bias - running_mean * weight * rsqrt(running_var + eps)).
Every shipped config uses it for the whole backbone (`RESNETS.NORM = FrozenBN`, reference
detectron2/config/defaults.py:471), in training too, so `lvc_amd.layers.Conv2d` folds it into the
epilogue of the conv kernel; this module only owns the four buffers (state_dict names `weight`,
`bias`, `running_mean`, `running_var`) and is never launched on its own inside the trunk.
"""
import torch
from torch import nn


class FrozenBatchNorm2d(nn.Module):
    _version = 3

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def affine(self):
        """(scale, shift) in fp32."""
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        # stand-alone use (outside Conv2d) is not on the hot path; plain broadcast arithmetic.
        scale, shift = self.affine()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            # pre-v2 checkpoints carry no running stats (reference batch_norm.py:67-89)
            if prefix + "running_mean" not in state_dict:
                state_dict[prefix + "running_mean"] = torch.zeros_like(self.running_mean)
            if prefix + "running_var" not in state_dict:
                state_dict[prefix + "running_var"] = torch.ones_like(self.running_var)
        if version is not None and version < 3:
            state_dict[prefix + "running_var"] -= self.eps
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def __repr__(self):
        return "FrozenBatchNorm2d(num_features={}, eps={})".format(self.num_features, self.eps)


def get_norm(norm, out_channels):
    """reference batch_norm.py:127-150.  Only "" and "FrozenBN" are on the path of the shipped configs."""
    if isinstance(norm, str):
        if len(norm) == 0:
            return None
        if norm != "FrozenBN":
            raise NotImplementedError(
                "lvc_amd implements NORM='' and 'FrozenBN' (every shipped config); got '{}'".format(norm))
        return FrozenBatchNorm2d(out_channels)
    return norm(out_channels)
