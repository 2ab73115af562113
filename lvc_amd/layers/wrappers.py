"""Conv2d / Linear parameter owners whose forward is the gfx950 implicit-GEMM kernel.

Mirrors reference detectron2/layers/wrappers.py:41-99 (Conv2d with optional `norm` and `activation`
sub-objects; parameter names `weight`, `bias`, `norm.*`) but runs conv + FrozenBN + bias + ReLU
(+ residual / upsample-add) as ONE launch of csrc/conv_igemm.hip.  Weights stay in the reference's
OIHW layout in the state_dict; the packed KRSC copy is rebuilt lazily whenever a parameter changes.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import kernels as K
from .batch_norm import FrozenBatchNorm2d
from .layout import require_device, to_nchw_view, to_nhwc


def cat(tensors, dim=0):
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


def nonzero_tuple(x):
    return x.nonzero(as_tuple=True) if x.dim() > 0 else x.unsqueeze(0).nonzero(as_tuple=True)


class _PackedCache:
    """Re-pack when any source tensor was modified in place or replaced (optimizer step, load_state_dict, .to())."""

    def __init__(self):
        self.key = None
        self.value = None

    def get(self, tensors, build):
        # (address, version): a parameter that moved to another device has another address; `str(t.device)` cost a microsecond per
        # tensor and layer, ~0.5 ms of host time per step
        key = tuple((t.data_ptr(), t._version) for t in tensors if t is not None)
        if key != self.key:
            self.value = build()
            self.key = key
        return self.value


class Conv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, norm=None,
                 activation=None, dilation=1, groups=1):
        super().__init__()
        if dilation != 1 or groups != 1:
            raise NotImplementedError("dilation/groups are not used by the shipped configs")
        ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        assert ks[0] == ks[1]
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = ks, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, ks[0], ks[1]))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.norm = norm
        self.activation = activation
        if activation is not None and activation not in (F.relu, F.relu_):
            raise NotImplementedError("only ReLU is fused (the only activation on the path)")
        self._cache = _PackedCache()
        self._cache_dgrad = _PackedCache()
        self._cache_affine = _PackedCache()
        self._range_state = {"tier": 0}      # this layer's range tier survives re-packs (kernels.check_conv_error_word)

    def _affine(self):
        """FrozenBN fold / bias of the epilogue; rebuilt only when one of ITS tensors changes (not on every
        optimizer step of the weights)."""
        bn = None
        srcs = [self.bias]
        if self.norm is not None:
            assert isinstance(self.norm, FrozenBatchNorm2d)
            bn = (self.norm.weight, self.norm.bias, self.norm.running_mean, self.norm.running_var)
            srcs += list(bn)
        return self._cache_affine.get(srcs, lambda: K.conv_affine(self.bias, bn, self.norm.eps if bn else 1e-5))

    def packed(self):
        aff = self._affine()
        srcs = [self.weight, aff[0], aff[1]]
        stem = self.in_channels == 3
        pc = self._cache.get(srcs, lambda: K.pack_conv(self.weight, stride=self.stride, pad=self.padding, stem=stem,
                                                       affine=aff))
        # precision policy of the 3x3 fp16-split kernel (kernels.HALO_S1): a layer whose output feeds discrete decisions sets
        # `two_acc` and keeps the main + cross accumulator form (the RPN head: objectness / deltas -> top-k, NMS)
        pc.two_acc = bool(getattr(self, "two_acc", False))
        pc.state = self._range_state
        return pc

    def _grad_lands_in_weight(self):
        """True inside a pass that will run the weight's AccumulateGrad node -- `loss.backward()` -- i.e. when the deferred weight gradient
        may be written to `weight.grad` in its place; False under `torch.autograd.grad(...)` / `backward(inputs=...)`, where the node must
        hand its gradient back to the engine."""
        w = self.weight
        if K._wgrad_sink(w) is None:
            # no registered taker for the deferred gradient (GradientBuckets registers one per parameter): anything that listens to the
            # autograd path of this parameter -- tensor hooks, or a data-parallel wrapper's hooks on the AccumulateGrad node
            # (torch DistributedDataParallel, which the reference's DefaultTrainer uses) -- must see the gradient arrive there
            if w._backward_hooks or getattr(w, "_post_accumulate_grad_hooks", None):
                return False
            if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                return False
        node = self.__dict__.get("_acc_node")
        if node is None or node[0] is not self.weight:
            with torch.enable_grad():
                node = self.__dict__["_acc_node"] = (self.weight, self.weight.view_as(self.weight).grad_fn.next_functions[0][0])
        try:
            return bool(torch._C._will_engine_execute_node(node[1]))
        except RuntimeError:       # "a leaf node was passed ... running autograd.grad()": the weight is one of its inputs
            return False

    def _stale(self, cache, tensors):
        return cache.key != tuple((t.data_ptr(), t._version) for t in tensors if t is not None)

    @staticmethod
    def prepack(convs, dgrad=None, holder=None):
        """Re-pack the stale operands of `convs` (trainable layers after an optimizer step) in grouped launches
        (`kernels.PrepackPlan`): what `packed()` / `packed_dgrad()` would build one layer at a time.  dgrad: the layers whose
        data-gradient operand is needed too (default: all of them).  holder: a dict that keeps the plan between steps -- the same
        set of stale layers is then re-packed into the same buffers (no per-layer host work)."""
        jobs, targets, sig = [], [], []
        for conv in convs:
            if conv.in_channels == 3 or conv.in_channels % 32:
                continue
            aff = conv._affine()
            srcs = [conv.weight, aff[0], aff[1]]
            if conv._stale(conv._cache, srcs):
                jobs.append({"weight": conv.weight, "kind": "fwd", "stride": conv.stride, "pad": conv.padding, "affine": aff,
                             "two_acc": bool(getattr(conv, "two_acc", False)), "tier": conv._range_state.get("tier", 0)})
                targets.append((conv._cache, srcs))
                sig.append((id(conv), 0, conv.weight.data_ptr(), id(aff[0]), jobs[-1]["two_acc"], jobs[-1]["tier"]))
            if dgrad is None or conv in dgrad:
                scale = aff[0] if conv.norm is not None else None
                srcs = [conv.weight, scale]
                if conv._stale(conv._cache_dgrad, srcs):
                    jobs.append({"weight": conv.weight, "kind": "dgrad", "stride": conv.stride, "pad": conv.padding, "scale": scale})
                    targets.append((conv._cache_dgrad, srcs))
                    sig.append((id(conv), 1, conv.weight.data_ptr(), id(scale)))
        if not jobs:
            return
        sig = (tuple(sig), K.DGRAD_SPLIT, K.HALO_S1, K.PW_S1, K.CONV_ENGINE, K.CONV_SPLIT)
        plan = holder.get("plan") if holder is not None else None
        if plan is not None and holder.get("sig") == sig and plan.reusable:
            plan.relaunch()
        else:
            plan = K.PrepackPlan(jobs)
            if holder is not None:
                holder["plan"], holder["sig"] = plan, sig
        for (cache, srcs), pc in zip(targets, plan.packed):
            cache.value = pc
            cache.key = tuple((t.data_ptr(), t._version) for t in srcs if t is not None)

    def packed_dgrad(self):
        """Packed weights of the data gradient (flipped, transposed, times the FrozenBN scale)."""
        scale = self._affine()[0] if self.norm is not None else None
        return self._cache_dgrad.get([self.weight, scale], lambda: K.pack_conv_dgrad(self.weight, scale, self.padding))

    def forward_nhwc(self, x, residual=None, res_mode=0, relu=None):
        """x: [N,H,W,C] contiguous.  relu=None -> this layer's own activation."""
        if relu is None:
            relu = self.activation is not None
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad
                                        or (self.bias is not None and self.bias.requires_grad)
                                        or (residual is not None and residual.requires_grad)):
            return _ConvFn.apply(x, self.weight, self.bias, residual, self, res_mode if residual is not None else 0, relu)
        return K.conv2d_nhwc(x, self.packed(), relu=relu, residual=residual, res_mode=res_mode)

    def forward(self, x):
        require_device(x, "Conv2d")
        xh = to_nhwc(x)
        if self.in_channels == 3 and xh.shape[-1] == 3:
            xh = F.pad(xh, (0, 1))  # generic caller: pad RGB to the 4-slot pixel the stem kernel reads
        return to_nchw_view(self.forward_nhwc(xh))

    def extra_repr(self):
        return "{}, {}, kernel_size={}, stride={}, padding={}".format(
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding)


class _ConvFn(torch.autograd.Function):
    """y = act(conv(x, W) * bn_scale + shift (+ residual | + up2(residual))) as the ONE fused forward launch; backward:
    ReLU mask (lvc_relu_backward), residual gradient (identity, or the 2x2 down-sum of the FPN top-down add), weight
    gradient (lvc_conv_wgrad_nhwc, FrozenBN scale folded in), bias gradient (lvc_colsum_atomic) and data gradient (the
    forward kernels on the flipped / transposed weights, `Conv2d.packed_dgrad`).  What ATen's conv2d / relu / add /
    interpolate backward do for BottleneckBlock (reference detectron2/modeling/backbone/resnet.py:195-211), FPN
    (fpn.py:109-144) and StandardRPNHead (rpn.py:120-139) when their parameters train."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, conv, res_mode, relu):
        if conv.in_channels == 3:
            raise NotImplementedError("training the 7x7 stem is not implemented (every shipped config has FREEZE_AT >= 1)")
        if K._WGRAD_ARMED[0] and not K.in_backward():
            # a backward() that raised left queued weight gradients behind.  (A forward INSIDE a live backward -- checkpoint
            # recomputation, a hook -- keeps the queue: those gradients are still to be delivered.)
            K.reset_wgrad_queue()
        if weight.requires_grad:
            K.note_wgrad_use(weight)
        y = K.conv2d_nhwc(x, conv.packed(), relu=relu, residual=residual, res_mode=res_mode)
        ctx.save_for_backward(x, y if relu else None)
        ctx.conv, ctx.res_mode, ctx.relu = conv, res_mode, relu
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        conv = ctx.conv
        g = K.relu_backward(dy, y) if ctx.relu else dy.contiguous()
        need_x, need_w, need_b, need_r = ctx.needs_input_grad[:4]
        dx = dw = db = dres = None
        if need_r:
            dres = g if ctx.res_mode == 1 else K.downsum2x2(g)
        if need_w:
            K.wgrad_use_done(conv.weight)
            R = conv.kernel_size[0]
            scale = conv.packed().scale if conv.norm is not None else None
            if K.can_defer_wgrad(x, g) and conv._grad_lands_in_weight():
                # off the critical path: queued, launched with the other layers' (kernels.flush_wgrad) and written to weight.grad
                # before backward() returns -- this node hands autograd no gradient for the weight
                K.defer_wgrad(conv.weight, x, g, scale, R, conv.stride, conv.padding)
            else:
                dw = K.conv_wgrad(x, g, scale, R, R, conv.stride, conv.padding)
                dw = dw.permute(0, 3, 1, 2).contiguous()   # [K,R,S,C] -> the parameter's OIHW
        if need_b:
            db = K.colsum_rows(g.view(-1, g.shape[-1]))
        if need_x:
            dx = K.conv_dgrad(g, conv.packed_dgrad(), x.shape, conv.stride)
        return dx, dw, db, dres, None, None, None


class _LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) with the fused GEMM epilogue; backward = masked upstream gradient (lvc_relu_backward),
    bias gradient (lvc_colsum) and the two GEMMs dX = dZ W, dW = dZ^T X on the same conv/GEMM kernel
    (`kernels.linear_backward`).  `w_view` maps a gradient in the layout of the packed weight back to the layout of
    the parameter (the box head packs fc1 in (h, w, c) column order for channels-last RoI features)."""

    @staticmethod
    def forward(ctx, x, weight, bias, packed, relu, w_packed_layout, w_view):
        y = K.linear(x.contiguous(), packed, relu=relu)
        ctx.save_for_backward(x, y if relu else None)
        ctx.relu = relu
        ctx.weight = w_packed_layout() if w_packed_layout is not None else weight
        ctx.w_view = w_view
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        dz = K.relu_backward(dy, y) if ctx.relu else dy.contiguous()
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx, dw = K.linear_backward(x, ctx.weight, dz, need_dx=need_dx, need_dw=need_dw)
        if dw is not None and ctx.w_view is not None:
            dw = ctx.w_view(dw)
        db = K.colsum_rows(dz) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None, None, None


def linear_fn(x, weight, bias, packed, relu=False, w_packed_layout=None, w_view=None):
    """Differentiable fused linear layer when anything requires grad, plain kernel call otherwise."""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return _LinearFn.apply(x, weight, bias, packed, relu, w_packed_layout, w_view)
    return K.linear(x.contiguous(), packed, relu=relu)


class Linear(nn.Module):
    """nn.Linear-compatible parameters (`weight` [out,in], `bias`), forward on the MFMA GEMM kernel."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self._cache = _PackedCache()
        self._range_state = {"tier": 0}

    def packed(self):
        pc = self._cache.get([self.weight, self.bias], lambda: K.pack_linear(self.weight, self.bias))   # two_acc: kernels.pack_linear
        pc.state = self._range_state
        return pc

    def forward(self, x, relu=False):
        require_device(x, "Linear")
        return linear_fn(x, self.weight, self.bias, self.packed(), relu=relu)

    def extra_repr(self):
        return "in_features={}, out_features={}, bias={}".format(self.in_features, self.out_features, self.bias is not None)
