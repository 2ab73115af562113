"""ResNet bottom-up trunk: `BasicStem`, `BottleneckBlock`, `ResNet`, `build_resnet_backbone`.

Same module tree / parameter names / config keys as reference
detectron2/modeling/backbone/resnet.py:101-211 (BottleneckBlock), :564-592 (BasicStem),
:648-763 (ResNet), :845-941 (builder), so reference checkpoints load unchanged.  Every conv+FrozenBN
(+ReLU, +residual add) is one launch of an implicit-GEMM MFMA kernel (fp32 operands split into fp16/bf16
planes with fp32 accumulation, fp32-accurate: DESIGN.md section 3; `LVC_CONV_ENGINE=f32` selects the exact fp32
MFMA form); activations stay NHWC fp32 in HBM between layers.  BasicBlock / DeepStem / Dropout / CLIP / Deform variants are not selected by any
shipped config and are not provided (the builder raises for them).
"""
import torch
import torch.nn.functional as F
from torch import nn

from ... import kernels as K
from ...layers import Conv2d, ShapeSpec, get_norm
from ...layers.layout import require_device, to_nchw_view, to_nhwc
from ...utils import weight_init
from .backbone import BACKBONE_REGISTRY, Backbone


class CNNBlockBase(nn.Module):
    """reference detectron2/layers/blocks.py: records in/out channels + stride; `freeze()` stops grads."""

    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        return self


FUSE_STRIDED_PROJECTION = True      # conv3 + a stride-2 projection shortcut as one GEMM (BottleneckBlock.can_fuse_projection)


class BottleneckBlock(CNNBlockBase):
    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN",
                 stride_in_1x1=False, dilation=1):
        super().__init__(in_channels, out_channels, stride)
        if num_groups != 1 or dilation != 1:
            raise NotImplementedError("grouped / dilated bottlenecks are not used by the shipped configs")
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=get_norm(norm, out_channels))
        else:
            self.shortcut = None
        stride_1x1, stride_3x3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=stride_1x1, bias=False,
                            norm=get_norm(norm, bottleneck_channels), activation=F.relu_)
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=stride_3x3, padding=1,
                            bias=False, norm=get_norm(norm, bottleneck_channels), activation=F.relu_)
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False,
                            norm=get_norm(norm, out_channels))
        for layer in [self.conv1, self.conv2, self.conv3, self.shortcut]:
            if layer is not None:
                weight_init.c2_msra_fill(layer)

    def can_fuse_projection(self):
        """Gradient-free passes only (inference, or a frozen block in training -- res2 under FREEZE_AT 2): conv3 and a
        stride-1 projection shortcut (res2.0: both 64 -> 256 at 1/4 resolution, both HBM streams) as ONE pointwise GEMM
        over the concatenated input [conv2 output | block input]."""
        grad_free = not torch.is_grad_enabled() or not any(p.requires_grad for p in self.parameters())
        # stride 2 in the 1x1 layers (res3.0 / res4.0 / res5.0 with STRIDE_IN_1X1): the shortcut samples x[:, ::2, ::2]; that sampling is
        # a copy into the buffer's tail channels (kernels.subsample2_into) -- the strided projection launch (168 us on res3.0 for 137 MB
        # read + 275 MB written) and the residual round trip of its output go away
        strided = self.shortcut is not None and self.shortcut.stride == 2 and self.conv1.stride == 2
        return (K.FUSE_PROJECTION and K.CONV_ENGINE == "bf16x3" and grad_free and self.shortcut is not None
                and (self.shortcut.stride == 1 or (strided and FUSE_STRIDED_PROJECTION)) and self.conv2.stride == 1
                and (self.conv2.out_channels + self.in_channels) % 32 == 0 and self.conv2.out_channels % 4 == 0)

    def _fused_projection(self):
        """relu(bn3(conv3(t)) + bn_s(shortcut(x))) = relu([s3 W3 | s_s W_s] [t | x] + (b3 + b_s)): the FrozenBN scales go
        into the weights (one fp32 rounding per weight), the shifts add.  Saves writing the shortcut tensor and reading
        it back as the residual (2 x 550 MB at batch 8)."""
        a3, asc = self.conv3._affine(), self.shortcut._affine()

        def build():
            w3, ws = self.conv3.weight.detach().float(), self.shortcut.weight.detach().float()
            if a3[0] is not None:
                w3 = w3 * a3[0].view(-1, 1, 1, 1)
            if asc[0] is not None:
                ws = ws * asc[0].view(-1, 1, 1, 1)
            shifts = [t for t in (a3[1], asc[1]) if t is not None]
            shift = None if not shifts else shifts[0] if len(shifts) == 1 else shifts[0] + shifts[1]
            return K.pack_conv(torch.cat([w3, ws], 1).contiguous(), affine=(None, shift))

        if not hasattr(self, "_cache_fused"):
            from ...layers.wrappers import _PackedCache
            self._cache_fused = _PackedCache()
        return self._cache_fused.get([self.conv3.weight, self.shortcut.weight, a3[0], a3[1], asc[0], asc[1]], build)

    def _grad_free(self):
        return not torch.is_grad_enabled() or not any(p.requires_grad for p in self.parameters())

    def chain_to(self, nxt, fused_projection):
        """conv3 (+ shortcut add + ReLU) of this block and conv1 (+ ReLU) of `nxt` as ONE launch (csrc/conv_pw_chain.hip): the
        4x-wide block output is written once and never read back by conv1.  Returns the packed pair, or None when the pair is
        outside that kernel's range (gradient passes, a strided conv1, channel counts it has no instance for, a layer that must
        keep two accumulators, the range-free split)."""
        if nxt is not None and nxt.fused_eligible():
            return None     # the next block is one launch of its own: it reads this block's output, not a conv1 result
        if not (K.CHAIN and K.CONV_ENGINE == "bf16x3" and K.CONV_SPLIT == "f16x2" and K.PW_S1 == 2 and nxt is not None
                and self._grad_free() and nxt._grad_free() and nxt.conv1.stride == 1 and self.conv2.stride == 1
                and not getattr(self.conv3, "two_acc", False) and not getattr(nxt.conv1, "two_acc", False)):
            return None
        k1 = self.conv2.out_channels + (self.in_channels if fused_projection else 0)
        if (k1, self.out_channels, nxt.conv1.out_channels) not in K.CHAIN_SHAPES:
            return None
        if not hasattr(self, "_cache_chain"):
            from ...layers.wrappers import _PackedCache
            self._cache_chain = _PackedCache()
            self._chain_state = {"off": False}
        if self._chain_state["off"] or self.conv3._range_state["tier"] or nxt.conv1._range_state["tier"]:
            return None     # the pair left the chain kernel's range once (|a| > 4094): two launches on their own tiers from then on
        pa = self._fused_projection() if fused_projection else self.conv3.packed()
        pb = nxt.conv1.packed()
        return self._cache_chain.get([pa.w, pa.scale, pa.shift, pb.w, pb.scale, pb.shift], lambda: K.pack_chain(pa, pb, self._chain_state))

    def fused_eligible(self):
        """The cheap half of `fused`: everything but the packing (flags, shapes, gradient-free pass, range state).  `ResNet.forward_nhwc`
        decides with it whether the stem writes a second copy of its output BEFORE the stem is launched -- at the top of a step the GPU
        queue is empty and every host microsecond in front of the first launches is exposed."""
        if not (K.BNECK and K.CONV_ENGINE == "bf16x3" and K.CONV_SPLIT == "f16x2" and self.conv1.stride == 1 and self.conv2.stride == 1
                and self.conv2.out_channels == 64):
            return False
        proj = self.shortcut is not None
        if (self.in_channels, self.conv2.in_channels, self.out_channels, proj) not in K.BNECK_SHAPES:
            return False
        if getattr(self, "_bneck_state", None) is not None and self._bneck_state["off"]:
            return False     # an operand left the kernel's range once (|a| > 4094): the block's layers run on their own tiers from then on
        if not self._grad_free():
            return False
        if proj and (self.shortcut.stride != 1 or not self.can_fuse_projection()):
            return False
        layers = (self.conv1, self.conv2, self.conv3, self.shortcut) if proj else (self.conv1, self.conv2, self.conv3)
        return not any(l.norm is None or getattr(l, "two_acc", False) or l._range_state["tier"] for l in layers)

    def fused(self):
        """The whole block as ONE launch (csrc/conv_bneck.hip: conv1's output in LDS, conv2's in registers) -- the packed weights, or None
        when the block is outside that kernel's range: gradient passes, strides, channel counts other than res2's (64 mid / 256 out,
        256 in with the identity shortcut or 64 in with a stride-1 projection), a layer off the one-accumulator tier, `K.BNECK` off."""
        if not self.fused_eligible():
            return None
        proj = self.shortcut is not None
        if not hasattr(self, "_cache_bneck"):
            from ...layers.wrappers import _PackedCache
            self._cache_bneck = _PackedCache()
            self._bneck_state = {"off": False}
        p1, p2 = self.conv1.packed(), self.conv2.packed()
        p3 = self._fused_projection() if proj else self.conv3.packed()
        return self._cache_bneck.get([p1.w, p1.scale, p1.shift, p2.w, p2.scale, p2.shift, p3.w, p3.scale, p3.shift],
                                     lambda: K.pack_bottleneck(p1, p2, p3, proj, self._bneck_state))

    def forward_chained(self, x, t, nxt, concat=None):
        """One block of a stage walked with look-ahead.  x: the block's input; t: conv1's output when the PREVIOUS block's
        tail already produced it (else None); nxt: the following block of the stage (or None).  Returns (block output,
        conv1 output of `nxt` or None).  concat: as `forward_nhwc`."""
        if t is None and concat is None:
            bk = self.fused()
            if bk is not None and x.shape[0] * x.shape[1] * x.shape[2] * max(x.shape[3], 256) * 4 < 0xFFF00000:
                return K.bottleneck_fused(x, bk), None
        if t is None:
            t = self.conv1.forward_nhwc(x)
        if (concat is None and self.shortcut is not None and self.shortcut.stride == 2 and self.can_fuse_projection()
                and self.chain_to(nxt, False) is None):      # res3.0 keeps its conv3 -> res3.1.conv1 chain (shortcut as the residual)
            # a stage head with stride 2: [conv2 output | x sampled at the even pixels] built here (res2.0's buffer comes from the stem)
            cb = self.conv2.out_channels
            concat = torch.empty(t.shape[0], t.shape[1], t.shape[2], cb + self.in_channels, device=t.device, dtype=torch.float32)
            K.subsample2_into(x, concat[..., cb:])
        if concat is not None:
            K.conv2d_nhwc(t, self.conv2.packed(), relu=True, out=concat)      # channels [0, bottleneck), row stride = buffer
            ch = self.chain_to(nxt, True)
            if ch is not None and concat.numel() < (1 << 29):
                return K.conv1x1_chain(concat, ch)
            return K.conv2d_nhwc(concat, self._fused_projection(), relu=True), None
        ch = self.chain_to(nxt, False)
        if (ch is None and self.conv2.activation is not None and self._grad_free()
                and not (torch.is_grad_enabled() and (t.requires_grad or x.requires_grad))):
            p2, p3 = self.conv2.packed(), self.conv3.packed()
            shortcut = self.shortcut.forward_nhwc(x) if self.shortcut is not None else x
            if K.presplit_pair_ok(t, p2, p3, shortcut):
                # conv2's output goes to conv3 as the two fp16 planes conv3 multiplies (res4 / res5: conv3 skips its operand split)
                return K.conv3x3_conv1x1_presplit(t, p2, p3, residual=shortcut, relu=True), None
            z = self.conv2.forward_nhwc(t)
            return self.conv3.forward_nhwc(z, residual=shortcut, res_mode=1, relu=True), None
        z = self.conv2.forward_nhwc(t)
        shortcut = self.shortcut.forward_nhwc(x) if self.shortcut is not None else x
        if ch is not None and shortcut.numel() < (1 << 29):
            return K.conv1x1_chain(z, ch, residual=shortcut)
        # conv3 + FrozenBN + residual add + ReLU in one epilogue (reference resnet.py:205-211)
        return self.conv3.forward_nhwc(z, residual=shortcut, res_mode=1, relu=True), None

    def forward_nhwc(self, x, concat=None):
        """concat: [N,H,W, bottleneck + in] buffer whose LAST in_channels already hold x (`can_fuse_projection`)."""
        return self.forward_chained(x, None, None, concat=concat)[0]

    def forward(self, x):
        return to_nchw_view(self.forward_nhwc(to_nhwc(x)))


class BasicStem(CNNBlockBase):
    def __init__(self, in_channels=3, out_channels=64, norm="BN"):
        super().__init__(in_channels, out_channels, 4)
        self.in_channels = in_channels
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=7, stride=2, padding=3, bias=False,
                            norm=get_norm(norm, out_channels), activation=F.relu_)
        weight_init.c2_msra_fill(self.conv1)

    def forward_nhwc(self, x4, second=None):
        """x4: [N,H,W,4] (RGB + zero slot).  second: optional callable shape -> [N,Hp,Wp,64] strided view that receives
        a copy of the output (the concat buffer of a fused projection block)."""
        if (K.CONV_ENGINE == "bf16x3" and K.CONV_SPLIT == "f16x2" and K.STEM_FUSED and self.conv1.out_channels == 64
                and self.conv1.norm is not None):
            return K.stem_conv_pool(x4, self.conv1.packed(), relu=True, second=second)   # conv + FrozenBN + ReLU + max-pool, one launch
        y = self.conv1.forward_nhwc(x4)
        y = K.maxpool2d_nhwc(y, 3, 2, 1)
        if second is not None:
            second(y.shape).copy_(y)
        return y

    def forward(self, x):
        return to_nchw_view(self.forward_nhwc(_as_nhwc4(x)))


def _as_nhwc4(x):
    """NCHW-shaped 3-channel image batch -> [N,H,W,4] buffer; zero-copy when x is the view that
    GeneralizedRCNN.preprocess_image hands out (storage already NHWC4)."""
    N, C, H, W = x.shape
    assert C == 3
    s = x.stride()
    if s == (H * W * 4, 1, W * 4, 4) and x.storage_offset() % 4 == 0:
        return x.as_strided((N, H, W, 4), (H * W * 4, W * 4, 4, 1), x.storage_offset())
    return F.pad(x.permute(0, 2, 3, 1), (0, 1)).contiguous()


class ResNet(Backbone):
    def __init__(self, stem, stages, num_classes=None, out_features=None):
        super().__init__()
        if num_classes is not None:
            raise NotImplementedError("classification head is not on the detection path")
        self.stem = stem
        current_stride = self.stem.stride
        self._out_feature_strides = {"stem": current_stride}
        self._out_feature_channels = {"stem": self.stem.out_channels}
        self.stages_and_names = []
        for i, blocks in enumerate(stages):
            assert len(blocks) > 0, len(blocks)
            for block in blocks:
                assert isinstance(block, CNNBlockBase), block
            name = "res" + str(i + 2)
            stage = nn.Sequential(*blocks)
            self.add_module(name, stage)
            self.stages_and_names.append((stage, name))
            current_stride = int(current_stride * _prod([k.stride for k in blocks]))
            self._out_feature_strides[name] = current_stride
            self._out_feature_channels[name] = blocks[-1].out_channels
        if out_features is None:
            out_features = [name]
        self._out_features = out_features
        assert len(self._out_features)
        children = [x[0] for x in self.named_children()]
        for f in self._out_features:
            assert f in children, "Available children: {}".format(", ".join(children))

    def forward_nhwc(self, x4):
        outputs = {}
        first = self.stages_and_names[0][0][0] if self.stages_and_names else None
        concat = []
        if (isinstance(first, BottleneckBlock) and first.shortcut is not None and first.shortcut.stride == 1 and first.can_fuse_projection()
                and not first.fused_eligible()):   # the stem writes into a concat buffer at ITS resolution: stride-1 projections only
            # the stem writes its output a second time, into the tail channels of res2.0's [conv2 output | x] buffer
            def second(shape):
                n, h, w, c = shape
                concat.append(torch.empty(n, h, w, first.conv2.out_channels + c, device=x4.device, dtype=torch.float32))
                return concat[0][..., first.conv2.out_channels:]
            x = self.stem.forward_nhwc(x4, second=second)
        else:
            x = self.stem.forward_nhwc(x4)
        if "stem" in self._out_features:
            outputs["stem"] = x
        for stage, name in self.stages_and_names:
            blocks = list(stage)
            t = None     # conv1 output of the next block when the previous block's tail launch produced it (BottleneckBlock.chain_to)
            for i, blk in enumerate(blocks):
                nxt = blocks[i + 1] if i + 1 < len(blocks) else None
                if isinstance(blk, BottleneckBlock):
                    x, t = blk.forward_chained(x, t, nxt if isinstance(nxt, BottleneckBlock) else None,
                                               concat=concat[0] if (blk is first and concat) else None)
                else:
                    x, t = blk.forward_nhwc(x), None
            if name in self._out_features:
                outputs[name] = x
        return outputs

    def forward(self, x):
        require_device(x, "ResNet")
        return {k: to_nchw_view(v) for k, v in self.forward_nhwc(_as_nhwc4(x)).items()}

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}

    def freeze(self, freeze_at=0):
        """Freeze the stem (1) and the first `freeze_at - 1` res stages (reference resnet.py:733-757)."""
        if freeze_at >= 1:
            self.stem.freeze()
        for idx, (stage, _) in enumerate(self.stages_and_names, start=2):
            if freeze_at >= idx:
                for block in stage.children():
                    block.freeze()
        return self

    @staticmethod
    def make_stage(block_class, num_blocks, first_stride=None, *, in_channels, out_channels, **kwargs):
        if first_stride is not None:
            assert "stride" not in kwargs and "stride_per_block" not in kwargs
            kwargs["stride_per_block"] = [first_stride] + [1] * (num_blocks - 1)
        blocks = []
        for i in range(num_blocks):
            curr = {}
            for k, v in kwargs.items():
                if k.endswith("_per_block"):
                    assert len(v) == num_blocks
                    curr[k[: -len("_per_block")]] = v[i]
                else:
                    curr[k] = v
            blocks.append(block_class(in_channels=in_channels, out_channels=out_channels, **curr))
            in_channels = out_channels
        return blocks


def _prod(xs):
    p = 1
    for x in xs:
        p *= x
    return p


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape):
    R = cfg.MODEL.RESNETS
    norm = R.NORM
    unsupported = []
    if R.get("D", False):
        unsupported.append("RESNETS.D (DeepStem/CLIP blocks)")
    if R.get("DROPOUT", 0):
        unsupported.append("RESNETS.DROPOUT")
    if any(R.DEFORM_ON_PER_STAGE):
        unsupported.append("RESNETS.DEFORM_ON_PER_STAGE")
    if R.DEPTH in (18, 34):
        unsupported.append("BasicBlock depths 18/34")
    if R.NUM_GROUPS != 1 or R.RES5_DILATION != 1:
        unsupported.append("grouped/dilated res5")
    if unsupported:
        raise NotImplementedError("not on the path of any shipped config: " + ", ".join(unsupported))
    stem = BasicStem(in_channels=input_shape.channels, out_channels=R.STEM_OUT_CHANNELS, norm=norm)
    freeze_at = cfg.MODEL.BACKBONE.FREEZE_AT
    out_features = R.OUT_FEATURES
    bottleneck_channels = R.NUM_GROUPS * R.WIDTH_PER_GROUP
    in_channels, out_channels = R.STEM_OUT_CHANNELS, R.RES2_OUT_CHANNELS
    num_blocks_per_stage = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[R.DEPTH]
    stages = []
    max_stage_idx = max({"res2": 2, "res3": 3, "res4": 4, "res5": 5}[f] for f in out_features)
    for idx, stage_idx in enumerate(range(2, max_stage_idx + 1)):
        first_stride = 1 if idx == 0 else 2
        blocks = ResNet.make_stage(
            block_class=BottleneckBlock, num_blocks=num_blocks_per_stage[idx],
            stride_per_block=[first_stride] + [1] * (num_blocks_per_stage[idx] - 1),
            in_channels=in_channels, out_channels=out_channels, norm=norm,
            bottleneck_channels=bottleneck_channels, stride_in_1x1=R.STRIDE_IN_1X1, dilation=1, num_groups=1)
        in_channels = out_channels
        out_channels *= 2
        bottleneck_channels *= 2
        stages.append(blocks)
    return ResNet(stem, stages, out_features=out_features).freeze(freeze_at)
