"""DINO ViT-S/8 descriptor network on the gfx950 kernels.

The reference obtains it with `torch.hub.load('facebookresearch/dino:main', 'dino_vits8')`
(tools/run_nearest_neighbours.py:292-293) and calls it on normalised 224 x 224 crops (`get_descriptors`, :102-128); the output
is the class token after the last LayerNorm, 384-d.  `VisionTransformer` here keeps that repository's parameter names and
shapes (`cls_token`, `pos_embed`, `patch_embed.proj.*`, `blocks.{i}.norm1/attn.qkv/attn.proj/norm2/mlp.fc1/mlp.fc2.*`,
`norm.*`), so its checkpoints load with `load_state_dict`; there is no network here, so tests use seeded random weights.

Forward, all on the device: patch gather + one GEMM (the 8 x 8 stride-8 convolution), class token + position embedding,
12 x [LayerNorm -> qkv GEMM -> multi-head attention (online softmax) -> projection GEMM with the residual in its epilogue ->
LayerNorm -> fc1 GEMM -> GELU -> fc2 GEMM with the residual], final LayerNorm of the class rows.  The GEMMs are the
fp32-accurate split-operand MFMA kernels of the detector (kernels.conv2d_nhwc on [M,1,1,C] rows).
"""
import math

import torch
from torch import nn

from .. import kernels as K


CLASS_ROWS_ONLY = True     # the last block evaluated at the class rows only (False: every token, then the class rows are read)


class _Cache:
    """Packed GEMM operands of a Linear, rebuilt when a parameter changes (as layers/wrappers.py does)."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, tensors, build):
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self.key:
            self.key, self.val = key, build()
        return self.val


class _Linear(nn.Module):
    def __init__(self, din, dout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dout, din))
        self.bias = nn.Parameter(torch.zeros(dout)) if bias else None
        nn.init.trunc_normal_(self.weight, std=0.02)
        self._packed = _Cache()

    def packed(self):
        return self._packed.get([self.weight] + ([self.bias] if self.bias is not None else []),
                                lambda: K.pack_linear(self.weight, self.bias, two_acc=False))

    def forward(self, x, residual=None, act=None):
        """x [M, din] device rows -> [M, dout]; `residual` [M, dout] is added in the GEMM epilogue; act="gelu": the exact
        GELU in the same epilogue."""
        pc = self.packed()
        M = x.shape[0]
        y = K.conv2d_nhwc(x.view(M, 1, 1, x.shape[1]), pc,
                          residual=residual.view(M, 1, 1, residual.shape[1]) if residual is not None else None, act=act)
        return y.view(M, -1)


class _Norm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps

    def forward(self, x):
        return K.layernorm(x, self.weight, self.bias, self.eps)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = _Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = _Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = _Linear(dim, hidden)
        self.fc2 = _Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias):
        super().__init__()
        self.norm1 = _Norm(dim)
        self.attn = _Attention(dim, num_heads, qkv_bias)
        self.norm2 = _Norm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, t, B, N):
        a = self.attn
        x = self.norm1(t)
        pc = a.qkv.packed()
        if K.can_qkv_planes(x, pc, a.num_heads, t.shape[1] // a.num_heads):
            # the qkv GEMM writes the attention's fp16 operand planes from its epilogue: no fp32 qkv tensor, no split pass (bit-identical)
            y = K.qkv_attention(x, pc, B, N, a.num_heads, a.scale)
        else:
            y = K.mha(a.qkv(x), B, N, a.num_heads, t.shape[1] // a.num_heads, a.scale)
        t = a.proj(y, residual=t)
        y = self.mlp.fc1(self.norm2(t), act="gelu")      # GELU in fc1's epilogue: the [M, 4 dim] hidden map makes one trip less
        return self.mlp.fc2(y, residual=t)

    def forward_class_rows(self, t, B, N):
        """The block's output at the class rows only, [B, dim] -- all the network reads of its LAST block: keys and values of every
        token still come from the full qkv GEMM, but attention runs for one query per image and the projection / MLP on B rows instead of
        B * N (785 x fewer)."""
        a = self.attn
        D = t.shape[1]
        qkv = a.qkv(self.norm1(t))
        y = K.mha_cls(qkv, B, N, a.num_heads, D // a.num_heads, a.scale)
        c = a.proj(y, residual=t.view(B, N, D)[:, 0].contiguous())
        y = self.mlp.fc1(self.norm2(c), act="gelu")
        return self.mlp.fc2(y, residual=c)


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self._packed = _Cache()


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=8, in_chans=3, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.embed_dim = self.num_features = embed_dim
        self.img_size = img_size
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.norm = _Norm(embed_dim)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)

    @property
    def device(self):
        return self.cls_token.device

    def forward(self, x, pixel_mean=None, pixel_std=None):
        """x [B, 3, 224, 224] crops on the device -> [B, embed_dim] class-token descriptors.  x is normalised already
        (reference `preprocess_crops`), or raw with pixel_mean / pixel_std given: the normalisation then happens inside the
        patch gather (one pass less over the crops, the same fp32 arithmetic)."""
        if not x.is_cuda:
            raise RuntimeError("lvc_amd VisionTransformer needs device tensors; there is no CPU path")
        B, C, H, W = x.shape
        pe = self.patch_embed
        if H != self.img_size or W != self.img_size:
            raise NotImplementedError("descriptor crops are %d x %d (get_crops_qe); position-embedding interpolation for other "
                                      "sizes is not on this path" % (self.img_size, self.img_size))
        kc = C * pe.patch_size * pe.patch_size
        kpad = (kc + 31) // 32 * 32
        pc = pe._packed.get([pe.proj.weight, pe.proj.bias], lambda: self._pack_patch(pe, kc, kpad))
        patches = K.vit_patchify(x.float().contiguous(), pe.patch_size, kpad, mean=pixel_mean, std=pixel_std)
        M = patches.shape[0]
        emb = K.conv2d_nhwc(patches.view(M, 1, 1, kpad), pc).view(M, self.embed_dim)
        N = pe.num_patches + 1
        t = K.vit_tokens(emb, self.cls_token.view(-1), self.pos_embed.view(N, self.embed_dim), B)
        for blk in self.blocks[:-1]:
            t = blk(t, B, N)
        if CLASS_ROWS_ONLY and N <= 1024 and self.embed_dim // self.blocks[-1].attn.num_heads == 64:
            return self.norm(self.blocks[-1].forward_class_rows(t, B, N))
        t = self.blocks[-1](t, B, N)
        cls_rows = t.view(B, N, self.embed_dim)[:, 0].contiguous()
        return self.norm(cls_rows)

    @staticmethod
    def _pack_patch(pe, kc, kpad):
        w = pe.proj.weight.detach().reshape(pe.proj.weight.shape[0], kc)
        if kpad != kc:
            w = torch.nn.functional.pad(w, (0, kpad - kc))
        return K.pack_linear(w.contiguous(), pe.proj.bias, two_acc=False)


def vit_small(patch_size=8, **kw):
    """dino_vits8 / dino_vits16 geometry (vision_transformer.py vit_small of the DINO repository)."""
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4.0, qkv_bias=True, **kw)


def seeded_state_dict_(model, seed=0):
    """Deterministic stand-in weights (there is no network for the DINO checkpoint): N(0, 0.02)-scale matrices as the
    published init, but non-trivial biases / LayerNorm parameters so that every term of the forward is exercised."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k == "norm.weight":
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith(".bias"):
            sd[k] = 0.05 * torch.randn(v.shape, generator=g)
        elif k in ("cls_token", "pos_embed"):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        else:
            fan_in = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) * (1.0 / math.sqrt(fan_in))
    model.load_state_dict(sd)
    return sd
