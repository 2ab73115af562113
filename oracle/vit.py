"""CPU oracle of the descriptor network of the label-verification step.  TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

The reference loads it with `torch.hub.load('facebookresearch/dino:main', cfg.QUERY_EXPAND.NN_MODEL)`
(tools/run_nearest_neighbours.py:292-293; NN_MODEL = dino_vits8 in configs/LABEL-Verification/dino_label_verification.yaml:11):
a third-party model that is NOT under /root/reference and cannot be fetched here (no network), so this file restates the
published architecture -- `VisionTransformer` of that repository's vision_transformer.py with the vit_small settings:
patch 8, embed_dim 384, depth 12, 6 heads, mlp_ratio 4, qkv_bias=True, LayerNorm eps 1e-6, exact GELU, class token and a
learned position embedding of 1 + (224/8)^2 = 785 rows, output = the normalised class token -- as a function of a state_dict
with that repository's parameter names.  "parity unpinned": the reference holds no test vector for it; the restatement is
checked against torch's own nn.MultiheadAttention-free arithmetic only (tests/test_oracle_golden.py) and serves as the
checker of the HIP path on seeded random weights.

get_descriptors / preprocess_crops follow tools/run_nearest_neighbours.py:95-128."""
import torch
import torch.nn.functional as F


def vit_forward(sd, x, patch_size=8, num_heads=6, eps=1e-6):
    """x [B,3,H,W] (already normalised) -> [B, D] class-token descriptors (DINO VisionTransformer.forward)."""
    w = sd["patch_embed.proj.weight"]
    D = w.shape[0]
    B = x.shape[0]
    t = F.conv2d(x, w, sd["patch_embed.proj.bias"], stride=patch_size).flatten(2).transpose(1, 2)      # [B, P, D]
    t = torch.cat([sd["cls_token"].expand(B, -1, -1), t], dim=1)
    assert t.shape[1] == sd["pos_embed"].shape[1], "crops are 224 x 224: no position-embedding interpolation on this path"
    t = t + sd["pos_embed"]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    hd = D // num_heads
    for i in range(depth):
        p = "blocks.%d." % i
        y = F.layer_norm(t, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, -1, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
        y = (attn @ v).transpose(1, 2).reshape(B, -1, D)
        t = t + F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        y = F.layer_norm(t, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        y = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        t = t + F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    t = F.layer_norm(t, (D,), sd["norm.weight"], sd["norm.bias"], eps)
    return t[:, 0]


def preprocess_crops(crops, mean, std):
    """tools/run_nearest_neighbours.py:95-99: (crops.float() - mean) / std, mean / std as [1,3,1,1]."""
    return (crops.float() - torch.tensor(mean).view(1, -1, 1, 1)) / torch.tensor(std).view(1, -1, 1, 1)
