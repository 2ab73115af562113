"""the descriptor network (DINO ViT-S/8 geometry) and `get_descriptors` on the device against the CPU oracle
(oracle/vit.py; seeded random weights -- the published checkpoint needs the network), and the whole verification chain
image -> crops -> descriptors -> kNN -> keep."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_layernorm_gelu_patchify_tokens():
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(1)
    x = torch.randn(1571, 384, generator=g) * 3 + 0.5
    w, b = torch.randn(384, generator=g), torch.randn(384, generator=g)
    y = K.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6).cpu()
    ref = F.layer_norm(x, (384,), w, b, 1e-6)
    assert float((y - ref).abs().max()) <= 2e-5
    x2 = torch.randn(785, 1536, generator=g) * 2
    assert float((K.gelu(x2.cuda()).cpu() - F.gelu(x2)).abs().max()) <= 4e-6      # values up to 8: a few ulp (erff vs the CPU erf)
    img = torch.randn(3, 3, 224, 224, generator=g)
    p = K.vit_patchify(img.cuda(), 8).cpu()
    ref = F.unfold(img, kernel_size=8, stride=8).transpose(1, 2).reshape(-1, 192)     # [B, C*64, P] -> rows (b, p)
    assert torch.equal(p, ref)
    emb = torch.randn(3 * 784, 384, generator=g)
    cls, pos = torch.randn(384, generator=g), torch.randn(785, 384, generator=g)
    t = K.vit_tokens(emb.cuda(), cls.cuda(), pos.cuda(), 3).cpu().view(3, 785, 384)
    ref = torch.cat([cls.expand(3, 1, 384), emb.view(3, 784, 384)], 1) + pos
    assert torch.equal(t, ref)


@pytest.mark.parametrize("mfma", [True, False])
@pytest.mark.parametrize("B,N", [(2, 785), (1, 197), (3, 64), (2, 1), (1, 130)])
def test_attention_matches_torch(B, N, mfma):
    """Both attention kernels (matrix-core: lvc_mha_mfma, the default; scalar fp32: lvc_mha) against fp64, at the error level of
    torch's own fp32 CPU evaluation; token counts that are not multiples of the 64-key tile / the 128-query block."""
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(N)
    H, Dh = 6, 64
    qkv = torch.randn(B * N, 3 * H * Dh, generator=g) * 1.5
    out = K.mha(qkv.cuda(), B, N, H, Dh, Dh ** -0.5, mfma=mfma).cpu()
    t = qkv.double().view(B, N, 3, H, Dh).permute(2, 0, 3, 1, 4)
    ref = (((t[0] @ t[1].transpose(-2, -1)) * Dh ** -0.5).softmax(-1) @ t[2]).transpose(1, 2).reshape(B * N, H * Dh)
    t32 = qkv.view(B, N, 3, H, Dh).permute(2, 0, 3, 1, 4)
    cpu32 = (((t32[0] @ t32[1].transpose(-2, -1)) * Dh ** -0.5).softmax(-1) @ t32[2]).transpose(1, 2).reshape(B * N, H * Dh)
    err = float((out.double() - ref).abs().max())
    err32 = float((cpu32.double() - ref).abs().max())
    print("attention (%s) B=%d N=%d: max abs error vs fp64 %.2e (torch CPU fp32: %.2e)" % ("mfma" if mfma else "valu", B, N, err, err32))
    assert err <= 2 * err32 + 1e-6


def _model(seed=0):
    from lvc_amd.modeling.vit import seeded_state_dict_, vit_small

    m = vit_small(8)
    sd = seeded_state_dict_(m, seed)
    return m.cuda().eval(), sd


def test_vit_s8_descriptors_match_oracle():
    from oracle import vit as ovit

    model, sd = _model()
    assert sum(p.numel() for p in model.parameters()) == 21670272       # ViT-S/8 without a head
    g = torch.Generator().manual_seed(7)
    x = torch.randn(9, 3, 224, 224, generator=g)                         # 9 crops: 7065 token rows -> the fp16x2 GEMMs
    with torch.no_grad():
        got = model(x.cuda()).cpu()
        ref = ovit.vit_forward(sd, x)
        one = model(x[:1].cuda()).cpu()                                 # 785 rows: the small-M engines
    rel = float((got - ref).abs().max() / ref.abs().max())
    rel1 = float((one - ref[:1]).abs().max() / ref.abs().max())
    print("ViT-S/8 class-token descriptors: max error relative to the descriptor scale %.2e (batch 9), %.2e (batch 1)" % (rel, rel1))
    assert got.shape == (9, 384) and rel <= 1e-4 and rel1 <= 1e-4
    # the last block at the class rows only (modeling/vit.py CLASS_ROWS_ONLY, csrc/vit.hip mha_cls_kernel) against every token + read
    from lvc_amd.modeling import vit as V

    V.CLASS_ROWS_ONLY = False
    try:
        with torch.no_grad():
            full = model(x.cuda()).cpu()
    finally:
        V.CLASS_ROWS_ONLY = True
    d = float((got - full).abs().max() / ref.abs().max())
    relf = float((full - ref).abs().max() / ref.abs().max())
    print("   class rows only vs every token: %.2e apart; every-token form vs the oracle %.2e" % (d, relf))
    assert d <= 2e-5 and relf <= 1e-4


def test_verification_chain_crops_descriptors_knn_keep():
    """image -> get_crops_qe -> preprocess_crops -> ViT -> assemble_tensors -> kNN top-10 -> vote, device path against the
    oracle restatements of every stage (reference tools/run_nearest_neighbours.py:95-162, 214-227, 285-325)."""
    from lvc_amd import label_verification as LV
    from lvc_amd import wire
    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn
    from oracle import knn as oknn
    from oracle import vit as ovit

    model, sd = _model(3)
    mean, std = [123.675, 116.280, 103.530], [58.395, 57.120, 57.375]
    g = torch.Generator().manual_seed(5)

    def items(n_img, n_box, seed0, ncls):
        out = []
        for i in range(n_img):
            img = syn.synthetic_image(seed0 + i, 300, 420).round()
            boxes = []
            for _ in range(n_box):
                x1 = int(torch.randint(0, 330, (1,), generator=g)); y1 = int(torch.randint(0, 220, (1,), generator=g))
                boxes.append([x1, y1, x1 + int(torch.randint(20, 88, (1,), generator=g)), y1 + int(torch.randint(20, 78, (1,), generator=g))])
            ins = [Instances((300, 420), gt_boxes=Boxes(torch.tensor([b], dtype=torch.float32))) for b in boxes]
            crops = wire.get_crops_qe(img[None].cuda(), ins, "context")
            inst = Instances((300, 420))
            inst.crops = crops
            inst.gt_classes = torch.randint(0, ncls, (n_box,), generator=g)
            out.append([{"image": img, "image_id": seed0 + i, "instances": inst}])
        return out

    shots_in, query_in = items(3, 12, 40, 3), items(2, 10, 60, 3)
    ref_shot_desc = [ovit.vit_forward(sd, ovit.preprocess_crops(d[0]["instances"].crops.cpu(), mean, std)) for d in shots_in]
    ref_query_desc = [ovit.vit_forward(sd, ovit.preprocess_crops(d[0]["instances"].crops.cpu(), mean, std)) for d in query_in]
    shot_cls_list = [d[0]["instances"].gt_classes.clone() for d in shots_in]
    shots = LV.get_descriptors(model, shots_in, mean, std)
    queries = LV.get_descriptors(model, query_in, mean, std)
    assert not shots[0]["instances"].has("crops") and "image" not in shots[0]
    for s, r in zip(shots + queries, ref_shot_desc + ref_query_desc):
        got = s["instances"].crop_feats
        assert got.shape == r.shape and float((got - r).abs().max() / r.abs().max()) <= 1e-4
    cls, desc = LV.assemble_tensors(shots)
    queries = LV.run_nearest_neighbours(cls, desc, queries, cosine=True)
    LV.get_nn_class_confirmatory(queries, 10)
    # oracle chain on the oracle's own descriptors
    rc = torch.cat(shot_cls_list); order = rc.argsort()
    rdesc = torch.cat(ref_shot_desc)[order]; rc = rc[order]
    same_top = same_keep = total = 0
    for q, rq in zip(queries, ref_query_desc):
        top = oknn.dense(rc, rdesc, rq, True)
        keep = (torch.mode(top[:, :10], dim=1)[0] == q["instances"].gt_classes).long()
        same_top += int((q["instances"].top10_shots == top).all(dim=1).sum())
        same_keep += int((q["instances"].keep == keep).sum())
        total += len(rq)
    print("verification chain: %d / %d query rows with identical top-10 class lists, %d / %d identical keep flags" % (same_top, total, same_keep, total))
    assert same_keep == total and same_top >= total - 1


@pytest.mark.parametrize("B,N", [(3, 785), (11, 197), (17, 128), (2, 197)])
def test_qkv_planes_from_the_gemm_epilogue_are_bit_identical(B, N, monkeypatch):
    """`kernels.qkv_attention` (the qkv GEMM writes the attention's fp16 operand planes from its epilogue, csrc/conv_pw_s1.hip PLANES instance;
    no fp32 qkv tensor, no mha_split pass) against the two-launch form `mha(linear(x))`: same arithmetic in the same order, so the attention
    output is bit-identical; a ViT block gives the same result with the fused path on and off; an operand beyond fp16's range raises the
    shared error word."""
    from lvc_amd import kernels as K
    from lvc_amd.modeling.vit import _Block

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + N)
    blk = _Block(384, 6, 4.0, True).to(dev).eval()
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.2))
    x = torch.randn(B * N, 384, generator=g).to(dev)
    a = blk.attn
    pc = a.qkv.packed()
    if B * N < 2048:      # fewer rows than the pointwise kernel takes: the block keeps the two-launch form
        assert not K.can_qkv_planes(x, pc, 6, 64)
        return
    assert K.can_qkv_planes(x, pc, 6, 64)
    with torch.no_grad():
        ref = K.mha(a.qkv(x), B, N, 6, 64, a.scale)
        for _ in range(2):       # the second call reuses the cached planes workspace (padding rows still zero)
            got = K.qkv_attention(x, pc, B, N, 6, a.scale)
            assert torch.equal(got, ref)
        y1 = blk(x, B, N)
        monkeypatch.setattr(K, "QKV_PLANES", False)
        y0 = blk(x, B, N)
        assert torch.equal(y1, y0)
        monkeypatch.setattr(K, "QKV_PLANES", True)
        K.clear_conv_error_word(dev)
        xb = x.clone()
        xb[0] = 3000.0           # inside the GEMM's own operand range (4094), far outside fp16 after the projection
        with torch.no_grad():
            a.qkv.weight.mul_(40.0)
        K.qkv_attention(xb, a.qkv.packed(), B, N, 6, a.scale)
        assert K.conv_error_word(dev) & 2
        K.clear_conv_error_word(dev)
