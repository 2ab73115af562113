"""Parity of the hand-written gfx950 kernels against the CPU oracle, through the C ABI.
GPU-only (-m gpu).  Tolerances: integer/index outputs bit-exact; fp32 conv within 2e-5 relative to
the layer's output scale (different but equally valid fp32 summation order); ROIAlign within 1e-6
relative (same operation order; only the reciprocal/division units differ in principle)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize(
    "N,C,H,W,K,R,stride,pad",
    [
        (2, 64, 50, 84, 64, 1, 1, 0),      # res2 1x1
        (1, 64, 40, 56, 64, 3, 1, 1),      # res2 3x3
        (2, 256, 28, 36, 128, 1, 2, 0),    # stride-in-1x1
        (1, 128, 25, 42, 128, 3, 1, 1),    # odd sizes, M tail
        (1, 256, 13, 21, 256, 3, 1, 1),    # p6-like
        (1, 256, 20, 28, 15, 1, 1, 0),     # rpn predictors: K=15 (N tail)
        (3, 512, 7, 9, 2048, 1, 1, 0),     # wide K
        (1, 2048, 5, 6, 256, 1, 1, 0),     # deep C (lateral5)
    ],
)
def test_conv_igemm_matches_cpu(N, C, H, W, K, R, stride, pad):
    from lvc_amd import kernels as k

    g = torch.Generator().manual_seed(N * 1000 + C + K + R)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, R, R, generator=g) * (2.0 / (C * R * R)) ** 0.5
    b = torch.randn(K, generator=g) * 0.1
    bn = (torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1,
          torch.randn(K, generator=g) * 0.1, torch.rand(K, generator=g) + 0.5)
    ref = F.conv2d(x, w, None, stride=stride, padding=pad)
    scale = bn[0] * (bn[3] + 1e-5).rsqrt()
    ref_bn = F.relu(ref * scale[None, :, None, None] + (bn[1] - bn[2] * scale)[None, :, None, None])
    ref_bias = ref + b[None, :, None, None]
    d = _dev()
    xd = _nhwc(x).to(d)
    pc = k.pack_conv(w.to(d), bn=[t.to(d) for t in bn], stride=stride, pad=pad)
    y = k.conv2d_nhwc(xd, pc, relu=True).cpu().permute(0, 3, 1, 2)
    tol = 2e-5 * float(ref_bn.abs().max())
    assert (y - ref_bn).abs().max() <= tol
    pc2 = k.pack_conv(w.to(d), bias=b.to(d), stride=stride, pad=pad)
    y2 = k.conv2d_nhwc(xd, pc2).cpu().permute(0, 3, 1, 2)
    assert (y2 - ref_bias).abs().max() <= 2e-5 * float(ref_bias.abs().max())


def test_conv_residual_and_upsample_add():
    from lvc_amd import kernels as k

    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 20, 28, generator=g)
    w = torch.randn(256, 64, 1, 1, generator=g) * 0.1
    res = torch.randn(2, 256, 20, 28, generator=g)
    top = torch.randn(2, 256, 10, 14, generator=g)
    d = _dev()
    pc = k.pack_conv(w.to(d))
    ref = F.conv2d(x, w)
    y1 = k.conv2d_nhwc(_nhwc(x).to(d), pc, relu=True, residual=_nhwc(res).to(d), res_mode=1)
    assert (y1.cpu().permute(0, 3, 1, 2) - F.relu(ref + res)).abs().max() < 1e-4
    y2 = k.conv2d_nhwc(_nhwc(x).to(d), pc, residual=_nhwc(top).to(d), res_mode=2)
    up = F.interpolate(top, scale_factor=2, mode="nearest")
    assert (y2.cpu().permute(0, 3, 1, 2) - (ref + up)).abs().max() < 1e-4


@pytest.mark.parametrize(
    "N,C,H,W,K,patch",
    [
        (1, 32, 7, 9, 128, None),        # one partial patch, one channel chunk
        (2, 64, 13, 21, 132, None),      # K tail (132 = 128 + 4)
        (1, 256, 50, 84, 256, None),     # p4-like: 9 x 2 patches of 6 x 42, stream-K splits tiles
        (3, 96, 33, 47, 260, "16,16"),   # square patches, ragged right/bottom edge, 3 N tiles
        (2, 128, 40, 60, 128, "3,70"),   # patch wider than the image
        (1, 64, 64, 64, 128, "10,24"),   # fragment rows straddle patch rows
        (2, 64, 40, 56, 64, None),       # 64-wide N tile (res2 conv2)
        (1, 32, 30, 30, 64, "8,32"),     # 64-wide N tile, one channel chunk
    ],
)
@pytest.mark.parametrize("split", ["bf16x3", "f16x2"])
def test_conv3x3_halo_matches_cpu(N, C, H, W, K, patch, split, monkeypatch):
    """csrc/conv3x3_halo.hip (forced even where the wrapper would pick the generic kernel) against F.conv2d on the
    CPU, with bias, FrozenBN scale/shift, ReLU, residual add and FPN upsample-add epilogues."""
    from lvc_amd import kernels as k

    from lvc_amd import _lib

    ph, pw = [int(v) for v in patch.split(",")] if patch else (0, 0)
    _lib.lib().lvc_set_halo_test_hooks(ph, pw, 1)      # fixed patch shape, no small-map fallback
    try:
        _halo_case(k, N, C, H, W, K, split, monkeypatch)
    finally:
        _lib.lib().lvc_set_halo_test_hooks(0, 0, 0)


def _halo_case(k, N, C, H, W, K, split, monkeypatch):
    monkeypatch.setattr(k, "CONV_ENGINE", "bf16x3")
    monkeypatch.setattr(k, "CONV_HALO", True)
    monkeypatch.setattr(k, "CONV_SPLIT", split)
    monkeypatch.setattr(k, "_HALO_H2_MIN_TILES", 0)
    g = torch.Generator().manual_seed(N * 1000 + C + K + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5
    b = torch.randn(K, generator=g) * 0.1
    bn = (torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1,
          torch.randn(K, generator=g) * 0.1, torch.rand(K, generator=g) + 0.5)
    res = torch.randn(N, K, H, W, generator=g)
    ref = F.conv2d(x, w, None, padding=1)
    scale = bn[0] * (bn[3] + 1e-5).rsqrt()
    ref_bn = F.relu(ref * scale[None, :, None, None] + (bn[1] - bn[2] * scale)[None, :, None, None] + res)
    d = _dev()
    xd = _nhwc(x).to(d)
    timer = k.LaunchTimer()
    monkeypatch.setattr(k, "CONV_TIMER", timer)
    pc = k.pack_conv(w.to(d), bn=[t.to(d) for t in bn], stride=1, pad=1)
    y = k.conv2d_nhwc(xd, pc, relu=True, residual=_nhwc(res).to(d), res_mode=1).cpu().permute(0, 3, 1, 2)
    assert timer.records[-1][3] == split + "_halo"
    assert (y - ref_bn).abs().max() <= 2e-5 * float(ref_bn.abs().max())
    pc2 = k.pack_conv(w.to(d), bias=b.to(d), stride=1, pad=1)
    y2 = k.conv2d_nhwc(xd, pc2).cpu().permute(0, 3, 1, 2)
    ref_bias = ref + b[None, :, None, None]
    assert (y2 - ref_bias).abs().max() <= 2e-5 * float(ref_bias.abs().max())
    # identical to the generic split-precision kernel up to the order in which stream-K partial tiles are added
    monkeypatch.setattr(k, "CONV_HALO", False)
    y3 = k.conv2d_nhwc(xd, pc2).cpu().permute(0, 3, 1, 2)
    assert (y2 - y3).abs().max() <= (4e-6 if split == "bf16x3" else 2e-5) * float(ref_bias.abs().max())
    if H % 2 == 0 and W % 2 == 0:
        monkeypatch.setattr(k, "CONV_HALO", True)
        top = torch.randn(N, K, H // 2, W // 2, generator=g)
        y4 = k.conv2d_nhwc(xd, pc2, residual=_nhwc(top).to(d), res_mode=2).cpu().permute(0, 3, 1, 2)
        ref4 = ref_bias + F.interpolate(top, scale_factor=2, mode="nearest")
        assert (y4 - ref4).abs().max() <= 2e-5 * float(ref4.abs().max())


@pytest.mark.parametrize(
    "N,C,H,W,K,stride,res_mode",
    [
        (2, 64, 40, 56, 256, 1, 1),      # short reduction (2 chunks), residual: conv_pw_bf16x3_kernel
        (2, 128, 30, 44, 512, 1, 1),     # 4 chunks, M tail
        (3, 256, 24, 36, 1024, 1, 1),    # 8 chunks with residual stays on the 128-row pointwise shape
        (3, 256, 24, 36, 512, 2, 0),     # 8 chunks, no residual, stride 2: conv_pw256_bf16x3_kernel (M = 648 < 2048 -> 128-row)
        (4, 256, 48, 64, 512, 2, 0),     # same with M = 3072: 256-row shape
        (2, 512, 40, 52, 128, 1, 0),     # 16 chunks: 256-row shape, M tail (4160 = 16 * 256 + 64)
        (2, 1024, 30, 36, 256, 1, 0),    # 32 chunks: stream-K splits tiles of the 256-row shape
        (2, 512, 40, 52, 256, 1, 2),     # FPN lateral with top-down upsample-add
        (2, 256, 40, 52, 64, 1, 0),      # narrow output: 64-channel tile of the 256-row shape
        (2, 256, 40, 52, 16, 1, 0),      # RPN-predictor width: 32-channel tile
        (8, 256, 24, 12, 128, 1, 2),     # upsample-add on a map narrower than a thread's row step (the stepped row decomposition wraps more than once)
        (2, 256, 40, 52, 64, 1, 1),      # residual on the 64-channel tile of the pipelined kernel
        (3, 128, 30, 44, 256, 1, 2),     # upsample-add below 256 input channels: the two-accumulator epilogue
    ],
)
@pytest.mark.parametrize("split", ["bf16x3", "f16x2"])
def test_pointwise_shapes_match_cpu(N, C, H, W, K, stride, res_mode, split, monkeypatch):
    """1x1 layers through the pointwise shapes of the split-precision kernels (csrc/conv_bf16x3.hip; with split = f16x2 and >= 2048
    output rows the pipelined kernel csrc/conv_pw_s1.hip, whose epilogue has one row loop per residual mode) against F.conv2d."""
    from lvc_amd import kernels as k

    monkeypatch.setattr(k, "CONV_SPLIT", split)
    g = torch.Generator().manual_seed(N * 100 + C + K + stride)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, 1, 1, generator=g) * (2.0 / C) ** 0.5
    b = torch.randn(K, generator=g) * 0.1
    ref = F.conv2d(x, w, b, stride=stride)
    Ho, Wo = ref.shape[2:]
    d = _dev()
    res = None
    if res_mode == 1:
        r = torch.randn(N, K, Ho, Wo, generator=g)
        ref = ref + r
        res = _nhwc(r).to(d)
    elif res_mode == 2:
        r = torch.randn(N, K, Ho // 2, Wo // 2, generator=g)
        ref = ref + F.interpolate(r, scale_factor=2, mode="nearest")
        res = _nhwc(r).to(d)
    ref = F.relu(ref)
    pc = k.pack_conv(w.to(d), bias=b.to(d), stride=stride, pad=0)
    y = k.conv2d_nhwc(_nhwc(x).to(d), pc, relu=True, residual=res, res_mode=res_mode).cpu().permute(0, 3, 1, 2)
    assert (y - ref).abs().max() <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("N,C,H,W,K", [(2, 256, 56, 88, 256), (1, 64, 72, 100, 64), (3, 128, 33, 47, 132)])
def test_conv3x3_pipelined_forms_vs_fp64(N, C, H, W, K, monkeypatch):
    """The three forms of the 3x3 fp16-split forward kernel (kernels.HALO_S1: 0 = round-1 kernel, 1 = pipelined tap loop with
    main + cross accumulators, 2 = pipelined with ONE accumulator and power-of-two operand scaling) against an fp64
    convolution: form 1 equals form 0's accuracy (same arithmetic, different summation order of stream-K partials only), form
    2 stays below the error of the reference's own fp32 CPU convolution (what "fp32-accurate" means for a layer)."""
    from lvc_amd import kernels as k

    monkeypatch.setattr(k, "CONV_ENGINE", "bf16x3")
    monkeypatch.setattr(k, "CONV_SPLIT", "f16x2")
    monkeypatch.setattr(k, "_HALO_H2_MIN_TILES", 0)
    g = torch.Generator().manual_seed(C + K)
    x = torch.randn(N, C, H, W, generator=g).relu_() * 3.0
    w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5
    w[:, :, 1, 1] *= torch.logspace(-3, 1, K)[:, None]          # rows of very different magnitude: the row scaling's case
    res = torch.randn(N, K, H, W, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double(), padding=1) + res.double())
    cpu = F.relu(F.conv2d(x, w, padding=1) + res)
    sc = float(ref.abs().max())
    rms_cpu = float((cpu.double() - ref).pow(2).mean().sqrt()) / sc
    d = _dev()
    pc = k.pack_conv(w.to(d), stride=1, pad=1)
    rms = {}
    for form in (0, 1, 2):
        monkeypatch.setattr(k, "HALO_S1", form)
        timer = k.LaunchTimer()
        monkeypatch.setattr(k, "CONV_TIMER", timer)
        y = k.conv2d_nhwc(_nhwc(x).to(d), pc, relu=True, residual=_nhwc(res).to(d), res_mode=1).cpu().permute(0, 3, 1, 2)
        assert timer.records[-1][3] == "f16x2_halo"
        rms[form] = float((y.double() - ref).pow(2).mean().sqrt()) / sc
        assert float((y.double() - ref).abs().max()) <= 2e-5 * sc, form
    monkeypatch.setattr(k, "CONV_TIMER", None)
    assert k.conv_error_word(d) == 0
    print("rms error / output scale: round-1 %.2e, pipelined two-acc %.2e, pipelined one-acc %.2e, CPU fp32 %.2e" % (rms[0], rms[1], rms[2], rms_cpu))
    assert rms[1] <= 1.05 * rms[0] + 1e-9
    assert rms[2] <= max(rms_cpu, 2.0 * rms[0])
    # the layer flag: `two_acc` keeps form 1 under HALO_S1 = 2 (bit-identical to form 1)
    monkeypatch.setattr(k, "HALO_S1", 1)
    y1 = k.conv2d_nhwc(_nhwc(x).to(d), pc, relu=True)
    monkeypatch.setattr(k, "HALO_S1", 2)
    pc.two_acc = True
    y2 = k.conv2d_nhwc(_nhwc(x).to(d), pc, relu=True)
    assert torch.equal(y1, y2) or float((y1 - y2).abs().max()) <= 2e-6 * sc    # stream-K partial order only


def test_f16x2_range_overflow_is_reported():
    """The two-way fp16 split cannot represent |a| > 65504: the staging code must raise bit 1 of the workspace error
    word (and only then), for the 3x3 halo kernel and for the pointwise kernel."""
    from lvc_amd import kernels as k

    d = _dev()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 40, 52, 256, generator=g).to(d)
    w3 = (torch.randn(128, 256, 3, 3, generator=g) * 0.02).to(d)
    w1 = (torch.randn(128, 256, 1, 1, generator=g) * 0.05).to(d)
    pc3, pc1 = k.pack_conv(w3, stride=1, pad=1), k.pack_conv(w1)
    old_min = k._HALO_H2_MIN_TILES
    k._HALO_H2_MIN_TILES = 0
    try:
        assert k.CONV_SPLIT == "f16x2"
        k.conv2d_nhwc(x, pc3); k.conv2d_nhwc(x, pc1)
        assert k.conv_error_word(x.device) & 2 == 0
        xb = x.clone(); xb[1, 7, 9, 33] = 7.0e4
        k.conv2d_nhwc(xb, pc1)
        assert k.conv_error_word(x.device) & 2 == 2
        k.clear_conv_error_word(x.device)
        assert k.conv_error_word(x.device) == 0
        k.conv2d_nhwc(xb, pc3)
        assert k.conv_error_word(x.device) & 2 == 2
        k.clear_conv_error_word(x.device)
        # the single-accumulator 3x3 form scales activations by 2^4: its range ends at 4094; the two-accumulator form's at 65504
        xc = x.clone(); xc[0, 3, 5, 7] = 5.0e3
        assert k.HALO_S1 == 2 and not pc3.two_acc
        k.conv2d_nhwc(xc, pc3)
        assert k.conv_error_word(x.device) & 2 == 2
        k.clear_conv_error_word(x.device)
        pc3.two_acc = True
        k.conv2d_nhwc(xc, pc3)
        assert k.conv_error_word(x.device) & 2 == 0
    finally:
        k._HALO_H2_MIN_TILES = old_min


def test_conv_stem_7x7():
    from lvc_amd import kernels as k

    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 3, 64, 96, generator=g) * 50
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    ref = F.conv2d(x, w, stride=2, padding=3)
    d = _dev()
    x4 = torch.zeros(2, 64, 96, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    pc = k.pack_conv(w.to(d), stride=2, pad=3, stem=True)
    y = k.conv2d_nhwc(x4.to(d), pc).cpu().permute(0, 3, 1, 2)
    assert (y - ref).abs().max() <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("N,H,W", [(1, 64, 96), (2, 70, 134), (1, 37, 41), (1, 800, 1344)])
def test_stem_conv_pool_fused_matches_cpu(N, H, W, monkeypatch):
    """csrc/stem_pool_h2.hip: conv 7x7/2 + FrozenBN + ReLU + max_pool2d(3, 2, 1) in one launch vs the same chain on the
    CPU (reference BasicStem.forward, resnet.py:588-592), including odd sizes whose patches straddle the borders."""
    from lvc_amd import kernels as k

    monkeypatch.setattr(k, "CONV_SPLIT", "f16x2")
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(N, 3, H, W, generator=g) * 50
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.5,
          torch.randn(64, generator=g) * 0.5, torch.rand(64, generator=g) + 0.5)
    scale = bn[0] * (bn[3] + 1e-5).rsqrt()
    ref = F.conv2d(x, w, stride=2, padding=3)
    ref = F.relu(ref * scale[None, :, None, None] + (bn[1] - bn[2] * scale)[None, :, None, None])
    ref = F.max_pool2d(ref, kernel_size=3, stride=2, padding=1)
    d = _dev()
    x4 = torch.zeros(N, H, W, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    pc = k.pack_conv(w.to(d), bn=[t.to(d) for t in bn], stride=2, pad=3, stem=True)
    y = k.stem_conv_pool(x4.to(d), pc).cpu().permute(0, 3, 1, 2)
    assert y.shape == ref.shape
    assert (y - ref).abs().max() <= 2e-5 * float(ref.abs().max())


def test_linear_matches_cpu():
    from lvc_amd import kernels as k

    g = torch.Generator().manual_seed(7)
    x = torch.randn(300, 1024, generator=g)
    w = torch.randn(81, 1024, generator=g) * 0.03
    b = torch.randn(81, generator=g)
    d = _dev()
    y = k.linear(x.to(d), k.pack_linear(w.to(d), b.to(d)), relu=False).cpu()
    ref = F.linear(x, w, b)
    assert (y - ref).abs().max() <= 2e-5 * float(ref.abs().max())


def _rand_rois(g, n, B, Wimg, Himg, smin=2.0, smax=600.0):
    cx = torch.rand(n, generator=g) * Wimg
    cy = torch.rand(n, generator=g) * Himg
    w = smin + torch.rand(n, generator=g) * (smax - smin)
    h = smin + torch.rand(n, generator=g) * (smax - smin)
    b = torch.randint(0, B, (n,), generator=g).float()
    x1 = (cx - w / 2).clamp(0, Wimg)
    y1 = (cy - h / 2).clamp(0, Himg)
    x2 = (cx + w / 2).clamp(0, Wimg)
    y2 = (cy + h / 2).clamp(0, Himg)
    return torch.stack([b, x1, y1, x2, y2], 1)


@pytest.mark.parametrize("scale,aligned,sr", [(0.25, True, 0), (0.125, True, 0), (1 / 32, True, 0),
                                               (0.0625, False, 2), (0.25, True, 2)])
def test_roi_align_nchw_matches_oracle(scale, aligned, sr):
    from lvc_amd import kernels as k
    from oracle import ops as oops

    g = torch.Generator().manual_seed(11)
    B, C = 2, 40
    H, W = int(800 * scale), int(1344 * scale)
    feat = torch.randn(B, C, H, W, generator=g)
    rois = _rand_rois(g, 200, B, 1333, 800)
    # degenerate / edge RoIs: zero area, full image, out of image (clipped boxes can touch the border)
    extra = torch.tensor([[0, 10, 10, 10, 10], [1, 0, 0, 1333, 800], [0, 1300, 780, 1333, 800],
                          [1, 0, 0, 0.5, 0.5], [0, 5, 5, 5, 300]], dtype=torch.float32)
    rois = torch.cat([rois, extra])
    ref = oops.roi_align_forward(feat, rois, scale, 7, 7, sr, aligned)
    d = _dev()
    out = k.roi_align_forward(feat.to(d), rois.to(d), scale, 7, 7, sr, aligned).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max() <= 1e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("sr,aligned", [(0, True), (2, False), (0, False), (3, True)])
def test_roi_align_fpn_nhwc_matches_oracle(sr, aligned):
    """The engine's all-level NHWC kernel (per-wave sample tables, LDS-staged windows): adaptive grids with hundreds of samples per bin
    (several table passes), fixed grids, both alignment conventions, and the degenerate RoIs of the NCHW test (zero area, the full
    image, boxes touching / beyond the border) on every level."""
    from lvc_amd import kernels as k
    from oracle import ops as oops

    g = torch.Generator().manual_seed(12 + sr)
    B, C = 2, 256
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    feats = [torch.randn(B, C, int(800 * s), int(1344 * s), generator=g) for s in scales]
    rois = _rand_rois(g, 300, B, 1333, 800)
    thin = _rand_rois(g, 60, B, 1333, 800, smin=2.0, smax=12.0)       # a few pixels wide ...
    thin[:, 4] = (thin[:, 2] + 300 + 400 * torch.rand(60, generator=g)).clamp(max=800)     # ... and hundreds tall (the bench's proposals)
    extra = torch.tensor([[0, 10, 10, 10, 10], [1, 0, 0, 1333, 800], [0, 1300, 780, 1333, 800], [1, 1332.5, 799.5, 1333, 800],
                          [0, 0, 0, 0.5, 0.5], [1, 640, 0, 700, 800]], dtype=torch.float32)
    rois = torch.cat([rois, thin, extra.repeat(4, 1)])
    n = rois.shape[0]
    levels = torch.randint(0, 4, (n,), generator=g).int()
    levels[-24:] = torch.arange(4).repeat_interleave(6).int()         # every degenerate RoI once per level
    d = _dev()
    out = k.roi_align_fpn_nhwc([_nhwc(f).to(d) for f in feats], scales, rois.to(d), levels.to(d), 7, 7, sr, aligned)
    out = out.cpu().permute(0, 3, 1, 2)
    for l in range(4):
        sel = (levels == l).nonzero().view(-1)
        ref = oops.roi_align_forward(feats[l], rois[sel], scales[l], 7, 7, sr, aligned)
        assert (out[sel] - ref).abs().max() <= 1e-6 * max(1.0, float(ref.abs().max()))


def test_roi_align_wave_form_at_the_edges_of_its_tables():
    """The engine kernel keeps a row's weights in one register (<= 64 window rows per output row) and a bin's in another (<= 64
    window columns); RoIs just inside and just beyond those limits -- 62 / 63 / 64 / 65 / 66 window columns, 62 .. 66 window rows per
    output row -- must agree with the oracle on either path (separable form / per-sample loop), as must RoIs hanging over the map."""
    from lvc_amd import kernels as k
    from oracle import ops as oops

    g = torch.Generator().manual_seed(5)
    H, W, C = 520, 120, 256
    feat = torch.randn(1, C, H, W, generator=g)
    rois = []
    for wcols in (60.2, 61.2, 62.2, 63.2, 64.2):          # window columns = floor(x1) + 1 - floor(x0) + 1 = wcols + ~2
        rois.append([0, 10.3, 20.0, 10.3 + wcols, 60.0])
        rois.append([0, W - wcols - 0.7, 5.0, W + 3.0, 40.0])         # the same width, hanging over the right edge
    for hrows in (60.2, 61.2, 62.2, 63.2, 64.2):          # window rows per output row = bin height + ~2
        rois.append([0, 30.0, 8.4, 50.0, 8.4 + 7 * hrows])
    rois.append([0, -5.0, -7.0, 20.0, 15.0])
    rois.append([0, 100.0, 500.0, 130.0, 530.0])
    rois = torch.tensor(rois, dtype=torch.float32)
    d = _dev()
    for sr, aligned in ((0, True), (2, True), (0, False)):
        out = k.roi_align_fpn_nhwc([_nhwc(feat).to(d)], [1.0], rois.to(d), None, 7, 7, sr, aligned).cpu().permute(0, 3, 1, 2)
        ref = oops.roi_align_forward(feat, rois, 1.0, 7, 7, sr, aligned)
        assert (out - ref).abs().max() <= 1e-6 * max(1.0, float(ref.abs().max())), (sr, aligned)


def test_roi_align_work_order_is_a_permutation_and_changes_no_value(monkeypatch):
    """lvc_roi_work_order (largest windows first) + lvc_roi_align_fpn_nhwc_ordered: the order is a permutation of the RoIs, its
    window areas fall bucket by bucket, and the pooled features are bit-identical to the launch in RoI order (NaN / degenerate
    boxes included: they land in the smallest bucket)."""
    import ctypes

    from lvc_amd import _lib
    from lvc_amd import kernels as k

    g = torch.Generator().manual_seed(77)
    B, C = 2, 256
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    feats = [torch.randn(B, int(800 * s), int(1344 * s), C, generator=g) for s in scales]
    rois = _rand_rois(g, 3000, B, 1333, 800)
    rois[5, 1:] = torch.tensor([10.0, 10.0, 10.0, 10.0])          # zero area
    rois[6, 1:] = torch.tensor([0.0, 0.0, 1333.0, 800.0])         # the whole image
    levels = torch.randint(0, 4, (3000,), generator=g).int()
    d = _dev()
    fd, rd, ld = [f.to(d) for f in feats], rois.to(d), levels.to(d)
    order = torch.full((3000,), -1, device=d, dtype=torch.int32)
    sc = (ctypes.c_float * 4)(*scales)
    rc = _lib.lib().lvc_roi_work_order(k.ptr(rd), k.ptr(ld), sc, ctypes.c_int(4), ctypes.c_int(3000), ctypes.c_int(7), k.ptr(order), None)
    assert rc == 0
    o = order.cpu().long()
    assert torch.equal(torch.sort(o)[0], torch.arange(3000))
    s_ = torch.tensor(scales)[levels.long()]
    area = ((rois[:, 3] - rois[:, 1]) * s_ + 2) * ((rois[:, 4] - rois[:, 2]) * s_ + 14)
    bucket = (2 * torch.log2(area)).floor()[o]
    assert (bucket[1:] <= bucket[:-1]).all() and o[0] != 5 and o[-1] != 6
    # the XCD-local order (what roi_align_fpn_nhwc uses for <= 16 images): a permutation; position i holds a RoI of image i % B while
    # every image has RoIs left, and inside an image the (level, 16-pixel band of the centre row) keys do not decrease
    order2 = torch.full((3000,), -1, device=d, dtype=torch.int32)
    assert _lib.lib().lvc_roi_work_order_xcd(k.ptr(rd), k.ptr(ld), ctypes.c_int(3000), ctypes.c_int(B), k.ptr(order2), None) == 0
    o2 = order2.cpu().long()
    assert torch.equal(torch.sort(o2)[0], torch.arange(3000))
    imgs = rois[:, 0].long()
    m = int(torch.bincount(imgs, minlength=B).min())
    assert torch.equal(imgs[o2[:m * B]], torch.arange(B).repeat(m))
    key = levels.long() * 64 + ((rois[:, 2] + rois[:, 4]) * 0.5 / 16).long().clamp(max=63)
    for b in range(B):
        kb = key[o2][imgs[o2] == b]
        assert (kb[1:] >= kb[:-1]).all()
    monkeypatch.setattr(k, "ROI_ORDER_MIN", 1 << 30)
    plain = k.roi_align_fpn_nhwc(fd, scales, rd, ld, 7, 7, 0, True)
    monkeypatch.setattr(k, "ROI_ORDER_MIN", 1)
    ordered = k.roi_align_fpn_nhwc(fd, scales, rd, ld, 7, 7, 0, True)
    assert torch.equal(plain, ordered)


@pytest.mark.parametrize("scale,aligned,sr", [(0.25, True, 0), (1 / 32, True, 0), (0.0625, False, 2)])
def test_roi_align_backward_nchw_matches_oracle(scale, aligned, sr):
    """lvc_roi_align_backward_nchw (atomic scatter, like ROIAlign_cuda.cu) vs the oracle's restatement of the
    reference CPU kernel: equal up to fp32 summation order -> 1e-5 of the largest accumulated gradient."""
    from lvc_amd import kernels as k
    from oracle import ops as oops

    g = torch.Generator().manual_seed(21)
    B, C = 2, 40
    H, W = int(800 * scale), int(1344 * scale)
    rois = _rand_rois(g, 200, B, 1333, 800)
    extra = torch.tensor([[0, 10, 10, 10, 10], [1, 0, 0, 1333, 800], [0, 1300, 780, 1333, 800],
                          [1, 0, 0, 0.5, 0.5], [0, 5, 5, 5, 300]], dtype=torch.float32)
    rois = torch.cat([rois, extra])
    grad = torch.randn(rois.shape[0], C, 7, 7, generator=g)
    ref = oops.roi_align_backward(grad, rois, scale, 7, 7, B, C, H, W, sr, aligned)
    d = _dev()
    out = k.roi_align_backward(grad.to(d), rois.to(d), scale, 7, 7, B, C, H, W, sr, aligned).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max()))


def test_roi_align_backward_golden_and_autograd():
    """The reference's own `_C.roi_align_backward` outputs (tests/golden/roi_align.npz, bwd_*) and the autograd
    wiring of layers.ROIAlign (reference roi_align.py:22-57): d(sum(out * g))/d(input) == roi_align_backward(g)."""
    from helpers import gold
    from lvc_amd import kernels as k
    from lvc_amd.layers import ROIAlign

    gd = gold("roi_align")
    d = _dev()
    for key in [kk for kk in gd if kk.startswith("bwd_")]:
        _, s, a, sr = key.split("_")
        out = k.roi_align_backward(gd["grad"].to(d), gd["rois"].to(d), float(s[1:]), 7, 7, 2, 8, 50, 84,
                                   int(sr[2:]), bool(int(a[1:]))).cpu()
        assert (out - gd[key]).abs().max() <= 1e-5 * max(1.0, float(gd[key].abs().max())), key
    feat = gd["feat"].to(d).requires_grad_(True)
    y = ROIAlign((7, 7), 0.25, 0, True)(feat, gd["rois"].to(d))
    (y * gd["grad"].to(d)).sum().backward()
    assert (feat.grad.cpu() - gd["bwd_s0.25_a1_sr0"]).abs().max() <= 1e-5 * float(gd["bwd_s0.25_a1_sr0"].abs().max())


def test_roi_align_fpn_backward_nhwc_matches_oracle():
    from lvc_amd import kernels as k
    from oracle import ops as oops

    g = torch.Generator().manual_seed(22)
    B, C = 2, 256
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    shapes = [(B, int(800 * s), int(1344 * s), C) for s in scales]
    rois = _rand_rois(g, 300, B, 1333, 800)
    levels = torch.randint(0, 4, (300,), generator=g).int()
    grad = torch.randn(300, 7, 7, C, generator=g)
    d = _dev()
    outs = k.roi_align_fpn_backward_nhwc(grad.to(d), shapes, scales, rois.to(d), levels.to(d), 0, True)
    for l in range(4):
        sel = (levels == l).nonzero().view(-1)
        ref = oops.roi_align_backward(grad[sel].permute(0, 3, 1, 2), rois[sel], scales[l], 7, 7, B, C,
                                      shapes[l][1], shapes[l][2], 0, True)
        got = outs[l].cpu().permute(0, 3, 1, 2)
        assert (got - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("sr", [0, 2])
def test_roi_align_fpn_backward_edge_rois(sr):
    """The separable backward (weight tables per axis, one atomic per footprint pixel) on the RoIs that stress its
    bookkeeping: boxes hanging over every edge of the map (dropped samples, clamped rows), degenerate and one-pixel boxes,
    boxes covering the whole map on the coarsest level, and RoIs too large for the tables (scatter path)."""
    from lvc_amd import kernels as k
    from oracle import ops as oops

    B, C = 2, 64
    scales = [1 / 4, 1 / 32]
    shapes = [(B, 60, 84, C), (B, 8, 11, C)]
    rois = torch.tensor([
        [0, -40.0, -30.0, 50.0, 60.0], [1, 300.0, 200.0, 400.0, 300.0], [0, 330.0, 10.0, 336.0, 239.0],
        [1, 10.0, 10.0, 10.0, 10.0], [0, 100.0, 100.0, 101.0, 101.5], [1, -500.0, -500.0, -300.0, -300.0],
        [0, 0.0, 0.0, 336.0, 240.0], [1, 5.0, 5.0, 330.0, 20.0], [0, 2.0, 3.0, 9.0, 230.0], [1, 0.0, 0.0, 336.0, 240.0],
    ])
    levels = torch.tensor([0, 0, 0, 0, 0, 0, 1, 0, 0, 0], dtype=torch.int32)
    g = torch.Generator().manual_seed(5)
    grad = torch.randn(rois.shape[0], 7, 7, C, generator=g)
    d = _dev()
    outs = k.roi_align_fpn_backward_nhwc(grad.to(d), shapes, scales, rois.to(d), levels.to(d), sr, True)
    for l in range(2):
        sel = (levels == l).nonzero().view(-1)
        ref = oops.roi_align_backward(grad[sel].permute(0, 3, 1, 2), rois[sel], scales[l], 7, 7, B, C,
                                      shapes[l][1], shapes[l][2], sr, True)
        got = outs[l].cpu().permute(0, 3, 1, 2)
        assert (got - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max())), l


def _nms_case(g, n, nidx, jitter):
    base = _rand_rois(g, max(4, n // 6), 1, 1333, 800, 8, 300)[:, 1:]
    pick = torch.randint(0, base.shape[0], (n,), generator=g)
    boxes = base[pick] + torch.randn(n, 4, generator=g) * jitter
    boxes[:, 2:] = torch.max(boxes[:, 2:], boxes[:, :2] + 0.5)
    scores = torch.randn(n, generator=g)
    idxs = torch.randint(0, nidx, (n,), generator=g)
    return boxes, scores, idxs


@pytest.mark.parametrize("n,nidx,thr", [(1, 1, 0.5), (63, 2, 0.7), (64, 1, 0.3), (65, 5, 0.5),
                                         (1000, 80, 0.5), (4819, 5, 0.7), (9000, 3, 0.7)])
def test_batched_nms_keep_indices_bit_exact(n, nidx, thr):
    from lvc_amd import kernels as k
    from oracle import ops as oops

    g = torch.Generator().manual_seed(n)
    boxes, scores, idxs = _nms_case(g, n, nidx, 4.0)
    scores[::7] = scores[0]  # exact score ties must resolve by index like the stable CPU sort
    ref = oops.batched_nms(boxes, scores, idxs, thr)
    d = _dev()
    got = k.batched_nms(boxes.to(d), scores.to(d), idxs.to(d), thr).cpu()
    assert got.dtype == torch.int64
    assert got.tolist() == ref.tolist()


def test_batched_nms_batch_with_counts_and_max_keep():
    from lvc_amd import kernels as k
    from oracle import ops as oops

    g = torch.Generator().manual_seed(99)
    B, Nmax = 3, 2000
    counts = torch.tensor([2000, 777, 0], dtype=torch.int32)
    boxes = torch.zeros(B, Nmax, 4)
    scores = torch.zeros(B, Nmax)
    idxs = torch.zeros(B, Nmax, dtype=torch.int32)
    for b in range(B):
        bb, ss, ii = _nms_case(g, Nmax, 5, 3.0)
        boxes[b], scores[b], idxs[b] = bb, ss, ii.int()
    d = _dev()
    keep, nk = k.batched_nms_batch(boxes.to(d), scores.to(d), idxs.to(d), counts.to(d), 0.7, max_keep=300)
    keep, nk = keep.cpu(), nk.cpu()
    for b in range(B):
        n = int(counts[b])
        ref = oops.batched_nms(boxes[b, :n], scores[b, :n], idxs[b, :n].long(), 0.7)[:300]
        assert int(nk[b]) == len(ref)
        assert keep[b, : len(ref)].tolist() == ref.tolist()


@pytest.mark.parametrize("thr,max_keep", [(0.7, 0), (0.5, 100), (0.3, 37)])
def test_nms_reduce_from_lds_equals_the_global_form_and_the_oracle(monkeypatch, thr, max_keep):
    """Lists of <= 1024 rows (the RPN's per-level lists) run the ordered reduce with the bit matrix in LDS (nms_reduce_lds_kernel);
    `lvc_set_nms_reduce_global(1)` (LVC_NMS_REDUCE_GLOBAL=1 at import) keeps them on nms_reduce_kernel: same keep lists, equal to the oracle's, for ragged counts around the
    64-row chunk edges, with and without a cap on the kept boxes."""
    from lvc_amd import kernels as k
    from oracle import ops as oops

    g = torch.Generator().manual_seed(int(thr * 100) + max_keep)
    counts = torch.tensor([1000, 999, 961, 960, 513, 129, 128, 65, 64, 63, 2, 1, 0], dtype=torch.int32)
    B, Nmax = len(counts), 1000
    boxes = torch.zeros(B, Nmax, 4)
    scores = torch.zeros(B, Nmax)
    idxs = torch.zeros(B, Nmax, dtype=torch.int32)
    for b in range(B):
        bb, ss, ii = _nms_case(g, Nmax, 1 + b % 3, 2.0 + b)
        ss[::5] = ss[0]
        boxes[b], scores[b], idxs[b] = bb, ss, ii.int()
    d = _dev()
    res = {}
    for glob in (False, True):
        from lvc_amd import _lib

        _lib.lib().lvc_set_nms_reduce_global(1 if glob else 0)
        try:
            keep, nk = k.batched_nms_batch(boxes.to(d), scores.to(d), idxs.to(d), counts.to(d), thr, max_keep=max_keep)
        finally:
            _lib.lib().lvc_set_nms_reduce_global(0)
        res[glob] = (keep.cpu(), nk.cpu())
    assert res[False][1].tolist() == res[True][1].tolist()
    for b in range(B):
        n = int(counts[b])
        ref = oops.batched_nms(boxes[b, :n], scores[b, :n], idxs[b, :n].long(), thr)
        if max_keep:
            ref = ref[:max_keep]
        for keep, nk in res.values():
            assert int(nk[b]) == len(ref)
            assert keep[b, : len(ref)].tolist() == ref.tolist()


def test_nms_empty():
    from lvc_amd import kernels as k

    d = _dev()
    out = k.nms(torch.zeros(0, 4, device=d), torch.zeros(0, device=d), 0.5)
    assert out.numel() == 0 and out.dtype == torch.int64


def test_c_abi_argument_errors_raise_runtime_error():
    """Bad shapes come back as status 1 + message -> RuntimeError (reference: AT_ASSERTM -> RuntimeError)."""
    from lvc_amd import kernels as k
    from lvc_amd._lib import LvcNativeError

    d = _dev()
    w = torch.randn(64, 48, 1, 1, device=d)
    with pytest.raises(AssertionError):
        k.pack_conv(w)                       # in_channels % 32 != 0 is rejected while packing
    pc = k.pack_conv(torch.randn(64, 64, 1, 1, device=d))
    with pytest.raises(AssertionError):
        k.conv2d_nhwc(torch.randn(1, 4, 4, 32, device=d), pc)   # channel mismatch
    with pytest.raises(LvcNativeError, match="k must be in 1..10"):
        k.knn_topk_vote(torch.zeros(4, 64, device=d), 64, torch.zeros(64, dtype=torch.int64, device=d), None, 11)
    assert issubclass(LvcNativeError, RuntimeError)


def test_roi_align_negative_size_sets_status():
    from lvc_amd import kernels as k
    from lvc_amd.modeling.roi_heads.roi_heads import check_status

    d = _dev()
    st = k.new_status(d)
    k.roi_align_forward(torch.zeros(1, 4, 8, 8, device=d), torch.tensor([[0, 5, 5, 2, 2.0]], device=d), 1.0, 7, 7, 0, True, status=st)
    with pytest.raises(RuntimeError, match="non-negative size"):
        check_status(int(st.item()))


@pytest.mark.gpu
def test_pack_split_in_one_launch_equals_pack_then_split():
    """lvc_pack_split_conv_weights (one launch) against lvc_pack_conv_weights + lvc_split_weights: same packed operand and
    the same fp16 / bf16 planes bit for bit, forward (mode 0) and data-gradient (mode 1, flipped / transposed / scaled)."""
    from lvc_amd import kernels as Kn
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(96, 64, 3, 3, generator=g) * 0.05).to(dev)
    scale = (torch.rand(96, generator=g) + 0.5).to(dev)
    for mode, sc, rows, cin in ((0, None, 128, 64), (1, scale, 128, 96)):
        wp0, none = Kn._pack_weights(w, sc, rows, cin, mode)
        assert none is None
        for planes, dt in ((2, torch.float16), (3, torch.bfloat16)):
            wp1, pl = Kn._pack_weights(w, sc, rows, cin, mode, planes)
            assert torch.equal(wp0, wp1) and pl.dtype == dt
            pc = Kn.PackedConv(wp0, None, None, rows, cin, 3, 3, 1, 1, wp0.shape[1], 0)
            ref = pc._split(planes)
            assert torch.equal(pl.view(torch.int16), ref.view(torch.int16))
    Kn.check_conv_error_word(dev)


@pytest.mark.gpu
def test_fused_projection_block_matches_two_launch_form(monkeypatch):
    """res2.0 in inference: conv3 + projection shortcut as one GEMM over [conv2 output | block input] (FrozenBN scales
    folded into the concatenated weights, the stem writing its output into the buffer's tail) against the two-launch
    form of the reference graph (resnet.py:195-211); both against an fp64 torch evaluation of the same modules."""
    from lvc_amd import kernels as Kn
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn
    dev = torch.device("cuda:0")
    model = build_model(base_rcnn_fpn()).eval()
    syn.conditioned_r50_fpn_(model)
    body = model.backbone.bottom_up
    blk = body.stages_and_names[0][0][0]
    x = torch.randn(2, 3, 96, 160, generator=torch.Generator().manual_seed(2)).to(dev) * 50
    outs = {}
    with torch.no_grad():
        for fused in (True, False):
            monkeypatch.setattr(Kn, "FUSE_PROJECTION", fused)
            assert blk.can_fuse_projection() == fused
            outs[fused] = body(x)["res2"].double().cpu()
        # fp64 evaluation of stem + res2 with torch ops on the same parameters
        def bn(t, n):
            s = n.weight.double() * (n.running_var.double() + n.eps).rsqrt()
            return t * s.view(1, -1, 1, 1) + (n.bias.double() - n.running_mean.double() * s).view(1, -1, 1, 1)
        def cv(t, c, relu):
            y = bn(F.conv2d(t, c.weight.double(), stride=c.stride, padding=c.padding), c.norm)
            return y.clamp_min(0) if relu else y
        t = F.max_pool2d(cv(x.double(), body.stem.conv1, True), 3, 2, 1)
        for b in body.stages_and_names[0][0]:
            sc = cv(t, b.shortcut, False) if b.shortcut is not None else t
            t = (cv(cv(cv(t, b.conv1, True), b.conv2, True), b.conv3, False) + sc).clamp_min(0)
        ref = t.cpu()
    scale = ref.abs().max().item()
    e_f = (outs[True] - ref).abs().max().item() / scale
    e_u = (outs[False] - ref).abs().max().item() / scale
    assert e_f < 2e-6 and e_u < 2e-6, (e_f, e_u)
    assert (outs[True] - outs[False]).abs().max().item() / scale < 2e-6


@pytest.mark.parametrize("h,w,dtype", [(37, 61, "f32"), (800, 1333, "u8"), (224, 224, "f32"), (33, 64, "u8")])
def test_preprocess_kernel_bit_exact(h, w, dtype):
    """GeneralizedRCNN.preprocess_image + ImageList padding in isolation (reference lvc/modeling/meta_arch/rcnn.py:324-333,
    structures/image_list.py:95-119): (x - mean) / std per channel, zero padding, a zero 4th channel slot -- bit for bit
    against the oracle's torch-CPU evaluation of the same two fp32 operations."""
    from lvc_amd import kernels as k
    from oracle import rcnn as orc

    g = torch.Generator().manual_seed(h * 1000 + w)
    img = torch.rand(3, h, w, generator=g) * 255.0
    if dtype == "u8":
        img = img.round().to(torch.uint8)
    for mean, std in (((103.53, 116.28, 123.675), (1.0, 1.0, 1.0)), ((123.675, 116.28, 103.53), (58.395, 57.12, 57.375))):
        Hp, Wp = (h + 31) // 32 * 32, (w + 31) // 32 * 32
        buf = torch.full((Hp, Wp, 4), 7.0, device=_dev())
        k.preprocess_into(img.to(_dev()), buf, mean, std)
        ref, sizes = orc.preprocess([img], mean, std, 32)
        assert sizes == [(h, w)] and tuple(ref.shape) == (1, 3, Hp, Wp)
        got = buf.cpu()
        assert torch.equal(got[..., :3].permute(2, 0, 1), ref[0])
        assert float(got[..., 3].abs().max()) == 0.0


def test_preprocess_batch_equals_per_image():
    """lvc_preprocess_batch_nhwc4 (one launch for the batch) against lvc_preprocess_nhwc4 image by image: bit-identical, for
    ragged sizes, float32 and uint8 inputs, and more than 16 images (two launches)."""
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(4)
    mean, std = [103.53, 116.28, 123.675], [1.0, 57.375, 58.395]
    for dtype in (torch.float32, torch.uint8):
        sizes = [(37, 50), (64, 64), (1, 3), (50, 61)] * 5
        imgs = [(torch.rand(3, h, w, generator=g) * 255).to(dtype).to("cuda:0") for h, w in sizes]
        Hp, Wp = 64, 96
        a = torch.full((len(imgs), Hp, Wp, 4), 7.0, device="cuda:0")
        b = torch.full((len(imgs), Hp, Wp, 4), 9.0, device="cuda:0")
        K.preprocess_batch_into(imgs, a, mean, std)
        for i, im in enumerate(imgs):
            K.preprocess_into(im, b[i], mean, std)
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,H,W,C,K,stride,act", [(2, 50, 84, 1024, 256, 1, "relu"), (1, 25, 42, 2048, 512, 1, None), (2, 57, 83, 512, 1024, 2, "relu"),
                                                  (300, 1, 1, 12544, 1024, 1, "relu"), (5000, 1, 1, 384, 1152, 1, "gelu"), (3, 40, 52, 256, 64, 1, "relu"),
                                                  (1, 31, 33, 288, 132, 1, None)])
def test_pointwise_pipelined_forms_vs_fp64(N, H, W, C, K, stride, act, monkeypatch):
    """csrc/conv_pw_s1.hip (residual-free pointwise layers with >= 256 input channels; kernels.PW_S1: 1 = two accumulators,
    2 = one) against fp64: form 1 is BIT-IDENTICAL to the LDS-DMA kernel (the same arithmetic in the same order), form 2 stays
    at the error level of an fp32 CPU evaluation; strides, ragged row / channel tails, the fused ReLU / GELU epilogues."""
    from lvc_amd import kernels as k

    monkeypatch.setattr(k, "CONV_ENGINE", "bf16x3")
    monkeypatch.setattr(k, "CONV_SPLIT", "f16x2")
    g = torch.Generator().manual_seed(C + K + H)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(K, C, 1, 1, generator=g) * (2.0 / C) ** 0.5
    b = torch.randn(K, generator=g) * 0.1
    xs = x[:, ::stride, ::stride]
    z = xs.double() @ w[:, :, 0, 0].double().t() + b.double()
    z32 = xs @ w[:, :, 0, 0].t() + b
    fn = {"relu": torch.relu, "gelu": torch.nn.functional.gelu, None: lambda t: t}[act]
    ref, cpu = fn(z), fn(z32)
    sc = float(ref.abs().max())
    rms_cpu = float((cpu.double() - ref).pow(2).mean().sqrt()) / sc
    d = _dev()
    pc = k.pack_conv(w.to(d), bias=b.to(d), stride=stride)
    out = {}
    for form in (0, 1, 2):
        monkeypatch.setattr(k, "PW_S1", form)
        y = k.conv2d_nhwc(x.to(d), pc, relu=act == "relu", act="gelu" if act == "gelu" else None).cpu()
        out[form] = y
        assert float((y.double() - ref).abs().max()) <= 2e-5 * sc, form
    assert k.conv_error_word(d) == 0
    rms = {f: float((out[f].double() - ref).pow(2).mean().sqrt()) / sc for f in out}
    print("pointwise %s C=%d K=%d: rms error / scale  LDS-DMA %.2e, pipelined two-acc %.2e, one-acc %.2e, CPU fp32 %.2e" % ((N, H, W), C, K, rms[0], rms[1], rms[2], rms_cpu))
    if N * ((H - 1) // stride + 1) * ((W - 1) // stride + 1) >= 2048 and C >= 64:     # both on a two-way fp16 kernel: same bits
        assert torch.equal(out[0], out[1]) or float((out[0] - out[1]).abs().max()) <= 2e-6 * sc
    assert rms[2] <= max(1.5 * rms_cpu, 2.0 * rms[0])


@pytest.mark.parametrize("N,H,W,C,K,stride,act,res_mode", [(2, 50, 84, 256, 1024, 1, "relu", 1), (3, 26, 42, 512, 256, 1, None, 2), (9000, 1, 1, 4112, 512, 1, "relu", 0),
                                                           (2, 57, 83, 512, 256, 2, "gelu", 0), (1, 64, 100, 1024, 256, 1, "relu", 0)])
def test_pointwise_256x256_tile_vs_fp64_and_the_256x128_tile(N, H, W, C, K, stride, act, res_mode, monkeypatch):
    """csrc/conv_pw_w2.hip (kernels.PW_W2: single-accumulator pointwise layers on the 256 x 256 workgroup tile) against fp64 and
    against the 256 x 128 tile of conv_pw_s1.hip it replaces: same operands, another fp32 summation order across the stream-K
    hand-offs.  Residual add, the nearest-x2 upsample-add of the FPN laterals, stride 2, row counts that are no multiple of 256, a
    contraction with an odd number of 32-channel chunks, ReLU / GELU."""
    from lvc_amd import kernels as k

    monkeypatch.setattr(k, "CONV_ENGINE", "bf16x3")
    monkeypatch.setattr(k, "CONV_SPLIT", "f16x2")
    monkeypatch.setattr(k, "PW_S1", 2)
    monkeypatch.setattr(k, "_PW_W2_MIN_C", 256)
    monkeypatch.setattr(k, "_PW_W2_MIN_ROWS", 1)
    g = torch.Generator().manual_seed(C + K + H)
    Cp = (C + 31) // 32 * 32          # packed layers carry C % 32 == 0; the tail channels beyond C are zero in x and w
    x = torch.zeros(N, H, W, Cp)
    x[..., :C] = torch.randn(N, H, W, C, generator=g)
    w = torch.zeros(K, Cp, 1, 1)
    w[:, :C] = torch.randn(K, C, 1, 1, generator=g) * (2.0 / C) ** 0.5
    b = torch.randn(K, generator=g) * 0.1
    xs = x[:, ::stride, ::stride]
    Ho, Wo = xs.shape[1], xs.shape[2]
    z = xs.double() @ w[:, :, 0, 0].double().t() + b.double()
    res = None
    if res_mode == 1:
        res = torch.randn(N, Ho, Wo, K, generator=g)
        z = z + res.double()
    elif res_mode == 2:
        res = torch.randn(N, Ho // 2, Wo // 2, K, generator=g)
        z = z + res.double().repeat_interleave(2, 1).repeat_interleave(2, 2)
    fn = {"relu": torch.relu, "gelu": torch.nn.functional.gelu, None: lambda t: t}[act]
    ref = fn(z)
    sc = float(ref.abs().max())
    d = _dev()
    pc = k.pack_conv(w.to(d), bias=b.to(d), stride=stride)
    out, tags = {}, {}
    for w2 in (False, True):
        monkeypatch.setattr(k, "PW_W2", w2)
        y = k.conv2d_nhwc(x.to(d), pc, relu=act == "relu", act="gelu" if act == "gelu" else None, residual=res.to(d) if res is not None else None,
                          res_mode=res_mode).cpu()
        assert float((y.double() - ref).abs().max()) <= 2e-5 * sc, w2
        out[w2] = y
    assert k.conv_error_word(d) == 0
    rms = {f: float((out[f].double() - ref).pow(2).mean().sqrt()) / sc for f in out}
    print("pointwise %s C=%d K=%d: rms error / scale 256x128 tile %.2e, 256x256 tile %.2e" % ((N, H, W), C, K, rms[False], rms[True]))
    assert not torch.equal(out[False], out[True]) or C <= 32      # the other kernel did run
    assert rms[True] <= 1.5 * rms[False] + 1e-9
    # range word: an activation beyond the single-accumulator form's 4094 raises the layer's word on this kernel too
    monkeypatch.setattr(k, "PW_W2", True)
    xb = x.clone()
    xb[0, 0, 0, 0] = 5000.0
    k.conv2d_nhwc(xb.to(d), pc)
    assert k.conv_error_word(d) & 2
    k.clear_conv_error_word(d)


@pytest.mark.parametrize("N,H,W,C,K,relu,bias", [(2, 100, 168, 256, 256, True, True), (1, 37, 53, 64, 192, False, True), (3, 16, 33, 128, 128, True, False),
                                                 (1, 8, 32, 32, 320, True, True), (2, 21, 7, 96, 128, False, False)])
@pytest.mark.parametrize("streamk", [0, 1, 2])
def test_conv3x3_winograd_vs_fp64_and_the_direct_kernel(N, H, W, C, K, relu, bias, streamk, monkeypatch):
    """csrc/conv3x3_wino.hip (kernels.CONV_WINO: Winograd F(2,3) along x, 6 products per output) against the fp64 convolution and the
    direct single-accumulator kernel it replaces on large maps: map sizes that are no multiple of the 8 x 32 / 16 x 16 patch, odd
    widths (the last pair's second pixel does not exist), channel counts off the 128-channel tile, with and without bias / ReLU.  Its
    error must stay at the direct kernel's level (measured: 0.7 x; scripts/winograd_error.py has the fp32 argument).  streamk: one
    workgroup per tile (0, the default), the (tile, chunk) list split evenly over the CUs (1: split tiles completed through the
    workspace, run-to-run identical) or persistent workgroups on whole tiles (2)."""
    from lvc_amd import kernels as k

    prev = k.WINO_STREAMK
    k.set_wino_streamk(streamk)
    try:
        _winograd_case(k, N, H, W, C, K, relu, bias, monkeypatch)
    finally:
        k.set_wino_streamk(prev)


def _winograd_case(k, N, H, W, C, K, relu, bias, monkeypatch):

    monkeypatch.setattr(k, "CONV_ENGINE", "bf16x3")
    monkeypatch.setattr(k, "CONV_SPLIT", "f16x2")
    monkeypatch.setattr(k, "_HALO_H2_MIN_TILES", 0)
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(N, H, W, C, generator=g).relu_()
    w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K, generator=g) * 0.1 if bias else None
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double() if bias else None, padding=1).permute(0, 2, 3, 1)
    if relu:
        ref = ref.relu()
    sc = float(ref.abs().max())
    d = _dev()
    pc = k.pack_conv(w.to(d), bias=b.to(d) if bias else None, pad=1)
    monkeypatch.setattr(k, "CONV_WINO", False)
    direct = k.conv2d_nhwc(x.to(d), pc, relu=relu).cpu()
    got = k.conv3x3_wino(x.to(d), pc, relu=relu).cpu()
    assert k.conv_error_word(d) == 0
    assert torch.equal(k.conv3x3_wino(x.to(d), pc, relu=relu).cpu(), got)       # split tiles are summed in worker order: deterministic
    e_w = (got.double() - ref).abs()
    e_d = (direct.double() - ref).abs()
    rms_w, rms_d = float(e_w.pow(2).mean().sqrt()) / sc, float(e_d.pow(2).mean().sqrt()) / sc
    print("3x3 %s C=%d K=%d: rms error / scale Winograd %.2e, direct %.2e; max %.2e / %.2e" % ((N, H, W), C, K, rms_w, rms_d, float(e_w.max()) / sc, float(e_d.max()) / sc))
    cpu = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1)
    if relu:
        cpu = cpu.relu()
    rms_cpu = float((cpu.double() - ref).pow(2).mean().sqrt()) / sc
    print("   the reference's own fp32 CPU convolution: %.2e" % rms_cpu)
    assert float(e_w.max()) <= 2e-5 * sc
    # at the level of the kernel it replaces, or of the reference's fp32 evaluation (small maps run the direct case on the bf16x3 kernels,
    # which are closer to fp64 than either)
    assert rms_w <= max(1.25 * rms_d, 1.5 * rms_cpu) + 1e-9
    # routed automatically for launches with enough tiles (and only then), same result
    monkeypatch.setattr(k, "CONV_WINO", True)
    monkeypatch.setattr(k, "_WINO_MIN_TILES", 1)
    auto = k.conv2d_nhwc(x.to(d), pc, relu=relu).cpu()
    assert torch.equal(auto, got) if K >= 128 else True
    # range word: a window value beyond the single-accumulator form's 4094 raises the layer's word
    xb = x.clone()
    xe = (W // 2) & ~1                                         # the two pixels of ONE pair: V1 = d1 + d2 = 6000
    xb[0, H // 2, xe, 0] = 3000.0
    xb[0, H // 2, xe + 1, 0] = 3000.0
    k.conv3x3_wino(xb.to(d), pc)
    assert k.conv_error_word(d) & 2
    k.clear_conv_error_word(d)


@pytest.mark.parametrize("shape", ["res2", "res2.0", "res3"])
def test_chained_conv3_conv1_matches_two_launches_and_fp64(shape):
    """csrc/conv_pw_chain.hip (kernels.CHAIN / LVC_CHAIN): a bottleneck's conv3 + FrozenBN + shortcut add + ReLU and the next block's
    conv1 + FrozenBN + ReLU as one launch (reference resnet.py:205-211 then :195-197) against the two launches and an fp64
    evaluation; a row count that is not a multiple of the 128-pixel workgroup, a row stride wider than the contraction (the
    res2.0 concat buffer), and the range word for an input beyond 4094."""
    from lvc_amd import kernels as k

    K1, N1, N2, res, ldx = {"res2": (64, 256, 64, True, 64), "res2.0": (128, 256, 64, False, 128), "res3": (128, 512, 128, True, 160)}[shape]
    M = 128 * 37 + 45
    g = torch.Generator().manual_seed(K1 + N1)
    x = torch.randn(M, 1, 1, ldx, generator=g).relu_()
    r = torch.randn(M, 1, 1, N1, generator=g).relu_() if res else None
    wa = torch.randn(N1, K1, 1, 1, generator=g) * (2.0 / K1) ** 0.5
    wb = torch.randn(N2, N1, 1, 1, generator=g) * (2.0 / N1) ** 0.5
    bna = (torch.rand(N1, generator=g) + 0.5, torch.randn(N1, generator=g) * 0.1, torch.randn(N1, generator=g) * 0.1, torch.rand(N1, generator=g) + 0.5)
    bnb = (torch.rand(N2, generator=g) + 0.5, torch.randn(N2, generator=g) * 0.1, torch.randn(N2, generator=g) * 0.1, torch.rand(N2, generator=g) + 0.5)
    d = _dev()
    pa, pb = k.pack_conv(wa.to(d), bn=[t.to(d) for t in bna]), k.pack_conv(wb.to(d), bn=[t.to(d) for t in bnb])
    ch = k.pack_chain(pa, pb)
    xd, rd = x.to(d), (r.to(d) if res else None)
    y1, y2 = k.conv1x1_chain(xd, ch, residual=rd)
    z1 = k.conv2d_nhwc(xd[..., :K1].contiguous(), pa, relu=True, residual=rd, res_mode=1 if res else 0)
    z2 = k.conv2d_nhwc(z1, pb, relu=True)
    sa, ta = k.conv_affine(None, bna)
    sb, tb = k.conv_affine(None, bnb)
    r1 = x.view(M, ldx)[:, :K1].double() @ wa.view(N1, K1).double().t() * sa.double() + ta.double()
    if res:
        r1 = r1 + r.view(M, N1).double()
    r1 = r1.relu()
    r2 = (r1 @ wb.view(N2, N1).double().t() * sb.double() + tb.double()).relu()
    for got, two, ref in ((y1, z1, r1), (y2, z2, r2)):
        scale = float(ref.abs().max())
        e_one = float((got.cpu().view(M, -1).double() - ref).abs().max()) / scale
        e_two = float((two.cpu().view(M, -1).double() - ref).abs().max()) / scale
        assert e_one <= 2e-6 and e_one <= 2.0 * e_two + 2e-7, (e_one, e_two)
    assert k.conv_error_word(d) == 0
    # the range word: one input element beyond 4094 (a legal fp32 activation) must be reported, for this launch's own slot
    xd2 = xd.clone()
    xd2[7, 0, 0, 3] = 5000.0
    k.conv1x1_chain(xd2, ch, residual=rd)
    with pytest.raises(k.Fp16RangeError):
        k.check_conv_error_word(d)
    assert ch.state["off"] and k.conv_error_word(d) == 0


@pytest.mark.parametrize("proj", [False, True])
@pytest.mark.parametrize("shape", [(1, 8, 32), (2, 13, 70), (1, 50, 84), (3, 9, 31), (1, 1, 1)])
def test_fused_bottleneck_matches_separate_launches_and_fp64(shape, proj):
    """csrc/conv_bneck.hip (kernels.BNECK / LVC_BNECK): conv1 -> conv2 -> conv3 + shortcut + ReLU of a res2 block (reference
    resnet.py:195-211) as one launch against the separate launches and an fp64 evaluation: whole tiles, ragged tile edges in both
    directions, more tiles than one workgroup's share, a single pixel; then the range word for an input beyond 4094."""
    import torch.nn.functional as F
    from lvc_amd import kernels as k

    N, H, W = shape
    cin = 64 if proj else 256
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W + cin)

    def bn(c):
        return (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5)

    w1 = torch.randn(64, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    w2 = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
    w3 = torch.randn(256, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    ws = torch.randn(256, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    b1, b2, b3, bs = bn(64), bn(64), bn(256), bn(256)
    x = torch.randn(N, H, W, cin, generator=g).relu_()
    d = _dev()
    dv = lambda ts: [t.to(d) for t in ts]
    p1, p2, p3 = k.pack_conv(w1.to(d), bn=dv(b1)), k.pack_conv(w2.to(d), bn=dv(b2), pad=1), k.pack_conv(w3.to(d), bn=dv(b3))
    a3, asc = k.conv_affine(None, b3), k.conv_affine(None, bs)
    if proj:
        wf = torch.cat([w3 * a3[0].view(-1, 1, 1, 1), ws * asc[0].view(-1, 1, 1, 1)], 1).contiguous()
        pf = k.pack_conv(wf.to(d), affine=(None, (a3[1] + asc[1]).to(d)))
        pss = k.pack_conv(ws.to(d), bn=dv(bs))
    bk = k.pack_bottleneck(p1, p2, pf if proj else p3, proj)
    xd = x.to(d)
    y = k.bottleneck_fused(xd, bk)
    t = k.conv2d_nhwc(k.conv2d_nhwc(xd, p1, relu=True), p2, relu=True)
    z = k.conv2d_nhwc(t, p3, relu=True, residual=k.conv2d_nhwc(xd, pss) if proj else xd, res_mode=1)

    def cb(v, w, b, pad=0):
        s_, t_ = k.conv_affine(None, b)
        return F.conv2d(v, w.double(), padding=pad) * s_.double().view(1, -1, 1, 1) + t_.double().view(1, -1, 1, 1)

    x0 = x.permute(0, 3, 1, 2).double()
    ref = (cb(cb(cb(x0, w1, b1).relu(), w2, b2, 1).relu(), w3, b3) + (cb(x0, ws, bs) if proj else x0)).relu().permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    e_one = float((y.cpu().double() - ref).abs().max()) / scale
    e_sep = float((z.cpu().double() - ref).abs().max()) / scale
    assert y.shape == (N, H, W, 256) and e_one <= 2e-6 and e_one <= 2.0 * e_sep + 3e-7, (e_one, e_sep)
    assert k.conv_error_word(d) == 0
    if shape == (2, 13, 70):
        x2 = xd.clone()
        x2[1, 12, 69, 5] = 5000.0      # the last pixel of the last (ragged) tile: a legal fp32 activation beyond the form's range
        k.bottleneck_fused(x2, bk)
        with pytest.raises(k.Fp16RangeError):
            k.check_conv_error_word(d)
        assert bk.state["off"] and k.conv_error_word(d) == 0


@pytest.mark.parametrize("shape", [(2, 50, 84, 256, 1024), (8, 25, 42, 512, 2048), (7, 37, 29, 256, 1024), (1, 46, 45, 64, 128)])
@pytest.mark.parametrize("with_res", [True, False])
def test_presplit_handover_bit_identical(shape, with_res, monkeypatch):
    """kernels.conv3x3_conv1x1_presplit (LVC_PRESPLIT; csrc `_presplit` entry points): a bottleneck's conv2 writes the two fp16 planes
    conv3 multiplies instead of the fp32 tensor conv3 would split (reference resnet.py:200-212).  Same products, same order: the
    result must equal the two ordinary launches BIT FOR BIT -- on res4's and res5's shapes, ragged tiles, and (pair forced) a narrow
    layer; `presplit_pair_ok` must agree with conv2d_nhwc's routing wherever it says yes.  Then the consumer's range word, raised by
    the producer."""
    from lvc_amd import kernels as k

    monkeypatch.setattr(k, "PRESPLIT", True)       # (off by default: kernels.PRESPLIT says why)
    N, H, W, cb, cout = shape
    g = torch.Generator().manual_seed(H * 100 + W + cb)
    d = _dev()

    def bn(c):
        return [t.to(d) for t in (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1,
                                  torch.rand(c, generator=g) + 0.5)]

    p2 = k.pack_conv((torch.randn(cb, cb, 3, 3, generator=g) * (2.0 / (9 * cb)) ** 0.5).to(d), bn=bn(cb), pad=1)
    p3 = k.pack_conv((torch.randn(cout, cb, 1, 1, generator=g) * (2.0 / cb) ** 0.5).to(d), bn=bn(cout))
    t = torch.randn(N, H, W, cb, generator=g).relu_().to(d)
    res = torch.randn(N, H, W, cout, generator=g).relu_().to(d) if with_res else None
    ok = k.presplit_pair_ok(t, p2, p3, res)
    assert ok == (cb >= 256), "presplit_pair_ok: res4 / res5 shapes qualify, a 64-channel pair does not"
    z = k.conv2d_nhwc(k.conv2d_nhwc(t, p2, relu=True), p3, relu=True, residual=res, res_mode=1 if with_res else 0)
    if not ok:
        return
    y = k.conv3x3_conv1x1_presplit(t, p2, p3, residual=res, relu=True)
    assert torch.equal(y, z)
    y2 = k.conv3x3_conv1x1_presplit(t, p2, p3, residual=res, relu=False)
    assert torch.equal(y2, k.conv2d_nhwc(k.conv2d_nhwc(t, p2, relu=True), p3, relu=False, residual=res, res_mode=1 if with_res else 0))
    assert k.conv_error_word(d) == 0
    if shape[1:3] == (37, 29):
        # conv2 output beyond conv3's single-accumulator range (|a| <= 4094): the PRODUCER raises conv3's word; conv3 moves a tier up and
        # the pair stops qualifying
        big = k.pack_conv((torch.randn(cb, cb, 3, 3, generator=g) * (2.0 / (9 * cb)) ** 0.5).to(d),
                          affine=(torch.full((cb,), 1.0, device=d), torch.full((cb,), 5000.0, device=d)), pad=1)
        k.conv3x3_conv1x1_presplit(t, big, p3, residual=res)
        with pytest.raises(k.Fp16RangeError):
            k.check_conv_error_word(d)
        assert p3.state["tier"] >= 1 and big.state["tier"] == 0 and not k.presplit_pair_ok(t, big, p3, res)
        assert k.conv_error_word(d) == 0


@pytest.mark.parametrize("jitter,nidx,thr,max_keep", [(4.0, 80, 0.5, 100), (0.02, 3, 0.5, 100), (4.0, 80, 0.5, 1000), (0.5, 1, 0.3, 64)])
def test_batched_nms_head_block_form_bit_exact(jitter, nidx, thr, max_keep):
    """lvc_batched_nms with max_keep << Nmax (the detection stage: 100 of ~10 000 candidates): the greedy pass runs on a head block of
    the sorted rows first and on the rest only for images that still lack boxes (csrc/nms.hip).  Keep lists must equal the oracle's
    first max_keep entries -- with counts below / inside / far beyond the head block, with so much overlap that the head block keeps
    fewer than max_keep boxes (the second block has to finish the list), and with score ties."""
    from lvc_amd import kernels as k
    from oracle import ops as oops

    g = torch.Generator().manual_seed(int(jitter * 100) + nidx)
    B, Nmax = 5, 12000
    counts = torch.tensor([12000, 900, 3000, 0, 7777], dtype=torch.int32)
    boxes = torch.zeros(B, Nmax, 4)
    scores = torch.zeros(B, Nmax)
    idxs = torch.zeros(B, Nmax, dtype=torch.int32)
    for b in range(B):
        bb, ss, ii = _nms_case(g, Nmax, nidx, jitter)
        ss[::11] = ss[3]
        if b == 2:      # 30 clusters of near-identical boxes: the head block keeps a few dozen, the rest of the list has to be walked
            few = bb[:30]
            bb = few[torch.randint(0, 30, (Nmax,), generator=g)] + torch.randn(Nmax, 4, generator=g) * 0.01
            ii = torch.zeros(Nmax, dtype=torch.int64)
        boxes[b], scores[b], idxs[b] = bb, ss, ii.int()
    d = _dev()
    keep, nk = k.batched_nms_batch(boxes.to(d), scores.to(d), idxs.to(d), counts.to(d), thr, max_keep=max_keep)
    keep, nk = keep.cpu(), nk.cpu()
    for b in range(B):
        n = int(counts[b])
        ref = oops.batched_nms(boxes[b, :n], scores[b, :n], idxs[b, :n].long(), thr)[:max_keep]
        assert int(nk[b]) == len(ref), (b, int(nk[b]), len(ref))
        assert keep[b, : len(ref)].tolist() == ref.tolist(), b


@pytest.mark.parametrize("hidden,pred_k,shapes,n", [(256, 16, ((37, 53), (19, 27), (10, 14)), 2), (128, 32, ((64, 96),), 1),
                                                   (256, 5, ((120, 168), (60, 84), (30, 42), (15, 21), (8, 11)), 3)])
def test_conv3x3_levels_with_a_pointwise_layer_on_top_vs_fp64(hidden, pred_k, shapes, n):
    """lvc_conv3x3_nhwc_f16_levels_pred at the edges of its contract: one or two 128-channel slices of the hidden layer, 5 .. 32
    pointwise outputs, one to five maps with ragged sizes (partial patches, maps smaller than a patch), batch 1 .. 3 -- against an fp64
    evaluation of relu(conv3x3) -> 1x1 (error of the order of the two launches': 2e-5 of the output scale), bit-identical run to run,
    and equal to the two separate launches within the pointwise layer's summation-order tolerance."""
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(hidden + pred_k)
    d = _dev()
    C = 64
    xs = [(torch.randn(n, h, w, C, generator=g) * 2).to(d) for h, w in shapes]
    wc = (torch.randn(hidden, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(d)
    bc = (torch.randn(hidden, generator=g) * 0.1).to(d)
    wp = (torch.randn(pred_k, hidden, 1, 1, generator=g) * (1.0 / hidden) ** 0.5).to(d)
    bp = torch.randn(pred_k, generator=g).to(d)
    pc = K.pack_conv(wc, bias=bc, stride=1, pad=1)
    pc.two_acc = True
    pred = K.pack_conv(wp, bias=bp)
    K.clear_conv_error_word(d)
    saved = K._HALO_H2_MIN_TILES
    K._HALO_H2_MIN_TILES = 1          # the small cases too on the pipelined kernel
    try:
        outs = K.conv3x3_levels_pred(xs, pc, pred, relu=True)
        again = K.conv3x3_levels_pred(xs, pc, pred, relu=True)
    finally:
        K._HALO_H2_MIN_TILES = saved
    assert outs is not None and len(outs) == len(xs)
    for x, o, o2 in zip(xs, outs, again):
        assert o.shape == (n, x.shape[1], x.shape[2], pred_k) and torch.equal(o, o2)
        ref = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wc.double(), bc.double(), padding=1)),
                                         wp.double(), bp.double()).permute(0, 2, 3, 1)
        scale = float(ref.abs().max())
        assert float((o.double() - ref).abs().max()) <= 2e-5 * scale
        two = K.conv2d_nhwc(K.conv2d_nhwc(x, pc, relu=True), pred)
        assert float((o - two).abs().max()) <= 2e-6 * scale
    assert K.conv_error_word(d) == 0


@pytest.mark.parametrize("h,w", [(25, 41), (50, 84), (1, 1), (2, 3)])
def test_subsample2_into_a_wider_row_is_the_strided_slice(h, w):
    """lvc_subsample2_nhwc (the input rows of a stride-2 projection shortcut next to conv2's output): out[n, i, j, :C] = x[n, 2i, 2j, :]
    for odd and even extents, written into the tail channels of a wider concat row without touching the rest."""
    from lvc_amd import kernels as K

    d = _dev()
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(2, h, w, 64, generator=g).to(d)
    hs, ws = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    buf = torch.full((2, hs, ws, 96), 7.0, device=d)
    K.subsample2_into(x, buf[..., 32:])
    assert torch.equal(buf[..., 32:], x[:, ::2, ::2, :]) and bool((buf[..., :32] == 7.0).all())
