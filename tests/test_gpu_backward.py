"""Backward into the trunk (SURVEY row 20 / BASELINE config 5 with `cascade_ubbr_R_50_FPN_base.yaml`, FREEZE_AT 2): the
weight-gradient kernel, the data gradient on the forward kernels, and the fused Conv2d / BottleneckBlock / FPN autograd
against torch's own conv2d backward in fp64 on the CPU (what the reference trains through).  Floating point: the
gradients must agree to 2e-5 of the tensor's norm (fp32 accumulation over up to 10^5 pixels; atomics reorder sums)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad,scaled", [
    (2, 25, 42, 64, 64, 3, 1, 1, True),
    (2, 51, 35, 256, 128, 1, 2, 0, True),
    (1, 50, 84, 256, 16, 1, 1, 0, False),     # the padded 15-channel RPN predictor
    (1, 13, 21, 256, 256, 3, 1, 1, False),
    (300, 1, 1, 1024, 1024, 1, 1, 0, False),  # Linear as the 1x1 case
    (2, 100, 168, 128, 512, 1, 1, 0, True),   # 4 tiles, 33 600 pixels: split over the pixel range
    (1, 37, 29, 96, 36, 3, 1, 1, False),      # channel tails inside the 128-wide tiles, rows narrower than a chunk
])
@pytest.mark.parametrize("engine", ["bf16x3", "f32"])
def test_conv_wgrad_matches_torch(N, H, W, C, K, R, stride, pad, scaled, engine, monkeypatch):
    """Both unscaled weight-gradient kernels (three-way bf16 split: the default; fp32 MFMA) on gradient-sized values
    against an fp64 torch reference."""
    from lvc_amd import kernels as Kn
    monkeypatch.setattr(Kn, "WGRAD_ENGINE", engine)
    g = torch.Generator().manual_seed(H * 131 + C)
    x = torch.randn(N, C, H, W, generator=g)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = torch.randn(N, K, Ho, Wo, generator=g) * 1e-4     # gradient-sized values
    scale = torch.rand(K, generator=g) + 0.5 if scaled else None
    w = torch.zeros(K, C, R, R, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), w, stride=stride, padding=pad)
    if scaled:
        y = y * scale.double().view(1, -1, 1, 1)
    (y * dy.double()).sum().backward()
    dev = torch.device("cuda:0")
    dw = Kn.conv_wgrad(x.permute(0, 2, 3, 1).contiguous().to(dev), dy.permute(0, 2, 3, 1).contiguous().to(dev),
                       scale.to(dev) if scaled else None, R, R, stride, pad)
    assert _rel(dw.permute(0, 3, 1, 2), w.grad) < TOL


def test_scatter_downsum_colsum():
    from lvc_amd import kernels as Kn
    dev = torch.device("cuda:0")
    x = torch.randn(2, 13, 21, 64, device=dev)
    y = Kn.scatter_stride2(x, 25, 42)
    ref = torch.zeros(2, 25, 42, 64, device=dev)
    ref[:, ::2, ::2] = x
    assert torch.equal(y, ref)
    y = Kn.scatter_stride2(x, 26, 41)
    ref = torch.zeros(2, 26, 41, 64, device=dev)
    ref[:, ::2, ::2] = x
    assert torch.equal(y, ref)
    z = torch.randn(2, 26, 42, 32, device=dev)
    d = Kn.downsum2x2(z)
    r = (z[:, ::2, ::2] + z[:, ::2, 1::2]) + (z[:, 1::2, ::2] + z[:, 1::2, 1::2])
    assert torch.equal(d, r)
    m = torch.randn(70001, 100, device=dev)                 # float4 rows, one partial column block, ragged last slab
    assert _rel(Kn.colsum_rows(m), m.double().sum(0)) < 1e-6
    for rows, cols in ((1, 256), (513, 260), (4099, 15), (777, 1024)):   # 15 columns: the 64-column scalar form
        m = torch.randn(rows, cols, device=dev)
        assert _rel(Kn.colsum_rows(m), m.double().sum(0)) < 1e-6


def _ref_conv(x, w, b, bn, stride, pad, relu, residual, res_mode):
    y = F.conv2d(x, w, b, stride=stride, padding=pad)
    if bn is not None:
        bw, bb, rm, rv = [t.double() for t in bn]
        sc = bw * (rv + 1e-5).rsqrt()
        y = y * sc.view(1, -1, 1, 1) + (bb - rm * sc).view(1, -1, 1, 1)
    if residual is not None:
        y = y + (residual if res_mode == 1 else F.interpolate(residual, scale_factor=2, mode="nearest"))
    return F.relu(y) if relu else y


@pytest.mark.parametrize("C,K,R,stride,pad,bn,bias,relu,res_mode,H,W", [
    (256, 64, 1, 1, 0, True, False, True, 0, 50, 84),     # conv1
    (64, 64, 3, 1, 1, True, False, True, 0, 50, 84),      # conv2
    (64, 256, 1, 1, 0, True, False, True, 1, 50, 84),     # conv3 + shortcut + ReLU
    (256, 128, 1, 2, 0, True, False, True, 0, 51, 85),    # strided conv1 (odd map)
    (512, 256, 1, 1, 0, False, True, False, 2, 50, 84),   # FPN lateral + up2(top)
    (256, 256, 3, 1, 1, False, True, False, 0, 26, 42),   # FPN output
    (256, 256, 3, 1, 1, False, True, True, 0, 13, 21),    # RPN conv
])
def test_conv2d_module_backward(C, K, R, stride, pad, bn, bias, relu, res_mode, H, W):
    from lvc_amd.layers import Conv2d, FrozenBatchNorm2d
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + K + R)
    norm = FrozenBatchNorm2d(K) if bn else None
    conv = Conv2d(C, K, kernel_size=R, stride=stride, padding=pad, bias=bias, norm=norm, activation=F.relu_ if relu else None)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(K, C, R, R, generator=g) * (2.0 / (C * R * R)) ** 0.5)
        if bias:
            conv.bias.copy_(torch.randn(K, generator=g) * 0.1)
        if bn:
            norm.weight.copy_(torch.rand(K, generator=g) + 0.5)
            norm.bias.copy_(torch.randn(K, generator=g) * 0.1)
            norm.running_mean.copy_(torch.randn(K, generator=g) * 0.1)
            norm.running_var.copy_(torch.rand(K, generator=g) + 0.5)
    conv = conv.to(dev)
    N = 2
    x = torch.randn(N, C, H, W, generator=g)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    res = None
    if res_mode == 1:
        res = torch.randn(N, K, Ho, Wo, generator=g)
    elif res_mode == 2:
        res = torch.randn(N, K, Ho // 2, Wo // 2, generator=g)
    dy = torch.randn(N, K, Ho, Wo, generator=g) * 1e-3
    # fp64 reference
    xr = x.double().requires_grad_(True)
    wr = conv.weight.detach().cpu().double().requires_grad_(True)
    br = conv.bias.detach().cpu().double().requires_grad_(True) if bias else None
    rr = res.double().requires_grad_(True) if res is not None else None
    bnr = [t.detach().cpu() for t in (norm.weight, norm.bias, norm.running_mean, norm.running_var)] if bn else None
    yr = _ref_conv(xr, wr, br, bnr, stride, pad, relu, rr, res_mode)
    (yr * dy.double()).sum().backward()
    # device
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True) if res is not None else None
    yd = conv.forward_nhwc(xd, residual=rd, res_mode=res_mode)
    assert _rel(yd.permute(0, 3, 1, 2), yr) < 1e-5
    yd.backward(dy.permute(0, 2, 3, 1).contiguous().to(dev))
    assert _rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < TOL
    assert _rel(conv.weight.grad, wr.grad) < TOL
    if bias:
        assert _rel(conv.bias.grad, br.grad) < TOL
    if res is not None:
        assert _rel(rd.grad.permute(0, 3, 1, 2), rr.grad) < TOL


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad,scaled", [
    (2, 25, 42, 64, 64, 3, 1, 1, True),
    (2, 51, 35, 256, 128, 1, 2, 0, True),
    (1, 50, 84, 256, 16, 1, 1, 0, False),
    (1, 13, 21, 256, 256, 3, 1, 1, False),
    (300, 1, 1, 1024, 1024, 1, 1, 0, False),
    (2, 100, 168, 128, 512, 1, 1, 0, True),
    (1, 37, 29, 96, 36, 3, 1, 1, False),       # channel tails inside the 128-wide tiles
])
def test_conv_wgrad_f16x2_matches_torch(N, H, W, C, K, R, stride, pad, scaled):
    """The fp16 MFMA form of the weight gradient (two-way operand split, LDS transpose reads) on gradients that a loss
    scale has brought into fp16's range: same fp64 reference and the same 2e-5 bar as the fp32 MFMA form."""
    from lvc_amd import kernels as Kn
    g = torch.Generator().manual_seed(H * 131 + C)
    x = torch.randn(N, C, H, W, generator=g)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = torch.randn(N, K, Ho, Wo, generator=g) * 0.05 * torch.exp(2.0 * torch.randn(N, K, Ho, Wo, generator=g))   # wide range
    scale = torch.rand(K, generator=g) + 0.5 if scaled else None
    w = torch.zeros(K, C, R, R, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), w, stride=stride, padding=pad)
    if scaled:
        y = y * scale.double().view(1, -1, 1, 1)
    (y * dy.double()).sum().backward()
    dev = torch.device("cuda:0")
    dw = Kn.conv_wgrad(x.permute(0, 2, 3, 1).contiguous().to(dev), dy.permute(0, 2, 3, 1).contiguous().to(dev),
                       scale.to(dev) if scaled else None, R, R, stride, pad, split="f16x2")
    Kn.check_conv_error_word(dev)
    assert _rel(dw.permute(0, 3, 1, 2), w.grad) < TOL


def test_conv_wgrad_f16x2_reports_out_of_range_gradients():
    from lvc_amd import kernels as Kn
    dev = torch.device("cuda:0")
    x = torch.randn(1, 8, 8, 64, device=dev)
    dy = torch.randn(1, 8, 8, 64, device=dev)
    dy[0, 3, 3, 5] = 1e6
    Kn.clear_conv_error_word(dev)
    Kn.conv_wgrad(x, dy, None, 1, 1, 1, 0, split="f16x2")
    assert Kn.conv_error_word(dev) & 2
    Kn.clear_conv_error_word(dev)


def test_linear_backward_keeps_tiny_gradients():
    """Unscaled gradients of a mean-reduced loss (1e-8 and below) through the FC backward at M >= 2048 rows and a
    contraction >= 128: the data-gradient GEMM must take the range-free split (DGRAD_SPLIT), not the two-way fp16 one,
    whose residual plane flushes below 2^-35 (its range word only sees overflow).  fp64 reference."""
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(11)
    M, Kin, Kout = 2560, 256, 192
    x = torch.randn(M, Kin, generator=g)
    w = torch.randn(Kout, Kin, generator=g) * 0.05
    for mag in (1e-6, 1e-8, 1e-10):
        dz = torch.randn(M, Kout, generator=g) * mag
        dx, dw = K.linear_backward(x.cuda(), w.cuda(), dz.cuda())
        rx = dz.double() @ w.double()
        rw = dz.double().t() @ x.double()
        ex = float((dx.cpu().double() - rx).norm() / rx.norm())
        ew = float((dw.cpu().double() - rw).norm() / rw.norm())
        print("gradient scale %.0e: dx rel %.2e, dw rel %.2e" % (mag, ex, ew))
        assert ex <= 5e-6 and ew <= 5e-6, mag


def test_grouped_wgrad_matches_single_launches():
    """lvc_conv_wgrad_group_bf16x3 + lvc_wgrad_finalize_group (round 5: the queued weight gradients of many layers in one launch,
    `kernels.flush_wgrad`) against one lvc_conv_wgrad_nhwc_bf16x3 launch per layer and against fp64: 30 jobs (more than one
    kernel-argument table), mixed 1x1 / 3x3 / strided / scaled shapes, ragged channel counts, a parameter used twice."""
    import ctypes

    from lvc_amd import _lib, kernels as K

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    shapes = [(2, 13, 17, 64, 128, 3, 1, 1, True), (2, 25, 42, 256, 256, 3, 1, 1, False), (1, 9, 11, 36, 20, 1, 1, 0, False),
              (2, 50, 84, 256, 64, 1, 1, 0, True), (2, 20, 30, 128, 256, 1, 2, 0, True), (3, 7, 9, 32, 160, 3, 1, 1, False)]
    jobs = []
    for i in range(30):
        N, H, W, C, Kc, R, stride, pad, scaled = shapes[i % len(shapes)]
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        x = torch.randn(N, H, W, C, generator=g).to(dev)
        dy = (torch.randn(N, Ho, Wo, Kc, generator=g) * 10.0 ** -(i % 7)).to(dev)       # gradients over many binades
        sc = (torch.rand(Kc, generator=g) + 0.5).to(dev) if scaled else None
        jobs.append((x, dy, sc, R, stride, pad))
    n = len(jobs)
    sizes = [j[1].shape[3] * j[3] * j[3] * j[0].shape[3] for j in jobs]
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + (s + 3) // 4 * 4)
    flat = torch.zeros(offs[-1], device=dev)
    PA = ctypes.c_void_p * n
    xs, dys, scs, dws, dsts = PA(), PA(), PA(), PA(), PA()
    sh = (ctypes.c_int * (10 * n))()
    fsh = (ctypes.c_int * (4 * n))()
    outs = []
    for j, (x, dy, sc, R, stride, pad) in enumerate(jobs):
        N, H, W, C = x.shape
        Kc = dy.shape[3]
        xs[j], dys[j], scs[j], dws[j] = x.data_ptr(), dy.data_ptr(), (sc.data_ptr() if sc is not None else None), flat.data_ptr() + 4 * offs[j]
        sh[10 * j: 10 * j + 10] = [N, H, W, C, Kc, R, R, stride, pad, Kc]
        beta = j % 2
        out = torch.full((Kc, C, R, R), 0.25 if beta else float("nan"), device=dev)
        outs.append(out)
        dsts[j] = out.data_ptr()
        fsh[4 * j: 4 * j + 4] = [Kc, C, R * R, beta]
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(_lib.lib().lvc_conv_wgrad_group_bf16x3(ctypes.c_int(n), xs, dys, scs, dws, sh, st), "group")
    _lib.check(_lib.lib().lvc_wgrad_finalize_group(ctypes.c_int(n), dws, dsts, fsh, st), "finalize")
    for j, (x, dy, sc, R, stride, pad) in enumerate(jobs):
        single = K._conv_wgrad(x, dy, sc, R, R, stride, pad).permute(0, 3, 1, 2)
        got = outs[j] - (0.25 if j % 2 else 0.0)
        xr = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(False)
        wr = torch.zeros(dy.shape[3], x.shape[3], R, R, dtype=torch.float64, requires_grad=True)
        yr = F.conv2d(xr, wr, stride=stride, padding=pad)
        gr = dy.double().cpu().permute(0, 3, 1, 2)
        if sc is not None:
            gr = gr * sc.double().cpu()[None, :, None, None]
        (yr * gr).sum().backward()
        scale = float(wr.grad.abs().max())
        tol = 2e-5 if j % 2 == 0 else 2e-5 + 1e-7 * 0.25 / max(scale, 1e-30)     # the accumulate form rounds (0.25 + dw) once
        assert float((got.cpu().double() - wr.grad).abs().max()) <= tol * scale, (j, jobs[j][0].shape)
        assert float((got - single).abs().max()) <= 1e-5 * scale + (1e-7 * 0.25 if j % 2 else 0.0), j


def test_deferred_wgrad_equals_the_immediate_path_shared_parameter_and_accumulation(monkeypatch):
    """`kernels.defer_wgrad`: a conv applied to two maps in one graph (the RPN head over pyramid levels: one accumulation buffer),
    `weight.grad` present when backward() returns, a second backward() accumulates into it -- against LVC_DEFER_WGRAD=0 (autograd's
    own AccumulateGrad)."""
    from lvc_amd import kernels as K
    from lvc_amd.layers import Conv2d

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(9)
    conv = Conv2d(64, 128, 3, padding=1, bias=True).to(dev)
    xa = torch.randn(2, 24, 40, 64, generator=g).to(dev)
    xb = torch.randn(2, 12, 20, 64, generator=g).to(dev)

    def run():
        conv.weight.grad = conv.bias.grad = None
        for _ in range(2):
            (conv.forward_nhwc(xa).square().sum() * 1e-3 + conv.forward_nhwc(xb).sum() * 1e-2).backward()
            assert conv.weight.grad is not None and not K._WGRAD_Q
        return conv.weight.grad.clone(), conv.bias.grad.clone()

    monkeypatch.setattr(K, "DEFER_WGRAD", False)
    w0, b0 = run()
    monkeypatch.setattr(K, "DEFER_WGRAD", True)
    for side in (False, True):       # the grouped launches on the pass's own stream / on a second stream next to the data-gradient chain
        monkeypatch.setattr(K, "WGRAD_SIDE_STREAM", side)
        w1, b1 = run()
        torch.cuda.synchronize()
        assert not K._WGRAD_PENDING
        assert _rel(w1, w0) < 1e-5 and _rel(b1, b0) < 1e-6
        assert w1.is_contiguous() and w1.shape == conv.weight.shape


def test_prepack_group_equals_the_per_layer_packing():
    """`Conv2d.prepack` (one grouped launch: csrc/weights.hip pack_group_kernel) fills the same operands, bit for bit, as `packed()` /
    `packed_dgrad()` + the lazy splits do one layer at a time: fp32 operand, plain fp16 planes (two-accumulator layers), row-scaled
    fp16 planes + row factors (one-accumulator layers), bf16 planes (short contractions, data gradients)."""
    from lvc_amd import kernels as K
    from lvc_amd.layers import Conv2d, FrozenBatchNorm2d

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    specs = [(256, 256, 3, 1, True, False), (256, 128, 3, 1, False, True), (512, 128, 1, 0, True, False), (64, 256, 1, 0, True, False),
             (32, 40, 1, 0, False, False), (128, 512, 1, 0, True, True), (1024, 2048, 1, 0, True, False)] * 5     # 35 layers: two tables
    convs = []
    for C, Kc, R, pad, bn, two in specs:
        norm = FrozenBatchNorm2d(Kc) if bn else None
        conv = Conv2d(C, Kc, R, padding=pad, bias=not bn, norm=norm)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(Kc, C, R, R, generator=g) * 10.0 ** float(torch.randint(-3, 2, (1,), generator=g)))
            if bn:
                norm.weight.copy_(torch.rand(Kc, generator=g) + 0.5)
                norm.running_var.copy_(torch.rand(Kc, generator=g) + 0.5)
        conv.two_acc = two
        convs.append(conv.to(dev))
    Conv2d.prepack(convs)
    got = [(c._cache.value, c._cache_dgrad.value) for c in convs]
    for c in convs:
        c._cache.key = c._cache_dgrad.key = None
    for c, (pf, pd) in zip(convs, got):
        rf, rd = c.packed(), c.packed_dgrad()
        assert rf is not pf and rd is not pd
        assert torch.equal(pf.w, rf.w) and torch.equal(pd.w, rd.w) and (pf.K, pf.C, pf.R, pf.stride, pf.pad, pf.Kg) == (rf.K, rf.C, rf.R, rf.stride, rf.pad, rf.Kg)
        assert (pd.K, pd.C, pd.R, pd.stride, pd.pad, pd.Kg) == (rd.K, rd.C, rd.R, rd.stride, rd.pad, rd.Kg)
        if pf._w2s is not None:
            planes, fac = rf.split2s()
            assert torch.equal(pf._w2s[0].view(torch.int16), planes.view(torch.int16)) and torch.equal(pf._w2s[1], fac)
        if pf._w2h is not None:
            assert torch.equal(pf._w2h.view(torch.int16), rf.split2h().view(torch.int16))
        if pf._w3 is not None:
            assert torch.equal(pf._w3.view(torch.int16), rf.split3().view(torch.int16))
        assert pd._w3 is not None and torch.equal(pd._w3.view(torch.int16), rd.split3().view(torch.int16))
    kinds = {("s" if p._w2s is not None else "h" if p._w2h is not None else "3") for p, _ in got}
    assert kinds == {"s", "h", "3"}
    # and the forward through a pre-packed layer equals the forward through a per-layer-packed one
    x = torch.randn(2, 20, 28, 256, generator=g).to(dev)
    y_ref = convs[0].forward_nhwc(x)
    convs[0]._cache.key = None
    Conv2d.prepack(convs[:1])
    assert torch.equal(convs[0].forward_nhwc(x), y_ref)


def test_prepack_plan_is_reused_after_an_optimizer_step(monkeypatch):
    """The plan kept between steps (`Conv2d.prepack(..., holder=)`): after an in-place parameter update the SAME buffers are filled
    again (no new PackedConv objects), lazily made planes of the old parameters are dropped, the Winograd operand is rebuilt, and
    the operands equal a fresh per-layer packing of the new parameters."""
    from lvc_amd import kernels as K
    from lvc_amd.layers import Conv2d, FrozenBatchNorm2d

    dev = torch.device("cuda", 0)
    monkeypatch.setattr(K, "_WINO_MIN_TILES", 1)
    g = torch.Generator().manual_seed(12)
    convs = [Conv2d(256, 256, 3, padding=1, bias=False, norm=FrozenBatchNorm2d(256)).to(dev), Conv2d(256, 64, 1, bias=True).to(dev)]
    holder = {}
    Conv2d.prepack(convs, holder=holder)
    plan = holder["plan"]
    first = [c.packed() for c in convs]
    first[0].split2h()                                    # a lazily made plane set of the OLD parameters
    x = torch.randn(1, 128, 160, 256, generator=g).to(dev)   # enough tiles for the Winograd route
    y_old = convs[0].forward_nhwc(x)
    assert "_wino" in first[0].state
    with torch.no_grad():
        for c in convs:
            c.weight.add_(torch.randn(c.weight.shape, generator=g).to(dev) * 0.05)
    Conv2d.prepack(convs, holder=holder)
    assert holder["plan"] is plan and [c.packed() for c in convs] == first and first[0]._w2h is None and "_wino" not in first[0].state
    y_new = convs[0].forward_nhwc(x)
    assert not torch.equal(y_new, y_old)
    got = [(c._cache.value.w.clone(), c._cache.value._w2s[0].clone(), c._cache_dgrad.value._w3.clone()) for c in convs]
    for c in convs:
        c._cache.key = c._cache_dgrad.key = None
    for c, (w, pl, d3) in zip(convs, got):
        rf, rd = c.packed(), c.packed_dgrad()
        assert torch.equal(w, rf.w) and torch.equal(pl.view(torch.int16), rf.split2s()[0].view(torch.int16)) and torch.equal(d3.view(torch.int16), rd.split3().view(torch.int16))
    assert torch.equal(convs[0].forward_nhwc(x), y_new)


def test_weight_gradient_is_not_deferred_under_autograd_grad():
    """`torch.autograd.grad` must get the weight gradient back from the node (nothing may be written to `.grad`), and `backward(inputs=[x])`
    must leave `weight.grad` alone: the deferred path (`kernels.defer_wgrad`) is taken only when the engine will run the weight's
    AccumulateGrad node."""
    from lvc_amd import kernels as K
    from lvc_amd.layers import Conv2d

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(21)
    conv = Conv2d(64, 128, 3, padding=1, bias=False).to(dev)
    x = torch.randn(2, 24, 40, 64, generator=g).to(dev).requires_grad_(True)
    conv.forward_nhwc(x).square().sum().backward()
    ref_w, ref_x = conv.weight.grad.clone(), x.grad.clone()
    conv.weight.grad = None
    x.grad = None
    gx, gw = torch.autograd.grad(conv.forward_nhwc(x).square().sum(), [x, conv.weight])
    assert conv.weight.grad is None and not K._WGRAD_Q and not K._WGRAD_ARMED[0]
    assert _rel(gw, ref_w) < 1e-5 and _rel(gx, ref_x) < 1e-6
    (gx2,) = torch.autograd.grad(conv.forward_nhwc(x).square().sum(), [x])
    assert conv.weight.grad is None and not K._WGRAD_Q and _rel(gx2, ref_x) < 1e-6
    conv.forward_nhwc(x).square().sum().backward(inputs=[x])
    assert conv.weight.grad is None and not K._WGRAD_Q and _rel(x.grad, ref_x) < 1e-6


def test_deferred_wgrad_queue_survives_a_failed_backward_and_buckets_take_the_gradients_in_place():
    """(1) A backward() that raises after weight gradients were queued leaves them behind; the next forward drops the stale queue
    (`kernels.reset_wgrad_queue`) and the next backward() delivers exactly its own gradients.  (2) With `GradientBuckets` and
    `zero_grad(set_to_none=True)` the deferred gradients are written straight into the bucket slices, step after step."""
    from lvc_amd import distributed as D, kernels as K
    from lvc_amd.layers import Conv2d

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(31)
    conv = Conv2d(64, 64, 3, padding=1, bias=True).to(dev)
    x = torch.randn(1, 16, 24, 64, generator=g).to(dev)

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, gr):
            raise RuntimeError("boom")

    xin = x.clone().requires_grad_(True)
    y = conv.forward_nhwc(Boom.apply(conv.forward_nhwc(xin)))       # the outer conv's gradient is queued, then the inner node raises
    with pytest.raises(RuntimeError, match="boom"):
        y.sum().backward()
    assert K._WGRAD_Q and K._WGRAD_ARMED[0] and conv.weight.grad is None
    conv.forward_nhwc(x).sum().backward()
    assert not K._WGRAD_Q and not K._WGRAD_ARMED[0]
    got = conv.weight.grad.clone()
    conv.weight.grad = None
    K.DEFER_WGRAD, prev = False, K.DEFER_WGRAD
    try:
        conv.forward_nhwc(x).sum().backward()
    finally:
        K.DEFER_WGRAD = prev
    assert _rel(got, conv.weight.grad) < 1e-5
    # (2)
    params = list(conv.parameters())
    for p in params:
        p.grad = None
    buckets = D.GradientBuckets(params, bucket_bytes=1 << 20)
    opt = torch.optim.SGD(params, lr=0.1)
    try:
        for step in range(2):
            opt.zero_grad(set_to_none=True)
            conv.forward_nhwc(x).square().sum().backward()
            buckets.finish()
            flat = buckets.buckets[0]["flat"]
            for p in params:
                assert p.grad is not None and flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4
            opt.step()
    finally:
        buckets.remove()


def test_shared_deferred_weight_is_delivered_once_after_its_last_use(monkeypatch):
    """a flush of the deferred weight gradients (every 24 jobs) DELIVERS its parameters -- `p.grad` is set and a bucketed
    exchange may start reducing the slice.  A weight used several times in one pass (the RPN head over the pyramid levels) whose uses
    straddle two flushes must wait for its last use: one delivery, carrying the whole gradient.  Also a forward inside a live backward
    (checkpoint recomputation) must not drop the queue, and a dead exchange's sinks must not receive gradients."""
    import gc
    from lvc_amd import kernels as K
    from lvc_amd.layers import Conv2d

    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    shared = Conv2d(32, 32, 3, padding=1, bias=False).to(dev)
    others = [Conv2d(32, 32, 1, bias=False).to(dev) for _ in range(30)]
    params = [shared.weight] + [c.weight for c in others]
    x = torch.randn(1, 12, 20, 32, device=dev)

    def loss():
        h = x
        for k in range(5):
            h = shared.forward_nhwc(h)
            for c in others[6 * k: 6 * k + 6]:
                h = c.forward_nhwc(h) * 0.5
        return h.square().sum()

    monkeypatch.setattr(K, "DEFER_WGRAD", False)
    loss().backward()
    ref = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    monkeypatch.setattr(K, "DEFER_WGRAD", True)

    class Exchange:      # what GradientBuckets does with its slices: hand them out, act when told a parameter is there
        def __init__(self):
            self.slots = {id(p): torch.zeros_like(p) for p in params}
            self.seen = {}
            for p in params:
                K.register_wgrad_sink(p, self.slot, self.ready, owner=self)

        def slot(self, q):
            return self.slots[id(q)]

        def ready(self, q):
            self.seen.setdefault(id(q), []).append(self.slots[id(q)].clone())

    ex = Exchange()
    with pytest.raises(RuntimeError, match="already has a deferred-gradient sink"):
        K.register_wgrad_sink(params[0], lambda q: None, lambda q: None, owner=ex)
    flushes = []
    real_flush = K.flush_wgrad
    monkeypatch.setattr(K, "flush_wgrad", lambda final=False: (flushes.append((final, len(K._WGRAD_Q))), real_flush(final))[1])
    loss().backward()
    torch.cuda.synchronize()
    assert len(flushes) >= 2 and not flushes[0][0] and flushes[-1][0], flushes      # at least one flush before the final one
    for p, r in zip(params, ref):
        assert len(ex.seen[id(p)]) == 1, "delivered %d times" % len(ex.seen[id(p)])
        assert _rel(ex.seen[id(p)][0], r) < 1e-5 and _rel(p.grad, r) < 1e-5       # complete when reported, and final
    assert not K._WGRAD_Q and not K._WGRAD_USES
    # a forward inside a live backward keeps the queue
    for p in params:
        p.grad = None
    ex.seen.clear()

    class Recompute(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, gr):
            assert K.in_backward() and K._WGRAD_Q          # the outer layers' gradients are queued
            n = len(K._WGRAD_Q)
            with torch.enable_grad():
                others[0].forward_nhwc(x.detach())         # a differentiable forward (recomputation) while the pass is running
            assert len(K._WGRAD_Q) == n
            return gr

    (others[2].forward_nhwc(Recompute.apply(others[1].forward_nhwc(x)))).sum().backward()
    assert others[2].weight.grad is not None and others[1].weight.grad is not None and not K._WGRAD_Q
    # a dropped exchange: its sinks die with it
    for p in params:
        p.grad = None
    del ex
    gc.collect()
    loss().backward()
    assert all(p.grad is not None for p in params) and K._wgrad_sink(params[0]) is None
    for p, r in zip(params, ref):
        assert _rel(p.grad, r) < 1e-5


def test_weight_gradient_takes_the_autograd_path_when_a_hook_listens():
    """A tensor hook or a post-accumulate hook on the weight (what a data-parallel wrapper other than GradientBuckets relies on) must see the
    gradient: such a parameter is not deferred."""
    from lvc_amd import kernels as K
    from lvc_amd.layers import Conv2d

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(41)
    conv = Conv2d(64, 64, 3, padding=1, bias=False).to(dev)
    x = torch.randn(1, 16, 24, 64, generator=g).to(dev)
    seen = []
    h = conv.weight.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.detach().clone()))
    conv.forward_nhwc(x).sum().backward()
    assert len(seen) == 1 and not K._WGRAD_Q and torch.equal(seen[0], conv.weight.grad)
    h.remove()
    conv.weight.grad = None
    h2 = conv.weight.register_hook(lambda gr: seen.append(gr.detach().clone()))
    conv.forward_nhwc(x).sum().backward()
    assert len(seen) == 2 and _rel(seen[1], seen[0]) < 1e-5
    h2.remove()
