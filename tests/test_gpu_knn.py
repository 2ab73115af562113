"""Label-verification kNN sweep vs golden vectors from the reference (tools/run_nearest_neighbours.py) and the
oracle.  `keep` and the vote are integer outputs: exact.  top10 class ids: exact on the fixture (no near ties)."""
import pytest
import torch

from helpers import gold

pytestmark = pytest.mark.gpu
D = "cuda:0"


@pytest.mark.parametrize("tag,cosine", [("cos", True), ("l2", False)])
def test_knn_golden(tag, cosine):
    from lvc_amd.label_verification import knn_sweep

    g = gold("knn")
    top, keep = knn_sweep(g["shot_classes"].to(D), g["shots"].to(D), g["q_desc"].to(D), g["q_classes"].to(D), 10, cosine)
    assert torch.equal(keep.cpu(), g["keep_" + tag])
    assert torch.equal(top.cpu(), g["top10_" + tag])


def test_knn_reference_signature_roundtrip():
    from lvc_amd.label_verification import get_nn_class_confirmatory, run_nearest_neighbours
    from lvc_amd.structures import Instances

    g = gold("knn")
    counts = g["counts"].tolist()
    qs, o = [], 0
    for n in counts:
        inst = Instances((10, 10))
        inst.crop_feats = g["q_desc"][o: o + n]
        inst.gt_classes = g["q_classes"][o: o + n]
        qs.append({"instances": inst})
        o += n
    run_nearest_neighbours(g["shot_classes"], g["shots"], qs, True)
    get_nn_class_confirmatory(qs, 10)
    assert torch.equal(torch.cat([q["instances"].keep for q in qs]), g["keep_cos"])
    assert torch.equal(torch.cat([q["instances"].top10_shots.reshape(-1, 10) for q in qs]), g["top10_cos"])


def test_knn_large_vs_oracle_dense():
    """BASELINE config 4 shape at 1/8 scale: 15k x 2400 x 1024, 80 classes x 30 shots."""
    from lvc_amd.label_verification import knn_sweep
    from oracle import knn as oknn

    g = torch.Generator().manual_seed(0)
    S, Dm, Q = 2400, 1024, 15000
    classes = torch.arange(80).repeat_interleave(30)
    centers = torch.randn(80, Dm, generator=g)
    shots = centers[classes] + 2.0 * torch.randn(S, Dm, generator=g) + 0.3
    qcls = torch.randint(0, 80, (Q,), generator=g)
    q = centers[qcls] + 2.5 * torch.randn(Q, Dm, generator=g) + 0.3
    det = torch.where(torch.rand(Q, generator=g) < 0.7, qcls, torch.randint(0, 80, (Q,), generator=g))
    top, keep = knn_sweep(classes.to(D), shots.to(D), q.to(D), det.to(D), 10, True)
    ref_top = oknn.dense(classes, shots, q, True)
    ref_keep = oknn.get_nn_class_confirmatory(ref_top, det, 10)
    # fp32 similarity ties between two shots of different classes can swap neighbours: allow 0.1 % of rows
    assert (top.cpu() != ref_top).any(dim=1).float().mean() <= 1e-3
    assert (keep.cpu() != ref_keep).float().mean() <= 1e-3


@pytest.mark.parametrize("S", [10, 63, 64, 700, 2400, 4096])
def test_topk_vote_ties_and_sizes(S):
    """knn_topk_vote_kernel against torch.topk semantics with the reference's tie rule made explicit (ties -> lower
    shot index, i.e. a stable descending sort): rows with few distinct values (every candidate filter overflows into
    the full-row path), ties exactly at the 10th place, constant rows, and every per-lane register bucket."""
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(S)
    Q = 300
    sims = torch.randn(Q, S, generator=g)
    sims[0] = 0.25                                               # constant row
    sims[1] = torch.randint(0, 3, (S,), generator=g).float()     # 3 distinct values
    sims[2] = torch.randint(0, 12, (S,), generator=g).float()    # ties around the 10th place
    sims[3, : min(S, 9)] = 5.0                                   # 9 clear winners, then a tie block
    sims[4] = torch.arange(S).float()                            # winners at the END of the row
    sims[5] = -torch.arange(S).float()
    if S > 300:
        sims[6, 100:400] = 7.0                                   # 300 tied candidates: beyond the LDS candidate list
    shot_classes = torch.randint(0, 80, (S,), generator=g)
    det = torch.randint(0, 80, (Q,), generator=g)
    order = torch.sort(sims, dim=1, descending=True, stable=True)[1][:, :10]
    ref_top = shot_classes[order]
    ref_keep = (torch.mode(ref_top, dim=1)[0] == det).long()
    top, keep = K.knn_topk_vote(sims.to(D), S, shot_classes.to(D), det.to(D), 10)
    assert torch.equal(top.cpu(), ref_top)
    assert torch.equal(keep.cpu(), ref_keep)


@pytest.mark.parametrize("Q,S,Dm,cosine", [(37, 300, 384, True), (2500, 600, 384, True), (5000, 1801, 96, True),
                                            (1000, 330, 384, False), (40000, 2400, 384, True),
                                            # every row length of the 16-bit verification kernel (ceil(S / 512) = 3, 6, 7, 8; the others above)
                                            (1500, 1300, 96, True), (1500, 2900, 96, True), (1500, 3500, 64, True), (1500, 4096, 64, True)])
def test_knn_sweep_shapes(Q, S, Dm, cosine):
    """Descriptor widths and set sizes other than the benchmark's -- 384 is the ViT-S/8 width the reference's DINO
    descriptors have (tools/run_nearest_neighbours.py:292-293) -- including query counts below the 256-row GEMM tile, shot
    counts that are not multiples of 64, more than one 32 768-query chunk, and the L2 branch (:155-159)."""
    from lvc_amd.label_verification import knn_sweep
    from oracle import knn as oknn

    g = torch.Generator().manual_seed(Q + S)
    ncls = max(2, S // 30)
    classes = torch.sort(torch.randint(0, ncls, (S,), generator=g))[0]
    centers = torch.randn(ncls, Dm, generator=g)
    shots = centers[classes] + 1.5 * torch.randn(S, Dm, generator=g) + 0.3
    qcls = torch.randint(0, ncls, (Q,), generator=g)
    q = centers[qcls] + 2.0 * torch.randn(Q, Dm, generator=g) + 0.3
    top, keep = knn_sweep(classes.to(D), shots.to(D), q.to(D), qcls.to(D), 10, cosine)
    ref_top = oknn.dense(classes, shots, q, cosine)
    ref_keep = oknn.get_nn_class_confirmatory(ref_top, qcls, 10)
    assert (top.cpu() != ref_top).any(dim=1).float().mean() <= 2e-3
    assert (keep.cpu() != ref_keep).float().mean() <= 2e-3


def _unit_rows(n, d, g):
    x = torch.randn(n, d, generator=g, dtype=torch.float64)
    return x / x.norm(dim=1, keepdim=True)


@pytest.mark.parametrize("q15", [False, True])
@pytest.mark.parametrize("Dm,clustered", [(64, False), (384, False), (1024, False), (2048, False), (384, True), (1024, True)])
def test_verify_kernel_recovers_exact_top10_from_noisy_prefilter(Dm, clustered, q15):
    """knn_verify_topk_vote_kernel with a pre-filter matrix perturbed by the WORST noise its contract allows (+-2^-10,
    uniformly random per entry -- far rougher than the fp16 GEMM): the class sequence of the exact ten best (fp64 ranking of
    the same rows) must come back wherever fp32 can tell the shots apart.  Random classes make nearly every candidate
    "flagged" (a shot of another class within margin: exact re-evaluation); `clustered` gives the shots of a class a common
    direction, so that most neighbours share the class and are ranked by their approximate values alone.  A block of 300
    identical shots drives rows through the all-shots path and the tie rule (lower shot index)."""
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(Dm + clustered)
    Q, S = 500, 1500
    if clustered:
        shot_classes = torch.arange(S) // 30
        centers = _unit_rows(S // 30, Dm, g)
        sn = centers[shot_classes] * 0.5 + _unit_rows(S, Dm, g)
        sn = sn / sn.norm(dim=1, keepdim=True)
        qn = centers[torch.randint(0, S // 30, (Q,), generator=g)] * 0.5 + _unit_rows(Q, Dm, g)
    else:
        shot_classes = torch.randint(0, 80, (S,), generator=g)
        sn = _unit_rows(S, Dm, g)
        qn = _unit_rows(Q, Dm, g)
    sn[200:500] = sn[200]                       # 300 identical shots
    qn[:20] = sn[200] * 0.8 + 0.6 * qn[:20]      # queries next to the block: > KV_MAX_CAND candidates
    qn = qn / qn.norm(dim=1, keepdim=True)
    if clustered:
        shot_classes = shot_classes.clone()
        shot_classes[200:500] = torch.randint(0, 50, (300,), generator=g)
    sims = qn @ sn.t()
    approx = (sims + (torch.rand(Q, S, generator=g, dtype=torch.float64) * 2 - 1) * 2.0 ** -10).float()
    det = torch.randint(0, 50, (Q,), generator=g)
    margin = 2.0 ** -9 + 2.0 ** -16
    if q15:
        # the 16-bit fixed-point form of the matrix (lvc_gemm_f16_q15's encoding) in rows padded to a multiple of 8 columns, the pad
        # columns holding values that would win if they were read; knn_verify_q15_kernel behind the same entry point
        from lvc_amd.label_verification import Q15_MARGIN
        pad = torch.full((Q, (S + 7) // 8 * 8), 32000, dtype=torch.int16)
        pad[:, :S] = (approx * 32766.0).round().clamp(-32766, 32766).to(torch.int16)
        approx_dev = pad.to(D)[:, :S]
        margin += Q15_MARGIN
    else:
        approx_dev = approx.to(D)
    top, keep = K.knn_verify_topk_vote(approx_dev, qn.float().contiguous().to(D), sn.float().contiguous().to(D),
                                       margin, shot_classes.to(D), det.to(D), 10)
    qs, ss = qn.float().double(), sn.float().double()      # the fp32 rows the kernel reads, evaluated in fp64
    s64 = qs @ ss.t()
    val, order = torch.sort(s64, dim=1, descending=True, stable=True)
    ref_top = shot_classes[order[:, :10]]
    clear = ((val[:, :10] - val[:, 1:11]) > 2e-6).all(dim=1)     # rows whose top 11 are separated beyond fp32 summation noise
    clear[:20] = True                                            # block rows: identical shots tie exactly in any precision ...
    assert clear.float().mean() > 0.9
    got = top.cpu()
    assert torch.equal(got[20:][clear[20:]], ref_top[20:][clear[20:]])
    # ... so their winners are the block's lowest indices unless a non-block shot beats them
    blk = (order[:20, :10] >= 200) & (order[:20, :10] < 500)
    assert blk.all(), "test construction: the block should own the top ten of the first 20 rows"
    assert torch.equal(got[:20], shot_classes[200:210].expand(20, 10))
    ref_keep = (torch.mode(got, dim=1)[0] == det).long()
    assert torch.equal(keep.cpu(), ref_keep)


def test_verify_kernel_normalises_raw_queries_like_rownorm():
    """Raw descriptors + (mu, den) from rownorm_h give the same answer as the pre-normalised rows: the float kernel redoes
    (q - mu) / den bit for bit; the q15 kernel divides the finished dot product of q - mu by den instead (one more rounding
    than on pre-normalised rows: rows whose neighbours tie to fp32 rounding may swap)."""
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(3)
    Q, S, Dm = 3000, 900, 384
    shots = (torch.randn(S, Dm, generator=g) + 0.4).to(D)
    q = (torch.randn(Q, Dm, generator=g) + 0.4).to(D)
    classes = torch.randint(0, 20, (S,), generator=g).to(D)
    det = torch.randint(0, 20, (Q,), generator=g).to(D)
    mu = K.colmean(shots)
    sn, sh, _ = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1)
    qn, qh, den = K.rownorm_h(q, mu=mu, eps=1e-8, mode=1)
    assert torch.equal(qn, K.rownorm(q, mu=mu, eps=1e-8, mode=1)) and torch.equal(qn, (q - mu) / den[:, None])
    _, qh2, den2 = K.rownorm_h(q, mu=mu, eps=1e-8, mode=1, want_rows=False)
    assert torch.equal(qh, qh2) and torch.equal(den, den2) and torch.equal(qh, qn.half())
    ap = K.gemm_f16(qh, sh)
    m = 2.0 ** -9 + 2.0 ** -16
    a = K.knn_verify_topk_vote(ap, qn, sn, m, classes, det, 10)
    b = K.knn_verify_topk_vote(ap, q, sn, m, classes, det, 10, mu=mu, den=den)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    from lvc_amd.label_verification import Q15_MARGIN
    ap16 = K.gemm_f16(qh, sh, q15=True)
    assert ap16.dtype == torch.int16 and ap16.stride(0) % 8 == 0
    assert (ap16.float() / 32766.0 - ap).abs().max().item() <= 0.5 / 32766.0 + 1e-7
    c = K.knn_verify_topk_vote(ap16, qn, sn, m + Q15_MARGIN, classes, det, 10)
    d = K.knn_verify_topk_vote(ap16, q, sn, m + Q15_MARGIN, classes, det, 10, mu=mu, den=den)
    for other in (c, d):     # fp32 evaluations of the same dot products in another sum order: only fp32-level ties may swap
        assert (a[0] != other[0]).any(dim=1).float().mean() <= 1e-3 and (a[1] != other[1]).float().mean() <= 1e-3


def test_two_stage_equals_single_stage(monkeypatch):
    """The fp16 pre-filter + exact verification and the full-precision similarity matrix pick the same neighbours (both rank
    fp32 evaluations of the same dot products; rows where two shots tie to fp32 rounding may swap)."""
    from lvc_amd import label_verification as LV

    g = torch.Generator().manual_seed(5)
    S, Dm, Q = 2400, 1024, 20000
    classes = torch.arange(80).repeat_interleave(30)
    centers = torch.randn(80, Dm, generator=g)
    shots = (centers[classes] + 2.0 * torch.randn(S, Dm, generator=g) + 0.3).to(D)
    qcls = torch.randint(0, 80, (Q,), generator=g)
    q = (centers[qcls] + 2.5 * torch.randn(Q, Dm, generator=g) + 0.3).to(D)
    out = {}
    for two_stage, emit in ((True, False), (False, False)):
        monkeypatch.setattr(LV, "KNN_TWO_STAGE", two_stage)
        out[two_stage, emit] = LV.knn_sweep(classes.to(D), shots, q, qcls.to(D), 10, True)
    assert (out[True, False][0] != out[False, False][0]).any(dim=1).float().mean() <= 5e-4
    assert (out[True, False][1] != out[False, False][1]).float().mean() <= 5e-4


def test_fp16_gemm_error_bound_and_exactness():
    """lvc_gemm_f16: (a) on operands that are exactly representable products (small integers) the result is exact for every
    tile-edge shape; (b) on unit-norm rows rounded to fp16 by lvc_rownorm_h the distance to the fp64 product of the fp32 rows
    stays inside the bound the verification margin is derived from (2^-10)."""
    from lvc_amd import kernels as K

    g = torch.Generator().manual_seed(11)
    for M, S, Dm in [(700, 333, 96), (257, 64, 64), (1, 10, 32), (5000, 2400, 384), (513, 257, 1024)]:
        a = torch.randint(-4, 5, (M, Dm), generator=g).half()
        b = torch.randint(-4, 5, (S, Dm), generator=g).half()
        y = K.gemm_f16(a.to(D), b.to(D))
        assert torch.equal(y.cpu(), a.float() @ b.float().t())
    for M, S, Dm in [(3000, 2400, 1024), (700, 333, 384)]:
        a, b = torch.randn(M, Dm, generator=g) + 0.2, torch.randn(S, Dm, generator=g) + 0.2
        a[0] = 0
        a[0, 3] = 5.0                                       # one-hot rows: the largest single-term products
        b[1] = 0
        b[1, 3] = -7.0
        an, ah, _ = K.rownorm_h(a.to(D), eps=1e-8, mode=1)
        bn, bh, _ = K.rownorm_h(b.to(D), eps=1e-8, mode=1)
        assert torch.equal(an, K.rownorm(a.to(D), eps=1e-8, mode=1)) and torch.equal(ah, an.half())
        y = K.gemm_f16(ah, bh)
        exact = an.double() @ bn.double().t()
        assert (y.double() - exact).abs().max().item() <= 2.0 ** -10
        # the 16-bit fixed-point form of the same products: rint(32766 x) of the value clamped to [-1, 1], rows padded to 8 columns
        y16 = K.gemm_f16(ah, bh, q15=True)
        assert y16.dtype == torch.int16 and y16.shape == y.shape and y16.stride(0) % 8 == 0
        want = y.double().clamp(-1.0, 1.0) * 32766.0
        assert (y16.double() - want).abs().max().item() <= 0.5 + 1e-6 and int(y16.min()) >= -32766 and int(y16.max()) <= 32766
    # NaN operands take the code that sorts first; the rest of the matrix is untouched
    ah2 = ah.clone()
    ah2[5, 7] = float("nan")
    y16n = K.gemm_f16(ah2, bh, q15=True)
    assert (y16n[5] == 32767).all() and torch.equal(y16n[:5], y16[:5]) and torch.equal(y16n[6:], y16[6:])


@pytest.mark.parametrize("cosine", [True, False])
def test_knn_sweep_more_shots_than_a_kernel_row(cosine):
    """S > 4096 (an LVIS-sized shot set; the reference has no cap): per-block top tens (lvc_knn_topk_candidates) + one merge /
    vote launch (lvc_knn_merge_vote) instead of the 4096-wide top-k kernels; same answers as the oracle -- and as a stable
    descending sort of the very same similarity matrix, exactly (the reference's tie rule at every step)."""
    from lvc_amd.label_verification import knn_sweep
    from oracle import knn as oknn

    g = torch.Generator().manual_seed(9)
    S, Dm, Q = 5000, 128, 700
    classes = torch.sort(torch.randint(0, 200, (S,), generator=g))[0]
    centers = torch.randn(200, Dm, generator=g)
    shots = centers[classes] + 1.5 * torch.randn(S, Dm, generator=g) + 0.3
    qcls = torch.randint(0, 200, (Q,), generator=g)
    q = centers[qcls] + 2.0 * torch.randn(Q, Dm, generator=g) + 0.3
    top, keep = knn_sweep(classes.to(D), shots.to(D), q.to(D), qcls.to(D), 10, cosine)
    ref_top = oknn.dense(classes, shots, q, cosine)
    ref_keep = oknn.get_nn_class_confirmatory(ref_top, qcls, 10)
    assert (top.cpu() != ref_top).any(dim=1).float().mean() <= 3e-3
    assert (keep.cpu() != ref_keep).float().mean() <= 3e-3
    # the block / merge kernels against a stable sort of ONE similarity matrix, with planted exact ties across blocks
    from lvc_amd import kernels as K

    sims = torch.randn(Q, 12300, generator=g)
    sims[:, 9000] = sims[:, 17]           # ties between blocks: the lower shot index must win
    sims[:50, 4100:4110] = 9.0            # ten equal maxima inside one block
    sims[50:60, ::1230] = 8.0             # ten equal values spread over the blocks
    cls = torch.randint(0, 1230, (12300,), generator=g)
    t2, k2 = K.knn_topk_vote_blocks(sims.to(D), 12300, cls.to(D), qcls.to(D), 10)
    order = torch.sort(sims, dim=1, descending=True, stable=True)[1][:, :10]
    assert torch.equal(t2.cpu(), cls[order])
    mode = torch.mode(cls[order], dim=1)[0]
    assert torch.equal(k2.cpu(), (mode == qcls).long())


def test_knn_sweep_pads_descriptor_width():
    """D % 32 != 0 (any descriptor network): zero columns are appended -- same ranking as the oracle on the unpadded rows."""
    from lvc_amd.label_verification import knn_sweep
    from oracle import knn as oknn

    g = torch.Generator().manual_seed(11)
    S, Dm, Q = 600, 100, 900
    classes = torch.sort(torch.randint(0, 40, (S,), generator=g))[0]
    centers = torch.randn(40, Dm, generator=g)
    shots = centers[classes] + 1.5 * torch.randn(S, Dm, generator=g) + 0.3
    qcls = torch.randint(0, 40, (Q,), generator=g)
    q = centers[qcls] + 2.0 * torch.randn(Q, Dm, generator=g) + 0.3
    for cosine in (True, False):
        top, keep = knn_sweep(classes.to(D), shots.to(D), q.to(D), qcls.to(D), 10, cosine)
        ref_top = oknn.dense(classes, shots, q, cosine)
        assert (top.cpu() != ref_top).any(dim=1).float().mean() <= 3e-3


@pytest.mark.parametrize("Dm", [64, 384, 1024])
def test_per_row_margins_bound_the_pre_filter_error(Dm):
    """`pre_filter_margins` (2 x the per-row bound from the rounding-residual norms `rownorm_h` measures) really bounds
    2 |fp16 pre-filter - fp64 product of the fp32 rows| for every (query, shot) pair -- random rows, rows with a few huge
    entries (the worst case of an fp16 rounding), and rows that are tiny after centring -- and is tighter than the worst case."""
    from lvc_amd import kernels as K
    from lvc_amd.label_verification import VERIFY_MARGIN, pre_filter_margins

    g = torch.Generator().manual_seed(Dm)
    S, Q = 700, 2000
    shots = torch.randn(S, Dm, generator=g) + 0.3
    q = torch.randn(Q, Dm, generator=g) + 0.3
    q[:50] = 0.3
    q[:50, :3] += 40.0 * torch.randn(50, 3, generator=g)          # three dominant coordinates
    q[50:60] = 0.3 + 1e-6 * torch.randn(10, Dm, generator=g)        # nearly the mean
    shots[:20, 5] += 30.0
    shots, q = shots.to(D), q.to(D)
    mu = K.colmean(shots)
    sn, sh, _, sres = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1, want_resid=True)
    qn, qh, _, qres = K.rownorm_h(q, mu=mu, eps=1e-8, mode=1, want_resid=True)
    assert torch.allclose(qres, (qn - qh.float()).norm(dim=1), rtol=1e-5, atol=1e-12)
    margins = pre_filter_margins(qres, sh.float().norm(dim=1).max(), sres.max(), Dm)
    # the sweep's one-launch form of the same numbers (lvc_knn_margins: the shots' largest fp16-row norm bounded by 1 + 1e-6 + max residual)
    from lvc_amd import kernels as K_
    from lvc_amd.label_verification import pre_filter_acc
    sres_max = K_.max_f32(sres)
    fused = K_.knn_margins(qres, sres_max, pre_filter_acc(Dm), 3.0e-5)
    torch.testing.assert_close(fused, pre_filter_margins(qres, 1.0 + 1e-6 + sres_max, sres_max, Dm) + 3.0e-5, rtol=2e-6, atol=0.0)
    err = (K.gemm_f16(qh, sh).double() - qn.double() @ sn.double().t()).abs().max(dim=1)[0]
    assert (2.0 * err <= margins.double()).all()
    assert margins.max().item() < VERIFY_MARGIN and margins.median().item() < 0.7 * VERIFY_MARGIN


def test_knn_sweep_edge_sizes():
    """No queries, one query, the minimum of ten shots, an odd number of shots, duplicate shots (exact ties -> lower index) and a
    zero query row (its normalised row is 0: every similarity ties at 0 -> the first ten shots)."""
    from lvc_amd.label_verification import knn_sweep
    from oracle import knn as oknn

    g = torch.Generator().manual_seed(2)
    Dm = 64
    for S, Q in [(10, 0), (10, 1), (11, 5), (333, 1), (64, 70)]:
        classes = torch.randint(0, 5, (S,), generator=g)
        shots = torch.randn(S, Dm, generator=g) + 0.2
        if S >= 64:
            shots[7] = shots[3]                                    # an exact duplicate
        q = torch.randn(Q, Dm, generator=g) + 0.2
        det = torch.randint(0, 5, (Q,), generator=g)
        if Q >= 5:
            q[2] = shots.mean(0)                                   # centred row = 0
            q[3] = shots[3]                                        # ties with the duplicate pair at the top
        top, keep = knn_sweep(classes.to(D), shots.to(D), q.to(D), det.to(D), 10, True)
        assert top.shape == (Q, 10) and keep.shape == (Q,)
        if Q:
            ref_top = oknn.dense(classes, shots, q, True)
            ref_keep = oknn.get_nn_class_confirmatory(ref_top, det, 10)
            rows = [i for i in range(Q) if not (Q >= 5 and i == 2)]     # the zero row: similarities are rounding noise around 0
            assert torch.equal(top.cpu()[rows], ref_top[rows]) and torch.equal(keep.cpu()[rows], ref_keep[rows])
