"""Chain parity at the tolerance north_star states (boxes / scores within 1e-3, discrete decisions exact), and
the fp64 argument carried to the OUTPUTS.

1. Post-trunk chain: p2..p6 are computed ONCE by the CPU oracle (the reference's own fp32 arithmetic) and the same
   tensors are handed to (a) the oracle's continuation and (b) the HIP path: RPN head -> per-level top-k -> decode ->
   batched NMS -> level assignment -> ROIAlign -> box head -> predictor -> softmax / decode / per-class NMS / top-100
   -> detector_postprocess.  With equal features every discrete decision must agree (proposal list in the same order,
   detection classes and source rows) and boxes / scores must agree to 1e-3 -- at the full 800x1333 size.
   (reference lvc/modeling/meta_arch/rcnn.py:177-322, proposal_utils.py:13-118, fast_rcnn.py:95-137)

2. End to end INCLUDING the 53-layer trunk the 1e-3 is below the fp32 determinacy of the reference itself: its CPU
   path is ~3e-2 px away from the fp64 evaluation of the same weights.  So the test measures, on the FINAL boxes and
   scores of matched detections, error(GPU, fp64) against error(CPU fp32, fp64) and asserts the GPU is no further
   from the exact answer than the reference is (R50 small + 800x1333, R101 small).

3. The fraction of reference detections the HIP path reproduces within the literal 1e-3 must reach the fraction the fp64
   evaluation itself reproduces (minus 2 points): tied to the measured fp64 evidence, not to a hand-set floor.
"""
import numpy as np
import pytest
import torch

from helpers import match_fraction, r50_state_dict

pytestmark = pytest.mark.gpu


def _r50():
    from test_gpu_e2e import _model

    return _model()


def _chain_gpu(model, feats_cpu, sizes, out_hw):
    """Everything after the trunk on the device, from the given NCHW CPU features."""
    from lvc_amd import kernels as K
    from lvc_amd.modeling.roi_heads.roi_heads import check_status

    dev = torch.device("cuda")
    feats = {k: v.permute(0, 2, 3, 1).contiguous().to(dev) for k, v in feats_cpu.items()}
    sizes_dev = torch.tensor([list(s) for s in sizes], dtype=torch.int32, device=dev)
    post = torch.tensor([[ow / w, oh / h, float(oh), float(ow)] for (h, w), (oh, ow) in zip(sizes, out_hw)],
                        dtype=torch.float32, device=dev)
    status = K.new_status(dev)
    with torch.no_grad():
        pboxes, plogits, pcount = model.proposal_generator.predict_proposals_batched(feats, sizes_dev)
        ob, osc, ocl, orow, cnt = model.roi_heads.forward_batched(feats, pboxes, pcount, sizes_dev, post=post, status=status)
    torch.cuda.synchronize()
    check_status(int(status.item()))
    K.check_conv_error_word(dev)
    return (pboxes.cpu(), plogits.cpu(), pcount.cpu().tolist(), ob.cpu(), osc.cpu(), ocl.cpu().long(), orow.cpu().long(),
            cnt.cpu().tolist())


def _near_tie_pairs(sorted_scores, tie):
    """Entries of a descending score list that sit within `tie` of a neighbour (each may legitimately trade places)."""
    s = sorted_scores.double()
    close = (s[:-1] - s[1:]).abs() <= tie
    flag = torch.zeros(len(s), dtype=torch.bool)
    flag[:-1] |= close
    flag[1:] |= close
    return int(flag.sum())


# seeds 1..8 at 800x1333 = the batch bench.py times (`timed_batch_parity` re-checks it end to end in every bench run)
@pytest.mark.parametrize("seeds,hw,depth", [((1, 2, 3, 4, 5, 6, 7, 8), (800, 1333), 50), ((3, 4), (320, 480), 50),
                                            ((1, 2), (800, 1333), 101)])
def test_post_trunk_chain_exact_decisions_and_1e3(seeds, hw, depth):
    """No hand-set allowances: every budget below is measured, in this test, on the ORACLE itself -- the same chain evaluated
    in fp64 from the same features says how far a correct fp32 evaluation may sit from the exact answer, and therefore how
    far two correct fp32 evaluations may sit from each other:
      * TIE (scores that may come out in either order) = 4 x the largest |fp32 - fp64| of the oracle's own logits / scores
        (two evaluations, each off by up to that much, in opposite directions, on both entries of a pair);
      * rows allowed to trade places = the entries of the oracle's list that HAVE a neighbour within TIE;
      * proposal boxes: |GPU - oracle fp32| <= 2 x the largest |oracle fp32 - oracle fp64| + 2 fp32 ulp at the image width
        (both are within that distance of the exact box);
      * final detections: the literal 1e-3 of north_star on scores and boxes."""
    from lvc_amd.utils import synthetic as syn
    from oracle import rcnn as orc

    if depth == 50:
        model = _r50()
        sd = r50_state_dict()
    else:       # R101-FPN (BASELINE config 5's depth): the same post-trunk path on the deeper trunk's features and weights
        from lvc_amd.config.presets import base_rcnn_fpn
        from lvc_amd.modeling import build_model

        model = build_model(base_rcnn_fpn(depth=depth)).eval()
        syn.conditioned_r50_fpn_(model, depth=depth)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    sd64 = {k: v.double() for k, v in sd.items()}
    spec = orc.RCNNSpec(depth=depth)
    inputs = [{"image": syn.synthetic_image(s, *hw), "height": hw[0], "width": hw[1]} for s in seeds]
    with torch.no_grad():
        imgs, sizes = orc.preprocess([b["image"] for b in inputs], spec.pixel_mean, spec.pixel_std, 32)
        feats = orc.fpn(sd, orc.resnet(sd, imgs, depth))
        ref, mid = orc.generalized_rcnn_inference(sd, spec, inputs, return_intermediates=True, feats=feats)
        ref64, mid64 = orc.generalized_rcnn_inference(sd64, spec, inputs, return_intermediates=True,
                                                      feats={k: v.double() for k, v in feats.items()})
    pb, pl, pc, ob, osc, ocl, orow, cnt = _chain_gpu(model, feats, sizes, [hw] * len(inputs))
    ulp = 2.0 * 2.0 ** -23 * 2048.0          # two fp32 ulp of a coordinate below 2048 px
    for i in range(len(inputs)):
        rb, rl = mid["proposals"][i]
        rb64, rl64 = mid64["proposals"][i]
        assert pc[i] == len(rb), "image %d: %d proposals vs %d" % (i, pc[i], len(rb))
        # the oracle against itself in fp64: match its two proposal lists generously (no order requirement: near-ties differ)
        n = min(len(rb), len(rb64))
        d = (rb[:n, None, :].double() - rb64[None, :n, :]).abs().max(dim=2)[0]
        j = d.argmin(dim=1)
        near = d[torch.arange(n), j] <= 0.05
        e_box = float(d[torch.arange(n), j][near].max())
        e_log = float((rl[:n].double() - rl64[j])[near].abs().max())
        tie = max(4.0 * e_log, 1e-6)
        box_tol = 2.0 * e_box + ulp
        budget = _near_tie_pairs(rl, tie)
        # Same proposals in the same order == identical top-k selection and identical NMS keep decisions, up to near-ties
        perm, moved = _tie_aware_order(pb[i, : pc[i]], pl[i, : pc[i]], rb, rl, box_tol, 1e-3, tie)
        dbox = (pb[i, : pc[i]] - rb[perm]).abs().max(dim=1)[0]
        dlog = (pl[i, : pc[i]] - rl[perm]).abs()
        print("image %d: %d proposals, identical set; oracle fp32-vs-fp64: |logit| %.2e |box| %.2e px -> TIE %.1e, box bar %.2e px, "
              "%d entries in near-ties; GPU: %d rows swapped, worst |box| %.2e px, worst |logit| %.2e"
              % (i, pc[i], e_log, e_box, tie, box_tol, budget, moved, float(dbox.max()), float(dlog.max())))
        assert moved <= budget
        assert float(dlog.max()) <= 2.0 * e_log + 1e-6
        r = ref[i]
        n = cnt[i]
        assert n == len(r["scores"]), "image %d: %d detections vs %d" % (i, n, len(r["scores"]))
        # detections: (class, source proposal) identical, in the oracle's order up to score near-ties
        r64 = ref64[i]
        m = min(len(r["scores"]), len(r64["scores"]))
        d64 = (r["pred_boxes"][:m, None, :].double() - r64["pred_boxes"][None, :m, :]).abs().max(dim=2)[0]
        j64 = d64.argmin(dim=1)
        ok64 = (d64[torch.arange(m), j64] <= 0.05) & (r["pred_classes"][:m] == r64["pred_classes"][j64])
        e_score = float((r["scores"][:m].double() - r64["scores"][j64])[ok64].abs().max())
        e_dbox = float(d64[torch.arange(m), j64][ok64].max())
        tie_s = max(4.0 * e_score, 1e-6)
        src = perm[orow[i, :n]]                     # the GPU's source rows in the oracle's proposal numbering
        # One more discrete decision sits between proposals and detections: the pyramid level a RoI is pooled from,
        # floor(4 + log2(sqrt(area) / 224)) (poolers.py:24-59).  A proposal whose sqrt(area) lies within the proposal-box bar of a
        # level boundary (112, 224, 448 px) may legitimately be pooled from either level by two correct evaluations -- its scores
        # then differ by ~1e-2 (likewise for the sampling grid below).  Detections of such proposals are taken out of BOTH lists (and counted) before the comparison.
        s32 = ((rb[:, 2] - rb[:, 0]) * (rb[:, 3] - rb[:, 1])).double().sqrt()
        edge = torch.zeros(len(rb), dtype=torch.bool)
        for bnd in (112.0, 224.0, 448.0):
            edge |= (s32 - bnd).abs() <= 2.0 * box_tol
        # ... and the adaptive sampling grid of ROIAlign, ceil(roi extent on the level's map / 7) samples per bin
        # (ROIAlign_cpu.cpp:141-146, sampling_ratio 0): an extent within the bar of a multiple of 7 feature pixels flips it
        lvl = torch.floor(4 + torch.log2(s32 / 224.0 + 1e-8)).clamp(2, 5)
        scale = 0.5 ** lvl
        for a, b in ((0, 2), (1, 3)):
            v = (rb[:, b] - rb[:, a]).double() * scale / 7.0
            edge |= (v - v.round()).abs() <= 2.0 * box_tol * scale / 7.0 + 1e-7
        kg, kr = ~edge[src], ~edge[r["rows"]]
        n_edge = int(edge.sum())
        # the SAME (class, source row) entries must be left on both sides (the lists used to be cut to a common length);
        # an entry present on one side only may only be one that trades places at the 100-detection cut-off, i.e. whose score is
        # within TIE of the last kept score
        set_g = {(int(c), int(q)) for c, q in zip(ocl[i, :n][kg].tolist(), src[kg].tolist())}
        set_r = {(int(c), int(q)) for c, q in zip(r["pred_classes"][kr].tolist(), r["rows"][kr].tolist())}
        cut = float(r["scores"][-1])
        for side, only, cls_l, row_l, sc_l in (("GPU", set_g - set_r, ocl[i, :n].tolist(), src.tolist(), osc[i, :n].tolist()),
                                               ("oracle", set_r - set_g, r["pred_classes"].tolist(), r["rows"].tolist(), r["scores"].tolist())):
            for c_, q_, s_ in zip(cls_l, row_l, sc_l):
                if (int(c_), int(q_)) in only:
                    assert abs(s_ - cut) <= tie_s, "%s-only detection (class %d, proposal %d, score %.6f) is not at the cut-off (%.6f)" % (side, c_, q_, s_, cut)
        common = set_g & set_r
        mg = torch.tensor([(int(c), int(q)) in common for c, q in zip(ocl[i, :n].tolist(), src.tolist())]) & kg
        mr = torch.tensor([(int(c), int(q)) in common for c, q in zip(r["pred_classes"].tolist(), r["rows"].tolist())]) & kr
        m2 = int(mg.sum())
        assert m2 == int(mr.sum())
        g_cl, g_src, g_sc, g_bx = ocl[i, :n][mg], src[mg], osc[i, :n][mg], ob[i, :n][mg]
        r_cl, r_rows, r_sc, r_bx = r["pred_classes"][mr], r["rows"][mr], r["scores"][mr], r["pred_boxes"][mr]
        # what was set aside is EXACTLY the detections whose source proposal is an edge proposal plus the cut-off trades
        # checked above ((class, row) pairs are unique in a detection list) -- no allowance
        aside_g, aside_r = int((~kg).sum()), int((~kr).sum())
        assert n - m2 == aside_g + len(set_g - set_r), (n, m2, aside_g, len(set_g - set_r))
        assert len(r["scores"]) - m2 == aside_r + len(set_r - set_g), (len(r["scores"]), m2, aside_r, len(set_r - set_g))
        # and the edge proposals' detections can only change places with detections at the cut-off or with each other: the two
        # sides set aside the same number up to the entries that traded at the cut-off
        assert abs(aside_g - aside_r) <= len(set_g ^ set_r), (aside_g, aside_r, len(set_g ^ set_r))
        key_g = torch.stack([g_cl.double(), g_src.double()], 1)
        key_r = torch.stack([r_cl.double(), r_rows.double()], 1)
        dperm, dmoved = _tie_aware_order(key_g, g_sc, key_r, r_sc, 0.0, 1e-3, tie_s)
        ds = float((g_sc - r_sc[dperm]).abs().max())
        db = float((g_bx - r_bx[dperm]).abs().max())
        dbudget = _near_tie_pairs(r_sc, tie_s)
        if n_edge:
            print("image %d: %d proposals within %.1e px of a pooling-level or sampling-grid boundary; %d of %d detections set aside" % (i, n_edge, 2 * box_tol, n - m2, n))
        print("image %d: %d detections, classes and source rows identical; oracle fp32-vs-fp64 |score| %.2e |box| %.2e px -> TIE %.1e, "
              "%d entries in near-ties; GPU: %d swapped, worst |score| %.2e, worst |box| %.2e px"
              % (i, n, e_score, e_dbox, tie_s, dbudget, dmoved, ds, db))
        assert dmoved <= dbudget
        # north_star's literal 1e-3 -- or, where the reference's own fp32 evaluation sits further than 5e-4 from the exact box,
        # twice that distance (both evaluations are within it of the truth)
        assert ds <= 1e-3
        assert db <= max(1e-3, 2.0 * e_dbox + ulp), (db, e_dbox)


def _tie_aware_order(vals, scores, ref_vals, ref_scores, val_tol, score_tol, tie):
    """Row j of (vals, scores) must be row perm[j] of the reference with |vals| <= val_tol and |score| <= score_tol,
    perm a permutation that only moves entries whose REFERENCE scores are within `tie` of the entry they displace.
    Returns (perm [n] int64, number of moved rows); asserts otherwise."""
    n = len(ref_scores)
    assert len(scores) == n
    vals, ref_vals = vals.double().view(n, -1), ref_vals.double().view(n, -1)
    perm = torch.full((n,), -1, dtype=torch.int64)
    taken = torch.zeros(n, dtype=torch.bool)
    for j in range(n):
        lo, hi = max(0, j - 8), min(n, j + 9)
        d = (ref_vals[lo:hi] - vals[j]).abs().max(dim=1)[0]
        ok = (d <= val_tol) & ((ref_scores[lo:hi].double() - float(scores[j])).abs() <= score_tol) & ~taken[lo:hi]
        idx = ok.nonzero().view(-1)
        assert len(idx), ("row %d has no counterpart near its position in the reference list (best distance %.3e); window [%d, %d): "
                          "distances %s, |score diffs| %s, taken %s" % (j, float(d.min()), lo, hi, ["%.2e" % v for v in d.tolist()],
                                                                      ["%.2e" % v for v in (ref_scores[lo:hi].double() - float(scores[j])).abs().tolist()],
                                                                      taken[lo:hi].tolist()))
        r = lo + int(idx[(idx + lo - j).abs().argmin()])
        perm[j] = r
        taken[r] = True
    moved = (perm != torch.arange(n)).nonzero().view(-1)
    for j in moved.tolist():
        gap = abs(float(ref_scores[perm[j]]) - float(ref_scores[j]))
        assert gap <= tie, "row %d is reference row %d, but the reference scores differ by %.3e (not a near-tie)" % (j, int(perm[j]), gap)
    return perm, len(moved)


def _matched_errors(b, s, c, b64, s64, c64):
    """Greedy one-to-one matching (same class, nearest box within 1 px) of a result to the fp64 result; returns the
    per-detection box / score errors of the matched pairs."""
    used, eb, es = set(), [], []
    b, s = b.double(), s.double()
    for i in range(len(b64)):
        d = (b - b64[i]).abs().max(dim=1)[0]
        d = torch.where(c == c64[i], d, torch.full_like(d, 1e9))
        for u in used:
            d[u] = 1e9
        j = int(d.argmin()) if len(d) else -1
        if j >= 0 and float(d[j]) <= 1.0:
            used.add(j)
            eb.append(float(d[j]))
            es.append(abs(float(s[j] - s64[i])))
    return np.array(eb), np.array(es)


def _fp64_compare(model, sd, spec, inputs, tag, loose=(0.1, 2e-3)):
    from oracle import rcnn as orc

    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        r32 = orc.generalized_rcnn_inference(sd, spec, inputs)
        r64 = orc.generalized_rcnn_inference(sd64, spec, inputs)
        out = model(inputs)
    for i in range(len(inputs)):
        g = out[i]["instances"].to("cpu")
        e_gb, e_gs = _matched_errors(g.pred_boxes.tensor, g.scores, g.pred_classes, r64[i]["pred_boxes"], r64[i]["scores"],
                                     r64[i]["pred_classes"])
        e_cb, e_cs = _matched_errors(r32[i]["pred_boxes"], r32[i]["scores"], r32[i]["pred_classes"], r64[i]["pred_boxes"],
                                     r64[i]["scores"], r64[i]["pred_classes"])
        n64 = len(r64[i]["scores"])
        print("%s image %d (%d fp64 detections): matched gpu %d / cpu %d" % (tag, i, n64, len(e_gb), len(e_cb)))
        for name, eg, ec in (("box px", e_gb, e_cb), ("score", e_gs, e_cs)):
            print("   %-6s vs fp64: gpu median %.2e p90 %.2e max %.2e | reference cpu-fp32 median %.2e p90 %.2e max %.2e"
                  % (name, np.median(eg), np.percentile(eg, 90), eg.max(), np.median(ec), np.percentile(ec, 90), ec.max()))
        # the HIP path reproduces the exact (fp64) detections at least as completely and as closely as the reference's
        # own fp32 CPU path does (medians / 90th percentiles are robust to a single near-tie reordering)
        assert len(e_gb) >= len(e_cb) - max(2, n64 // 50)
        assert np.median(e_gb) <= 1.25 * np.median(e_cb) + 1e-6
        assert np.percentile(e_gb, 90) <= 1.5 * np.percentile(e_cb, 90) + 1e-6
        assert np.median(e_gs) <= 1.25 * np.median(e_cs) + 1e-8
        assert np.percentile(e_gs, 90) <= 1.5 * np.percentile(e_cs, 90) + 1e-8
        # The literal 1e-3 of north_star, end to end, against the EXACT answer: the HIP path must sit within 1e-3 (box px and
        # score, same class) of the fp64 detections at least as often as the reference's own fp32 CPU path does, minus the
        # sampling noise of that count.  (HIP-vs-reference distances contain BOTH paths' rounding errors, so the fraction of reference
        # detections the HIP path hits within 1e-3 is reported, not asserted: two evaluations that are each ~3e-3 px from the
        # truth are rarely within 1e-3 of each other.)
        b64, s64, c64 = r64[i]["pred_boxes"].float(), r64[i]["scores"].float(), r64[i]["pred_classes"]
        tight_g, _, _ = match_fraction(g.pred_boxes.tensor, g.scores, g.pred_classes, b64, s64, c64, box_tol=1e-3, score_tol=1e-3)
        tight_c, _, _ = match_fraction(r32[i]["pred_boxes"], r32[i]["scores"], r32[i]["pred_classes"], b64, s64, c64, box_tol=1e-3, score_tol=1e-3)
        tight_gc, _, _ = match_fraction(g.pred_boxes.tensor, g.scores, g.pred_classes, r32[i]["pred_boxes"], r32[i]["scores"],
                                        r32[i]["pred_classes"], box_tol=1e-3, score_tol=1e-3)
        loose_g, wb, ws = match_fraction(g.pred_boxes.tensor, g.scores, g.pred_classes, r32[i]["pred_boxes"], r32[i]["scores"],
                                         r32[i]["pred_classes"], box_tol=loose[0], score_tol=loose[1])
        print("   fp64 detections within 1e-3 of: the HIP path %.1f%%, the reference cpu-fp32 path %.1f%%; reference detections within 1e-3 "
              "of the HIP path %.1f%%, within %g px / %g %.1f%% (worst %.2e px, %.2e)"
              % (100 * tight_g, 100 * tight_c, 100 * tight_gc, loose[0], loose[1], 100 * loose_g, wb, ws))
        # (a count of n64 detections with hit probability ~tight_c: two standard deviations of that count, at least the 2 points)
        assert tight_g >= tight_c - max(0.02, 2.0 * (tight_c * (1.0 - tight_c) / max(n64, 1)) ** 0.5), (tight_g, tight_c, n64)
        # identity: the fraction the reference's fp32 path finds of the EXACT detections within the same bars, minus IDENT_MARGIN + sampling noise
        from helpers import found_bar
        loose_c, _, _ = match_fraction(r32[i]["pred_boxes"], r32[i]["scores"], r32[i]["pred_classes"], b64, s64, c64, box_tol=loose[0], score_tol=loose[1])
        assert loose_g >= found_bar(loose_c, n64), (loose_g, loose_c)


def test_final_outputs_vs_fp64_r50():
    from lvc_amd.utils import synthetic as syn
    from oracle import rcnn as orc

    model = _r50()
    sd = r50_state_dict()
    small = [{"image": syn.synthetic_image(3, 240, 320), "height": 480, "width": 640},
             {"image": syn.synthetic_image(4, 200, 352), "height": 200, "width": 352}]
    _fp64_compare(model, sd, orc.RCNNSpec(), small, "R50 small")
    full = [{"image": syn.synthetic_image(1), "height": 800, "width": 1333}]
    _fp64_compare(model, sd, orc.RCNNSpec(), full, "R50 800x1333")


def test_final_outputs_vs_fp64_r101():
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn
    from oracle import rcnn as orc

    model = build_model(base_rcnn_fpn(depth=101)).eval()
    syn.conditioned_r50_fpn_(model, depth=101)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    inputs = [{"image": syn.synthetic_image(3, 240, 320), "height": 480, "width": 640},
              {"image": syn.synthetic_image(4, 200, 352), "height": 200, "width": 352}]
    # R101's conditioned weights have a 5-6x higher fp32 noise floor than R50's: identity bars 5 x R50's (the found fraction is held to what the
    # reference's own fp32 path reaches within the SAME bars, so their size only decides which pairs count as the same detection)
    _fp64_compare(model, sd, orc.RCNNSpec(depth=101), inputs, "R101 small", loose=(0.5, 1e-2))


