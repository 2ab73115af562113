"""TEST INFRASTRUCTURE (oracle/): how far apart two evaluations of the detector may be, measured instead of set by hand.

The reference computes in fp32 (MODEL.DEVICE=cpu); the answer its outputs approximate is the fp64 evaluation of the same
weights.  `fp32_vs_fp64` runs the oracle's restatement of the reference path (oracle/rcnn.py, pinned against the reference's
outputs in tests/test_oracle_golden.py) in both precisions on the given inputs and returns the statistics of their difference
over the matched detections: the reference path's own rounding noise on THAT input.  `gate` then holds another evaluation (the
HIP path) to a small multiple of it -- medians, 90th percentiles, extremes AND the found fraction (round 6: nothing hand-set).  Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / timed_batch_parity
leg -- never by the product.
"""
import torch

LOOSE_BOX, LOOSE_SCORE = 0.1, 2e-3      # identity bar: which detection is which (near-tie flips in top-k / NMS fall outside)
K_NOISE = 2.0                           # median / p90 of |hip - reference| may be this multiple of the reference's own fp32-vs-fp64 noise (measured: 1.6 at 8 x 800x1333, BENCH_r04)
K_MAX = 3.0                             # the same for the LARGEST matched difference on the images the noise was measured on: the extreme of a
                                        # heavy-tailed sample, and of the difference of TWO fp32 evaluations (measured 1.0 - 2.6, BENCH_r05)
IDENT_MARGIN = 0.02                     # found fraction may fall this far below the fraction the reference path finds of its OWN fp64 detections
IDENT_K = 40.0                          # identity bars derived from the noise: this multiple of its median (R50: 40 x 2.5e-3 px = the 0.1 px above)
WIDE_BOX, WIDE_SCORE = 1.0, 5e-2        # bars of the first, generous matching those medians are measured with
FLOOR_BOX, FLOOR_SCORE = 2e-4, 2e-6     # fp32 resolution of a ~1000 px coordinate / of a score: no bar below it


def match_pairs(boxes, scores, classes, gboxes, gscores, gclasses, box_tol=LOOSE_BOX, score_tol=LOOSE_SCORE):
    """Greedy one-to-one matching of the detections (g*) to (boxes, scores, classes) of the same class within the loose bars.
    Returns the list of (|box| error, |score| error) of the matched pairs."""
    used, pairs = set(), []
    for i in range(len(gboxes)):
        db = (boxes - gboxes[i]).abs().max(dim=1)[0]
        ds = (scores - gscores[i]).abs()
        ok = (db <= box_tol) & (ds <= score_tol) & (classes == gclasses[i])
        for u in used:
            ok[u] = False
        idx = ok.nonzero().view(-1)
        if len(idx):
            j = int(idx[db[idx].argmin()])
            used.add(j)
            pairs.append((float(db[j]), float(ds[j])))
    return pairs


def _q(vals, q):
    if not vals:
        return 0.0
    v = sorted(vals)
    return v[min(len(v) - 1, int(q * len(v)))]


def deviation(dets_a, dets_b, box_tol=LOOSE_BOX, score_tol=LOOSE_SCORE):
    """dets_*: per image (boxes [n,4], scores [n], classes [n]) on the CPU; b is the side whose detections must be found in a.
    -> dict of counts and median / p90 / max of |box|, |score| differences over the pairs matched within the identity bars."""
    pairs, total, counts_equal = [], 0, True
    for (ab, asc, ac), (bb, bsc, bc) in zip(dets_a, dets_b):
        counts_equal &= len(asc) == len(bsc)
        total += len(bsc)
        pairs += match_pairs(ab.double(), asc.double(), ac.long(), bb.double(), bsc.double(), bc.long(), box_tol, score_tol)
    db, ds = [p[0] for p in pairs], [p[1] for p in pairs]
    return {"detections": total, "matched": len(pairs), "matched_fraction": len(pairs) / max(1, total), "counts_equal": bool(counts_equal),
            "box_median": _q(db, 0.5), "box_p90": _q(db, 0.9), "box_max": max(db, default=0.0),
            "score_median": _q(ds, 0.5), "score_p90": _q(ds, 0.9), "score_max": max(ds, default=0.0)}


def _dets(results):
    return [(r["pred_boxes"], r["scores"], r["pred_classes"]) for r in results]


def proposal_deviation(props_a, props_b, tol=LOOSE_BOX):
    """props_*: per image [n,4] proposal boxes (find_top_rpn_proposals' output).  -> the fraction of b's boxes that a holds within
    `tol` px, and the largest per-image difference of the counts: what the stage BEFORE the heads hands over (a few percent of the
    proposals lost barely move the 100 final detections of a full-size image, so the detections alone do not see it)."""
    found = total = 0
    dcount = 0
    for a, b in zip(props_a, props_b):
        dcount = max(dcount, abs(len(a) - len(b)))
        total += len(b)
        if len(a) and len(b):
            d = (b.double()[:, None, :] - a.double()[None, :, :]).abs().max(dim=2)[0].min(dim=1)[0]
            found += int((d <= tol).sum())
    return {"proposals": total, "found_fraction": found / max(1, total), "count_diff_max": dcount}


def fp32_vs_fp64(sd, spec, inputs, res32=None, box_tol=LOOSE_BOX, score_tol=LOOSE_SCORE, derive_identity=False, props32=None):
    """The oracle in fp32 (or `res32`, its precomputed fp32 results) against the oracle in fp64 on `inputs`.
    derive_identity: the identity bars are not given but MEASURED -- IDENT_K x the medians of a first matching with generous bars
    (never below LOOSE_*); returned as `box_tol` / `score_tol` of the result.  For weights whose noise floor is not R50's (R101)."""
    from . import rcnn as orc

    with torch.no_grad():
        if res32 is None:
            res32 = orc.generalized_rcnn_inference(sd, spec, inputs)
        sd64 = {k: v.double() for k, v in sd.items()}      # the oracle computes in the state_dict's dtype
        if props32 is not None:      # per image [n,4]: the fp32 run's proposals -> the same statistics one stage earlier
            res64, mid64 = orc.generalized_rcnn_inference(sd64, spec, inputs, return_intermediates=True)
            props64 = [p[0] for p in mid64["proposals"]]
            del mid64
        else:
            res64 = orc.generalized_rcnn_inference(sd64, spec, inputs)
    if derive_identity:
        wide = deviation(_dets(res32), _dets(res64), WIDE_BOX, WIDE_SCORE)
        box_tol = max(LOOSE_BOX, IDENT_K * wide["box_median"])
        score_tol = max(LOOSE_SCORE, IDENT_K * wide["score_median"])
    out = deviation(_dets(res32), _dets(res64), box_tol, score_tol)
    out["box_tol"], out["score_tol"] = box_tol, score_tol
    if props32 is not None:
        out["proposal_stage"] = proposal_deviation(props32, props64)
    return out


def gate(dev, noise, k=K_NOISE, dev_same=None, prop_dev=None):
    """dev = deviation(hip, reference fp32), noise = fp32_vs_fp64 on the same (or representative) inputs; dev_same: the deviation
    restricted to the images the noise was measured on (default: dev) -- the extremes of two samples compare only at equal size.
    -> (ok, bars, message).  Identity: equal counts, and the fraction of the reference's detections found within the identity bars
    at most IDENT_MARGIN below the fraction the reference path finds of its own fp64 detections (measured in the same run, not a
    hand-set floor).  Accuracy: median and p90 of the matched |box| / |score| differences within k x the reference path's own
    noise, the largest within K_MAX x its largest."""
    if dev_same is None:
        dev_same = dev
    bars = {"box_median": max(FLOOR_BOX, k * noise["box_median"]), "box_p90": max(FLOOR_BOX, k * noise["box_p90"]),
            "score_median": max(FLOOR_SCORE, k * noise["score_median"]), "score_p90": max(FLOOR_SCORE, k * noise["score_p90"]),
            "box_max": max(FLOOR_BOX, K_MAX * noise["box_max"]), "score_max": max(FLOOR_SCORE, K_MAX * noise["score_max"]),
            "matched_fraction": noise["matched_fraction"] - IDENT_MARGIN}
    bad = []
    if not dev["counts_equal"]:
        bad.append("detection counts differ")
    if dev["matched_fraction"] < bars["matched_fraction"]:
        bad.append("only %.1f %% of the reference detections found within the identity bars (the reference path finds %.1f %% of its own "
                   "fp64 detections; bar %.1f %%)" % (100 * dev["matched_fraction"], 100 * noise["matched_fraction"], 100 * bars["matched_fraction"]))
    for key, bar in bars.items():
        if key == "matched_fraction":
            continue
        val = dev_same[key] if key.endswith("_max") else dev[key]
        if val > bar:
            mult = K_MAX if key.endswith("_max") else k
            bad.append("%s %.2e > %.2e (= %.0f x the reference path's own fp32-vs-fp64 %s %.2e)" % (key, val, bar, mult, key, noise[key]))
    if prop_dev is not None and "proposal_stage" in noise:
        # prop_dev = proposal_deviation(hip proposals, reference fp32 proposals) on the images the noise was measured on
        pn = noise["proposal_stage"]
        bars["proposals_found_fraction"] = pn["found_fraction"] - IDENT_MARGIN
        bars["proposals_count_diff_max"] = 2 * pn["count_diff_max"] + 2
        if prop_dev["found_fraction"] < bars["proposals_found_fraction"]:
            bad.append("only %.1f %% of the reference's proposals found within %.2g px (the reference path finds %.1f %% of its own fp64 proposals)"
                       % (100 * prop_dev["found_fraction"], LOOSE_BOX, 100 * pn["found_fraction"]))
        if prop_dev["count_diff_max"] > bars["proposals_count_diff_max"]:
            bad.append("proposal counts differ by %d (the reference path's fp32 and fp64 runs: by %d)" % (prop_dev["count_diff_max"], pn["count_diff_max"]))
    return not bad, bars, "; ".join(bad) if bad else "ok"
