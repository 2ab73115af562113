"""Shared test helpers: golden loading, the conditioned R50-FPN state_dict, detection matching."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gold(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" and z[k].ndim > 0 else z[k]) for k in z.files}


def r50_state_dict():
    """The conditioned synthetic R50-FPN weights every e2e fixture was generated with: shapes from the
    reference state_dict key fixture, values from lvc_amd.utils.synthetic, FrozenBN stats from the calibration
    fixture."""
    from lvc_amd.utils import synthetic as syn

    keys = gold("r50_fpn_state_dict_keys")
    template = {}
    for k, shp in zip(keys["keys"].tolist(), keys["shapes"].tolist()):
        template[k] = torch.zeros(eval(shp))
    from oracle import rcnn as orc

    for i, s in enumerate((32, 64, 128, 256, 512)):
        template["proposal_generator.anchor_generator.cell_anchors.%d" % i] = orc.generate_cell_anchors((s,), (0.5, 1.0, 2.0))
    calib = gold("r50_bn_calibration")
    return syn.conditioned_state_dict(template, seed=0, bn_calibration=calib)


def r101_state_dict():
    """Same for R101-FPN (tests/golden/r101_fpn_state_dict_keys.npz, r101_bn_calibration.npz: the weights of
    e2e_r101_fpn_small / e2e_r101_fpn_800x1333)."""
    from lvc_amd.utils import synthetic as syn
    from oracle import rcnn as orc

    keys = gold("r101_fpn_state_dict_keys")
    template = {k: torch.zeros(eval(shp)) for k, shp in zip(keys["keys"].tolist(), keys["shapes"].tolist())}
    for i, s in enumerate((32, 64, 128, 256, 512)):
        template["proposal_generator.anchor_generator.cell_anchors.%d" % i] = orc.generate_cell_anchors((s,), (0.5, 1.0, 2.0))
    return syn.conditioned_state_dict(template, seed=0, bn_calibration=gold("r101_bn_calibration"))


def found_bar(p0, n, margin=0.02):
    """Lower bar of a found fraction over n items when a valid fp32 evaluation of the same path (the oracle, fp32 vs fp64) finds the
    fraction p0 of its own exact answers: p0 - margin (oracle/noise.py IDENT_MARGIN) - two standard deviations of a count of n."""
    n = max(1, n)
    return p0 - margin - 2.0 * (max(p0 * (1.0 - p0), 1.0 / n) / n) ** 0.5


def match_detections(boxes, scores, classes, gboxes, gscores, gclasses, tol=1e-3):
    """Set-equality of detections within `tol` (order may differ where scores are closer than tol).
    Returns (ok, message)."""
    if len(boxes) != len(gboxes):
        return False, "count %d vs %d" % (len(boxes), len(gboxes))
    used = set()
    worst = 0.0
    for i in range(len(gboxes)):
        d = (boxes - gboxes[i]).abs().max(dim=1)[0] + (scores - gscores[i]).abs()
        d = d + (classes != gclasses[i]).float() * 1e6
        for u in used:
            d[u] = 1e9
        j = int(d.argmin())
        if float(d[j]) > 2 * tol:
            return False, "golden detection %d unmatched (best distance %g)" % (i, float(d[j]))
        worst = max(worst, float(d[j]))
        used.add(j)
    return True, "worst %g" % worst


def match_fraction(boxes, scores, classes, gboxes, gscores, gclasses, box_tol, score_tol):
    """Greedy one-to-one matching of golden detections to predictions; returns (fraction matched,
    worst box error among matched, worst score error among matched)."""
    used = set()
    matched, wb, ws = 0, 0.0, 0.0
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
            matched += 1
            wb, ws = max(wb, float(db[j])), max(ws, float(ds[j]))
    return matched / max(1, len(gboxes)), wb, ws
