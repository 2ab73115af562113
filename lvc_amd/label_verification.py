"""Label verification: mean-centred cosine kNN over crop descriptors + majority vote
(reference tools/run_nearest_neighbours.py: `assemble_tensors` :131-139, `run_nearest_neighbours`
:142-162, `get_nn_class_confirmatory` :214-227), as one batched device sweep.

`knn_sweep` is the MI355X path: all queries of all images in one [Q,D] tensor; the reference's
per-image Python loop and its [q_i,S,D] broadcast disappear.  `run_nearest_neighbours` /
`get_nn_class_confirmatory` keep the reference's list-of-dicts signatures on top of it.
"""
import os

import torch

from . import kernels as K

# LVC_KNN_TWO_STAGE=0: materialise the full-precision similarity matrix (three MFMAs per block) and rank it directly
KNN_TWO_STAGE = os.environ.get("LVC_KNN_TWO_STAGE", "1") != "0"
# LVC_KNN_ROW_MARGINS=0: the worst-case margin 2^-9 for every row instead of the per-row bound from the measured rounding residuals
KNN_ROW_MARGINS = True
# the pre-filter similarities as 16-bit fixed point (lvc_gemm_f16_q15: half the matrix round trip between the two stages);
# its quantisation error (half a step of 1 / 32766, twice: containment argument of csrc/knn.hip) joins every margin
KNN_Q15 = True
Q15_MARGIN = 2.0 * (0.5 / 32766.0) + 1e-6
# unit-norm rows: |fp16 dot - exact| <= 2^-11 (|q| rounding) + 2^-11 (|s| rounding) + 2^-22 + fp32 accumulation
# < 2^-10 (Cauchy-Schwarz on sum |q_i s_i|); the candidate window is twice that plus slack for the accumulation order
VERIFY_MARGIN = 2.0 ** -9 + 2.0 ** -16
TWO_STAGE_CHUNK = 1 << 18
MAX_SHOTS_PER_LAUNCH = 4096   # knn_topk_vote_kernel / knn_verify_topk_vote_kernel: 64 values per lane
QUERY_CHUNK = 32768  # rows of the similarity matrix materialised at once (x S x 4 bytes)


def preprocess_crops(data, mean, std):
    """reference tools/run_nearest_neighbours.py:95-99: concatenate the items' crops, (x - mean) / std per channel."""
    crops = torch.cat([x["instances"].crops for x in data])
    mean = torch.as_tensor(mean, dtype=torch.float32, device=crops.device).view(1, -1, 1, 1)
    std = torch.as_tensor(std, dtype=torch.float32, device=crops.device).view(1, -1, 1, 1)
    return (crops.float() - mean) / std


def get_descriptors(model, data_loader, pixel_mean, pixel_std):
    """reference get_descriptors (:102-128) over an iterable of one-image batches whose Instances carry `crops`
    ([n,3,224,224], lvc_amd.wire.get_crops_qe): normalise, run the descriptor network on the device, replace `crops` by
    `crop_feats` on the CPU, drop `image`.  Returns the list of items (deep copies are not needed: nothing else holds them)."""
    out = []
    with torch.no_grad():
        for data in data_loader:
            # raw crops go to the device; (x - mean) / std happens inside the network's patch gather (lvc_vit_patchify_norm)
            crops = torch.cat([x["instances"].crops for x in data]).to(model.device)
            feats = _descriptor_pass(model, crops, pixel_mean, pixel_std) if len(crops) else torch.zeros(0, model.embed_dim)
            item = data[0]
            item.pop("image", None)
            item["instances"].remove("crops")
            item["instances"].set("crop_feats", feats)
            item["instances"] = item["instances"].to("cpu")
            out.append(item)
    return out


def _descriptor_pass(model, crops, pixel_mean=None, pixel_std=None):
    """One batch of crops through the descriptor network.  Its Linear layers run on the fp16x2 split kernels, so the pass
    ends by reading the conv range word (as the detector passes do): an activation beyond fp16's range repeats the batch on
    the range-free bf16x3 kernels instead of returning invalid descriptors (and instead of leaving the bit for the next
    detector pass to trip over)."""
    from .modeling.roi_heads.roi_heads import run_with_fallbacks

    def once():
        feats = model(crops, pixel_mean, pixel_std) if pixel_mean is not None else model(crops)
        K.check_conv_error_word(crops.device)
        return feats

    return run_with_fallbacks(model, once)


def assemble_tensors(shot_features):
    """reference :131-139: concatenate shot classes/descriptors, sorted by class."""
    classes = torch.cat([x["instances"].gt_classes for x in shot_features])
    sorter = classes.argsort()
    desc = torch.cat([x["instances"].crop_feats for x in shot_features])
    return classes[sorter], desc[sorter]


def pre_filter_acc(D):
    """The accumulation term of `pre_filter_margins`."""
    return (2.0 * (D / 16.0 + 16.0) + 32.0) * 2.0 ** -24


def pre_filter_margins(qres, sh_max, sres_max, D):
    """margin [Q] = 2 x the error bound of the fp16 pre-filter for each query row against ANY shot:
        |q.s - q_h.s_h| = |(q - q_h).s_h + q.(s - s_h)| <= |q - q_h| |s_h| + |q| |s - s_h|      (Cauchy-Schwarz, twice)
    with the residual norms `rownorm_h` measured (qres per row, the shots' largest) and |q| <= 1 + 1e-5, plus the fp32
    accumulation of the two evaluations.  A sum of terms x_i is off by at most (roundings on a term's path) x (error per rounding)
    x sum |x_i|, here sum |q_i s_i| <= |q| |s| <= 1: a pre-filter term passes one 16-wide MFMA step (at most 16 additions inside,
    whatever their order) and the D / 16 - 1 accumulator updates behind it, each taken as 2^-23 (twice the round-to-nearest
    unit, so that a truncating adder is covered); the exact re-evaluation of csrc/knn.hip sums 16 fmas per lane and a six-level
    tree (22 roundings of 2^-24, taken as 32).  About half of the worst case 2^-9 (`VERIFY_MARGIN`) on real rows: fewer shots
    inside the window, fewer exact dot products."""
    acc = pre_filter_acc(D)
    eps = qres * sh_max + sres_max * (1.0 + 1e-5) + acc
    return (2.0 * (1.0 + 1e-4)) * eps


def knn_sweep(shot_classes, shot_descriptors, query_descriptors, detector_classes=None, k=10, cosine=True):
    """shot_descriptors [S,D], shot_classes [S] int64, query_descriptors [Q,D] (device, fp32; any S >= 10, any D).
    Returns (top10_shots [Q,10] int64 class ids, keep [Q] int64 or None)."""
    shots = shot_descriptors.contiguous().float()
    q = query_descriptors.contiguous().float()
    S, D = shots.shape
    Q = q.shape[0]
    if D % 32 != 0:
        # the GEMM kernels contract in chunks of 32: zero columns change neither dot products, norms nor the column mean
        pad = 32 - D % 32
        shots = torch.nn.functional.pad(shots, (0, pad))
        q = torch.nn.functional.pad(q, (0, pad))
        D += pad
    shot_classes = shot_classes.to(torch.int64).contiguous()
    large = S > MAX_SHOTS_PER_LAUNCH     # beyond the top-k kernels' row length (LVIS-sized shot sets): ranked in blocks + one merge launch
    two_stage = KNN_TWO_STAGE and cosine and D <= 2048 and S >= 10 and not large
    if cosine:
        mu = K.colmean(shots)
        if two_stage:
            sn, sh, _, sres = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1, want_resid=True)
            # pieces of the per-row error bound of the fp16 pre-filter (pre_filter_margins): the largest fp16-row norm and the
            # largest rounding-residual norm over the shots
            # |s_h| <= |s| + |s - s_h| = 1 + residual (the rows are unit vectors to 1e-7): no separate norm pass
            sres_max = K.max_f32(sres)
            sh_max = 1.0 + 1e-6 + sres_max
        else:
            sn = K.rownorm(shots, mu=mu, eps=1e-8, mode=1)
            pc = K.pack_linear(sn)
    else:
        # ranking by -cdist(q, s) == ranking by q.s - |s|^2/2 (|q| is constant per row)
        pc = K.pack_linear(shots)
        pc.shift = (-0.5 * (shots * shots).sum(1)).contiguous()
    tops, keeps = [], []
    # the two-stage path keeps only transient fp16 / fp32 copies per chunk: take as many rows as the GEMM's 2 GiB output allows
    chunk = min(TWO_STAGE_CHUNK, (2 ** 31 - 1) // (4 * S)) if two_stage else QUERY_CHUNK
    if large:
        chunk = max(256, min(QUERY_CHUNK, (2 ** 31 - 1) // (4 * S)))
    for s0 in range(0, max(Q, 1), chunk):
        qc = q[s0: s0 + chunk]
        if qc.shape[0] == 0:
            break
        dc = detector_classes[s0: s0 + chunk].to(torch.int64) if detector_classes is not None else None
        if two_stage:
            # fp16 similarities (one MFMA per block instead of three) as a pre-filter, exact fp32 re-evaluation of the few shots
            # that can reach the top ten (csrc/knn.hip: knn_verify_topk_vote_kernel states the containment argument)
            _, qh, den, qres = K.rownorm_h(qc, mu=mu, eps=1e-8, mode=1, want_rows=False, want_resid=True)
            extra = Q15_MARGIN if KNN_Q15 else 0.0
            margins = K.knn_margins(qres, sres_max, pre_filter_acc(D), extra)      # pre_filter_margins(qres, sh_max, sres_max, D) + extra, one launch
            t, kp = K.knn_verify_topk_vote(K.gemm_f16(qh, sh, q15=KNN_Q15), qc, sn, VERIFY_MARGIN + extra, shot_classes, dc, k, mu=mu, den=den,
                                           margins=margins if KNN_ROW_MARGINS else None)
        else:
            qn = K.rownorm(qc, mu=mu, eps=1e-8, mode=1) if cosine else qc
            sims = K.linear(qn, pc)
            if large:
                # shot sets beyond one launch's row length: per-block top tens + one merge / vote launch, all on the device
                t, kp = K.knn_topk_vote_blocks(sims, S, shot_classes, dc, k, block=MAX_SHOTS_PER_LAUNCH)
            else:
                t, kp = K.knn_topk_vote(sims, S, shot_classes, dc, k)
        tops.append(t)
        keeps.append(kp)
    if not tops:
        dev = q.device
        return torch.empty(0, 10, dtype=torch.int64, device=dev), (
            torch.empty(0, dtype=torch.int64, device=dev) if detector_classes is not None else None)
    if len(tops) == 1:     # the usual case: no copy of the results
        return tops[0], (keeps[0] if detector_classes is not None else None)
    return torch.cat(tops), (torch.cat(keeps) if detector_classes is not None else None)


def run_nearest_neighbours(shot_classes, shot_descriptors, query_features, cosine=True, device="cuda"):
    """Reference signature (:142-162): sets `top10_shots` on every item's Instances."""
    feats = [d["instances"].get("crop_feats") for d in query_features]
    counts = [len(f) for f in feats]
    if sum(counts):
        allq = torch.cat(feats).to(device)
        top, _ = knn_sweep(shot_classes.to(device), shot_descriptors.to(device), allq, None, 10, cosine)
        top = top.to(feats[0].device)
    else:
        top = torch.empty(0, 10, dtype=torch.int64)
    o = 0
    for d, n in zip(query_features, counts):
        d["instances"].set("top10_shots", top[o: o + n])
        o += n
    return query_features


def get_nn_class_confirmatory(query_features, k):
    """Reference signature (:214-227): keep[i] = 1 iff mode(top10_shots[i,:k]) == detector class
    (`gt_classes` field); ties in the vote resolve to the smallest class id (torch.mode)."""
    for d in query_features:
        inst = d["instances"]
        votes = inst.get("top10_shots")
        keep = torch.zeros(len(inst), dtype=torch.int64)
        if len(inst):
            nn_class = torch.mode(votes[:, :k].cpu(), dim=1)[0]
            keep = (nn_class == inst.gt_classes.cpu()).to(torch.int64)
        inst.set("keep", keep)


def knn_sweep_distributed(shot_classes, shot_descriptors, query_descriptors, detector_classes, k=10, cosine=True,
                          sweep=None):
    """Data-parallel form of the sweep (reference tools/run_nearest_neighbours.py:301-325): every rank contributes the
    shots it extracted -> one all-gather of the fp32 rows (the reference pickles them through a gloo group), each rank
    sweeps ITS queries, and the per-rank results (top10 class ids, keep) are gathered to rank 0 in rank order.
    Shot order: the reference concatenates the per-rank lists, each already class-sorted by `assemble_tensors`
    (`torch.cat`, :306-309), so equal similarities rank in (rank, class, position) order there; here the concatenation is
    re-sorted by class with a STABLE sort, so they rank in (class, rank, position) order.  The two orders differ only between
    shots of exactly equal similarity to a query, which leaves the class-id sequence -- the sweep's output -- unchanged
    unless the tied shots also belong to different classes (the reference's own order is unspecified there: unstable sort).  Returns (top10, keep) on rank 0, (None, None)
    elsewhere.  `sweep` defaults to `knn_sweep` (injectable so the collective plumbing can be tested on CPU/gloo)."""
    from . import distributed as D

    sweep = sweep or knn_sweep
    all_desc = D.all_gather_rows(shot_descriptors.contiguous())
    all_cls = D.all_gather_rows(shot_classes.contiguous())
    order = all_cls.argsort(stable=True)
    top, keep = sweep(all_cls[order], all_desc[order], query_descriptors, detector_classes, k, cosine)
    return D.gather_rows(top), D.gather_rows(keep)
