"""CPU oracle of the label-verification sweep.  TEST INFRASTRUCTURE ONLY.
Restates tools/run_nearest_neighbours.py:142-162 (`run_nearest_neighbours`) and :214-227
(`get_nn_class_confirmatory`) on flat tensors; `dense` is the same math as one GEMM (used as the fair CPU
baseline: the reference's per-image broadcast formulation is ~500x slower than its arithmetic)."""
import torch
import torch.nn.functional as F


def run_nearest_neighbours(shot_classes, shots, queries, counts, cosine=True):
    """queries [Q,D] split into per-image chunks of `counts`; returns top10 class ids [Q,10]."""
    crop_mean = shots.mean(dim=0, keepdim=True)
    out, o = [], 0
    for n in counts:
        q = queries[o: o + n]
        o += n
        if cosine:
            sim = F.cosine_similarity(shots.sub(crop_mean).unsqueeze(0).float(), q.sub(crop_mean).unsqueeze(1).float(), dim=-1)
        else:
            sim = torch.cdist(shots.unsqueeze(0).float(), q.unsqueeze(0).float()).squeeze(0).t().mul(-1.0)
        out.append(shot_classes[sim.topk(10, dim=-1)[1]])
    return torch.cat(out) if out else torch.empty(0, 10, dtype=torch.int64)


def dense(shot_classes, shots, queries, cosine=True):
    mu = shots.mean(0, keepdim=True)
    if cosine:
        s = F.normalize(shots - mu, dim=1, eps=1e-8)
        q = F.normalize(queries - mu, dim=1, eps=1e-8)
        sim = q @ s.t()
    else:
        sim = queries @ shots.t() - 0.5 * (shots * shots).sum(1)[None]
    return shot_classes[sim.topk(10, dim=-1)[1]]


def get_nn_class_confirmatory(top10, det_classes, k):
    nn = torch.mode(top10[:, :k], dim=1)[0]
    return (nn == det_classes).long()
