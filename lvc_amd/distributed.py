"""Data-parallel plumbing of the hot path: one process per GPU, torch.distributed over RCCL/xGMI
(backend string "nccl" on ROCm) or gloo on CPU for tests.

What the reference does and what is kept:
  * inference: images shard contiguously per rank (detectron2/data/samplers/distributed_sampler.py:191-194),
    no collective in the forward; results are gathered to rank 0 (lvc/evaluation/coco_evaluation.py:119-123);
  * kNN: shots all-gathered so that every rank holds all S shots (tools/run_nearest_neighbours.py:303-309; the
    reference pickles through a gloo group -- here it is ONE all_gather of the fp32 tensor), queries stay
    sharded, results gathered to rank 0 (:323-325);
  * fine-tune (cfg 3): DDP-style gradient all-reduce (mean) of the trainable tensors each step
    (lvc/engine/defaults.py:326-331).  cfg 3 trains 4 tensors / 0.41 MB, a pure-latency collective, so the
    gradients are flattened into one bucket = one all-reduce.  broadcast_buffers=False: FrozenBN buffers are
    never synced.
"""
import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_range(n, rank=None, world=None):
    """InferenceSampler rule: shard = ceil(n/world); rank r gets [r*shard, min((r+1)*shard, n))."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    shard = (n - 1) // world + 1 if n > 0 else 0
    begin = shard * rank
    return range(min(begin, n), min(shard * (rank + 1), n))


def all_gather_rows(t):
    """All-gather tensors that differ in dim 0 across ranks (shots of each rank) -> concatenation, rank order."""
    world = get_world_size()
    if world == 1:
        return t
    n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], 0)


def gather_rows(t, dst=0):
    """Gather variable-length rows to `dst` (None elsewhere)."""
    world = get_world_size()
    if world == 1:
        return t
    full = all_gather_rows(t)
    return full if get_rank() == dst else None


def allreduce_gradients_(params, average=True):
    """One flattened-bucket all-reduce of the gradients of `params` (in place); returns the bucket bytes."""
    params = [p for p in params if p.requires_grad and p.grad is not None]
    world = get_world_size()
    if not params:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
    o = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[o: o + n].view_as(p.grad))
        o += n
    return flat.numel() * flat.element_size()
