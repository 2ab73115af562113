"""Data-parallel plumbing of the hot path: one process per GPU, torch.distributed over RCCL/xGMI
(backend string "nccl" on ROCm) or gloo on CPU for tests.

What the reference does and what is kept:
  * inference: images shard contiguously per rank (detectron2/data/samplers/distributed_sampler.py:191-194),
    no collective in the forward; results are gathered to rank 0 only (`gather_rows`; lvc/evaluation/coco_evaluation.py:119-123);
  * kNN: shots all-gathered so that every rank holds all S shots (tools/run_nearest_neighbours.py:303-309; the
    reference pickles through a gloo group -- here it is ONE all_gather of the fp32 tensor), queries stay
    sharded, results gathered to rank 0 (:323-325);
  * fine-tune (cfg 3): DDP-style gradient all-reduce (mean) of the trainable tensors each step
    (lvc/engine/defaults.py:326-331).  cfg 3 trains 4 tensors / 0.41 MB, a pure-latency collective, so the
    gradients are flattened into one bucket = one all-reduce.  broadcast_buffers=False: FrozenBN buffers are
    never synced.
  * base training (faster_rcnn_R_50_FPN_base.yaml / cascade_ubbr_R_50_FPN_base.yaml: 72-82 tensors, ~165 MB of fp32
    gradients): `GradientBuckets` -- DDP's schedule re-stated for xGMI: parameters are bucketed in reverse
    registration order (the order backward produces them), a bucket's all-reduce is launched asynchronously from the
    post-accumulate hook of its last gradient, so the collectives of res5/FPN/heads run on RCCL's stream under the
    backward of res4/res3.  Buckets are 64 MB by default: a ring all-reduce over point-to-point xGMI links is bound
    per link (~153 GB/s), and a 64 MB bucket keeps each of the 2(N-1) ring steps above the ~1 MB where the link
    saturates at N=8, while 3 buckets still leave two of them fully hidden under backward.
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
    """Gather variable-length rows to `dst` ONLY (None elsewhere): the reference's `comm.gather` (detectron2/utils/comm.py:
    177-217 -- every rank pads to the largest length, the destination alone receives), without the pickle: one all-gather of
    the lengths, then one `dist.gather` of the padded tensors.  N - 1 of N ranks neither allocate nor receive the results."""
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
    if get_rank() == dst:
        out = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, out, dst=dst)
        return torch.cat([o[:s] for o, s in zip(out, sizes)], 0)
    dist.gather(pad, None, dst=dst)
    return None


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


def _reduce_scatter_sum(shard, flat):
    """shard <- this rank's 1/world slice of the sum of `flat` over ranks.  RCCL: one reduce_scatter_tensor (async handle
    returned).  gloo has no reduce-scatter: the same result as `world` rooted reductions of the slices (the rs_ag path then runs
    its padding / shard / all-gather logic on CPU test worlds exactly as under RCCL); returns None (already complete)."""
    world, rank = get_world_size(), get_rank()
    if dist.get_backend() != "gloo":
        return dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, async_op=True)
    n = shard.numel()
    for r in range(world):
        piece = flat[r * n: (r + 1) * n].clone()
        dist.reduce(piece, dst=r, op=dist.ReduceOp.SUM)
        if r == rank:
            shard.copy_(piece)
    return None


def _all_gather_shards(flat, shard):
    if dist.get_backend() != "gloo":
        dist.all_gather_into_tensor(flat, shard)
        return
    n = shard.numel()
    dist.all_gather([flat[r * n: (r + 1) * n] for r in range(get_world_size())], shard)


class GradientBuckets:
    """Bucketed, backward-overlapped gradient all-reduce (mean) for data-parallel training.

        buckets = GradientBuckets(model.parameters())
        loss.backward()          # hooks copy each gradient into its bucket; full buckets all-reduce asynchronously
        buckets.finish()         # wait, average; every p.grad is now a view of its (reduced) bucket
        optimizer.step()

    Collectives are issued in bucket order on every rank (a bucket that fills early waits for its predecessors), so
    ranks whose autograd engines finish gradients in a different order still agree on the sequence.  Parameters that
    receive no gradient in a step contribute zeros (DDP's find_unused_parameters behaviour, without the graph walk)."""

    def __init__(self, params, bucket_bytes=64 << 20, average=True, mode=None):
        """mode "all_reduce" (default; env LVC_GRAD_EXCHANGE overrides): one ring all-reduce per bucket.  mode "rs_ag": a
        reduce-scatter launched from the hook plus an all-gather in finish() -- the same bytes per link as the ring
        (2 (N-1)/N of the bucket) but as two collectives RCCL can spread over all seven xGMI links of a GPU, and the
        all-gather half moves out of the backward's shadow only when it has to (SURVEY.md section 5 sizes cfg 5's 361.5 MB
        for it).  No 8-GPU curve exists yet for either, so the default stays the plain ring."""
        import os

        self.mode = mode or os.environ.get("LVC_GRAD_EXCHANGE", "all_reduce")
        assert self.mode in ("all_reduce", "rs_ag"), self.mode
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        self.buckets = []
        self._where = {}
        cur, size = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self._close(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self._close(cur)
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        # deferred weight gradients (kernels.defer_wgrad) bypass AccumulateGrad: they are written straight into the bucket slice
        # (when the parameter has no .grad yet) and reported through the same hook
        from . import kernels as K

        for p in self.params:
            K.register_wgrad_sink(p, self._slot, self._hook, owner=self)
        self._next = 0
        self.bytes_reduced = 0

    def _slot(self, p):
        bi, off = self._where[id(p)]
        return self.buckets[bi]["flat"][off: off + p.numel()].view_as(p)

    def _close(self, plist):
        total = sum(p.numel() for p in plist)
        world = get_world_size()
        padded = (total + world - 1) // world * world if self.mode == "rs_ag" else total   # equal shards per rank
        flat = torch.zeros(padded, device=plist[0].device, dtype=plist[0].dtype)
        b = {"params": plist, "flat": flat, "pending": len(plist), "work": None, "launched": False, "seen": set(), "shard": None}
        o = 0
        for p in plist:
            self._where[id(p)] = (len(self.buckets), o)
            o += p.numel()
        self.buckets.append(b)

    def _hook(self, p):
        if p.grad is None:      # AccumulateGrad ran on an undefined gradient (a deferred weight gradient: it arrives through the sink)
            return
        from . import kernels as K

        if K.wgrad_queued(p):   # more of this parameter's gradient is still to come (another use's deferred job): not final yet
            return
        bi, off = self._where[id(p)]
        b = self.buckets[bi]
        view = b["flat"][off: off + p.numel()].view_as(p)
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view
        if id(p) not in b["seen"]:
            b["seen"].add(id(p))
            b["pending"] -= 1
        self._launch_ready()

    def _launch_ready(self, force=False):
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b["pending"] > 0 and not force:
                return
            if force and b["pending"] > 0:      # parameters unused this step: their slots are zeros
                for p in b["params"]:
                    if id(p) not in b["seen"]:
                        bi, off = self._where[id(p)]
                        b["flat"][off: off + p.numel()].zero_()
                        p.grad = b["flat"][off: off + p.numel()].view_as(p)
            if get_world_size() > 1 or (self.mode == "rs_ag" and dist.is_available() and dist.is_initialized()):
                if self.mode == "rs_ag":
                    world = get_world_size()
                    if b["shard"] is None:
                        b["shard"] = torch.empty(b["flat"].numel() // world, device=b["flat"].device, dtype=b["flat"].dtype)
                    b["work"] = _reduce_scatter_sum(b["shard"], b["flat"])
                    b["scattered"] = True
                else:
                    b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True)
            b["launched"] = True
            self.bytes_reduced += b["flat"].numel() * b["flat"].element_size()
            self._next += 1

    def finish(self):
        """Launch what is left, wait for every bucket, apply the mean; re-arm for the next step."""
        self._launch_ready(force=True)
        world = get_world_size()
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            if b.get("scattered"):
                b["scattered"] = False
                _all_gather_shards(b["flat"], b["shard"])
            if self.average and world > 1:
                b["flat"] /= world
            b["pending"], b["launched"] = len(b["params"]), False
            b["seen"].clear()
        self._next = 0
        n, self.bytes_reduced = self.bytes_reduced, 0
        return n

    def remove(self):
        from . import kernels as K

        for h in self._handles:
            h.remove()
        for p in self.params:
            K.unregister_wgrad_sink(p)
