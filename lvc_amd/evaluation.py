"""Evaluation loop of the detector with batches in flight on several HIP streams.

The reference's `inference_on_dataset` (lvc/evaluation/evaluator.py:85-157) issues one batch, synchronises
(`torch.cuda.synchronize()` :124) and only then builds the next one.  On MI355X the tail of a batch -- RPN top-k, the two
NMS chains, candidate filtering: ~1.2 ms of latency-bound launches that occupy a handful of CUs -- leaves the chip mostly
idle, while the trunk of the NEXT batch could already run.  `PipelinedInference` issues batches round-robin on `depth`
streams (each with its own stream-K workspace, `kernels.conv_workspace`) and hands results back in submission order:
506 -> 541 img/s at depth 2 on one GPU (`scripts/probe_two_streams.py`), results bit-identical to the one-stream loop.

Why this cannot deadlock although every conv launch is a set of persistent workers that wait on each other: a worker only
ever waits for the FIRST work item of its successor, which the successor publishes before anything else; of the workers
resident at any moment all but the highest-numbered one of each launch therefore finish without outside help and free
their CU for whichever launch is next in either queue.
"""
import collections

import torch

from . import kernels as K
from .modeling.roi_heads.roi_heads import CandidateOverflow, instances_from_batched, widen_limits


def _limits(model):
    """The widenable limits a pass is launched under (`roi_heads.widen_limits`): (candidate-list capacity, operand split, how often
    layers have been re-routed to a wider range tier)."""
    heads = getattr(model, "roi_heads", None)
    return (getattr(heads, "det_max_candidates", None), K.CONV_SPLIT, K.RANGE_EPOCH)


class PipelinedInference:
    def __init__(self, model, depth=2):
        assert depth >= 1
        self.model = model
        self.streams = [torch.cuda.Stream(device=model.device) for _ in range(depth)]
        self._busy = [False] * depth   # one uncollected ticket per stream: the conv error word lives in the stream's workspace
        self._n = 0

    def submit(self, batched_inputs, do_postprocess=True, collectable=True):
        """Launch one batch; returns a ticket for `collect`.  Nothing is synchronised here.  A stream carries one
        uncollected batch at a time (its status / error words are per stream), so at most `depth` tickets are open.
        collectable=False: a throughput-only launch (timing loops) whose results are never read; no ticket."""
        k = self._n % len(self.streams)
        if self._busy[k]:      # either kind of launch would share the open ticket's status / range words
            raise RuntimeError("PipelinedInference: collect() the oldest ticket before submitting batch %d "
                               "(depth %d)" % (self._n, len(self.streams)))
        if not collectable:
            s = self.streams[k]
            self._n += 1
            s.wait_stream(torch.cuda.current_stream(self.model.device))
            with torch.cuda.stream(s), torch.no_grad():
                self.model.inference_batched(batched_inputs, do_postprocess)
            return None
        s = self.streams[k]
        self._n += 1
        self._busy[k] = True
        s.wait_stream(torch.cuda.current_stream(self.model.device))   # inputs produced on the caller's stream
        limits = _limits(self.model)
        with torch.cuda.stream(s), torch.no_grad():
            out = self.model.inference_batched(batched_inputs, do_postprocess)
        sizes = []
        for inp in batched_inputs:
            ref = inp["image"].shape[-2:] if "image" in inp else inp["raw"].shape[:2]
            sizes.append((inp.get("height", int(ref[0])), inp.get("width", int(ref[1]))))
        return (out, s, sizes, batched_inputs, do_postprocess, k, limits)

    def collect(self, ticket):
        """Wait for that batch only and build its `Instances` (the reference's per-image output dicts)."""
        (ob, osc, ocl, cnt, status), s, sizes, batched_inputs, do_postprocess, k, limits = ticket
        cur = torch.cuda.current_stream(self.model.device)
        rerun = False
        try:
            with torch.cuda.stream(s):
                try:
                    insts = instances_from_batched(ob, osc, ocl, cnt, sizes, status)   # one D2H read on that stream
                except (CandidateOverflow, K.Fp16RangeError) as e:
                    # a limit the reference does not have was hit by this batch.  Widen it (a no-op when an earlier ticket
                    # already did) and run the batch again if it was LAUNCHED under narrower limits than the current ones --
                    # batches submitted before the widening overflow one after the other and each needs its own re-run.
                    widen_limits(self.model, e)
                    if _limits(self.model) == limits:
                        raise
                    rerun = True
        finally:
            self._busy[k] = False      # whatever the read raised (timeout, status word): the stream is free for the next ticket
        # the results were allocated and written on the side stream: order the caller's stream behind it
        cur.wait_stream(s)
        if rerun:
            with torch.no_grad():   # on the caller's stream: the outputs then belong to it like any other result
                return self.model.inference(batched_inputs, do_postprocess=do_postprocess)
        for t in (ob, osc, ocl, cnt):   # tell the caching allocator that the caller's stream uses them too
            t.record_stream(cur)
        return [{"instances": r} for r in insts]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
            with torch.cuda.stream(s):
                K.check_conv_error_word(self.model.device)


class GraphedInference:
    """`inference_batched` captured once in a hipGraph and replayed: the ~150 kernel launches of a batch become one graph launch,
    so the host side of a step is a few microseconds whatever the Python around it costs -- what matters when eight ranks' launch
    loops share one host.  The forward qualifies as written: fixed shapes, no host read, no allocation outside torch's caching
    allocator (which gives the capture a private pool), per-batch constants cached on the device, the stream-K workspace owned
    by the capture stream.

        g = GraphedInference(model, batch)        # batch: list of {"image": CHW tensor, "height", "width"}, fixed sizes
        boxes, scores, classes, count, status = g.replay(next_batch)      # device tensors, overwritten by the next replay
        instances = g.instances()                 # one D2H read, as GeneralizedRCNN.inference does

    Outputs are bit-identical to the eager call (tests/test_gpu_pipeline.py).  A capacity / range condition (roi_heads.
    widen_limits) raised by `instances()` invalidates the graph: build a new one after the limits changed."""

    def __init__(self, model, batched_inputs, do_postprocess=True):
        assert all("image" in b for b in batched_inputs), "graphed inference takes preprocessed {'image': CHW} inputs"
        self.model = model
        dev = model.device
        self.static = []
        for b in batched_inputs:
            c = dict(b)
            c["image"] = b["image"].to(dev).clone().contiguous()
            self.static.append(c)
        self.sizes = []
        for inp in self.static:
            h, w = inp["image"].shape[-2:]
            self.sizes.append((inp.get("height", int(h)), inp.get("width", int(w))))
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(2):       # packs weights, fills the constant cache, creates this stream's workspace
                model.inference_batched(self.static, do_postprocess)
            self.stream.synchronize()
            K.check_conv_error_word(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.out = model.inference_batched(self.static, do_postprocess)
        torch.cuda.current_stream(dev).wait_stream(self.stream)

    def replay(self, batched_inputs=None):
        if batched_inputs is not None:
            assert len(batched_inputs) == len(self.static)
            for dst, src in zip(self.static, batched_inputs):
                assert src["image"].shape == dst["image"].shape and src["image"].dtype == dst["image"].dtype, "graph shapes are fixed"
                dst["image"].copy_(src["image"], non_blocking=True)
        self.graph.replay()
        return self.out

    def instances(self):
        """One D2H read of the counts, the status word and the conv kernels' error words -- of the CAPTURE stream's workspace
        (`kernels.conv_workspace` is per stream, and the graph baked that stream's pointer in)."""
        ob, osc, ocl, cnt, status = self.out
        cur = torch.cuda.current_stream(self.model.device)
        with torch.cuda.stream(self.stream):
            self.stream.wait_stream(cur)       # replay() was issued on the caller's stream
            try:
                insts = instances_from_batched(ob, osc, ocl, cnt, self.sizes, status)
            finally:
                cur.wait_stream(self.stream)
        return [{"instances": r} for r in insts]


def inference_on_dataset(model, data_loader, depth=2):
    """Yield (inputs, outputs) for every batch of `data_loader`, in order, keeping `depth` batches in flight."""
    pipe = PipelinedInference(model, depth)
    pending = collections.deque()
    for inputs in data_loader:
        if len(pending) == depth:      # the stream the next batch goes to still carries the oldest one
            i, t = pending.popleft()
            yield i, pipe.collect(t)
        pending.append((inputs, pipe.submit(inputs)))
    while pending:
        i, t = pending.popleft()
        yield i, pipe.collect(t)
    pipe.synchronize()
