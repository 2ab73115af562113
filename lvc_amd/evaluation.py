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


class PipelinedInference:
    def __init__(self, model, depth=2):
        assert depth >= 1
        self.model = model
        self.streams = [torch.cuda.Stream(device=model.device) for _ in range(depth)]
        self._n = 0

    def submit(self, batched_inputs, do_postprocess=True):
        """Launch one batch; returns a ticket for `collect`.  Nothing is synchronised here."""
        s = self.streams[self._n % len(self.streams)]
        self._n += 1
        s.wait_stream(torch.cuda.current_stream(self.model.device))   # inputs produced on the caller's stream
        with torch.cuda.stream(s), torch.no_grad():
            out = self.model.inference_batched(batched_inputs, do_postprocess)
        sizes = []
        for inp in batched_inputs:
            ref = inp["image"].shape[-2:] if "image" in inp else inp["raw"].shape[:2]
            sizes.append((inp.get("height", int(ref[0])), inp.get("width", int(ref[1]))))
        return (out, s, sizes, batched_inputs, do_postprocess)

    def collect(self, ticket):
        """Wait for that batch only and build its `Instances` (the reference's per-image output dicts)."""
        (ob, osc, ocl, cnt, status), s, sizes, batched_inputs, do_postprocess = ticket
        with torch.cuda.stream(s):
            try:
                insts = instances_from_batched(ob, osc, ocl, cnt, sizes, status)   # one D2H read on that stream
            except (CandidateOverflow, K.Fp16RangeError) as e:
                # a limit the reference does not have was hit by this batch: widen it and run the batch again, here
                if not widen_limits(self.model, e):
                    raise
                with torch.no_grad():
                    return self.model.inference(batched_inputs, do_postprocess=do_postprocess)
        # the results were allocated and written on the side stream: order the caller's stream behind it and tell the
        # caching allocator that the caller's stream uses them too
        cur = torch.cuda.current_stream(self.model.device)
        cur.wait_stream(s)
        for t in (ob, osc, ocl, cnt):
            t.record_stream(cur)
        return [{"instances": r} for r in insts]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
            with torch.cuda.stream(s):
                K.check_conv_error_word(self.model.device)


def inference_on_dataset(model, data_loader, depth=2):
    """Yield (inputs, outputs) for every batch of `data_loader`, in order, keeping `depth` batches in flight."""
    pipe = PipelinedInference(model, depth)
    pending = collections.deque()
    for inputs in data_loader:
        pending.append((inputs, pipe.submit(inputs)))
        if len(pending) > depth:
            i, t = pending.popleft()
            yield i, pipe.collect(t)
    while pending:
        i, t = pending.popleft()
        yield i, pipe.collect(t)
    pipe.synchronize()
