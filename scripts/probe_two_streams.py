"""Throughput of the detector with batches issued round-robin on S HIP streams (the tail of batch i -- top-k, NMS, the
RoI head's small launches -- overlaps the trunk of batch i+1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
import lvc_amd.kernels as K
model = build_model(base_rcnn_fpn()).eval()
syn.conditioned_r50_fpn_(model)
dev = torch.device("cuda:0")
batches = [[{"image": syn.synthetic_image(1 + (i + 8 * b) % 16).to(dev), "height": 800, "width": 1333} for i in range(8)] for b in range(2)]
def run(S, n=40):
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [None] * S
    def go(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % S]), torch.no_grad():
                outs[i % S] = model.inference_batched(batches[i % 2])
    go(2 * S + 2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    go(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for s in streams:
        with torch.cuda.stream(s):
            K.check_conv_error_word(dev)
    return 8 * n / dt, outs
ref = None
for S in (1, 2, 3):
    v, outs = run(S)
    cnt = [int(o[3].sum()) for o in outs if o is not None]
    print("streams %d: %.1f img/s   detections per batch %s" % (S, v, cnt), flush=True)
