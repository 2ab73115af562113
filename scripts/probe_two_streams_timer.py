import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
import lvc_amd.kernels as K
model = build_model(base_rcnn_fpn()).eval()
syn.conditioned_r50_fpn_(model)
dev = torch.device("cuda:0")
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]
for S in (1, 2):
    streams = [torch.cuda.Stream() for _ in range(S)]
    for i in range(6):
        with torch.cuda.stream(streams[i % S]), torch.no_grad(): model.inference_batched(batch)
    torch.cuda.synchronize()
    timer = K.LaunchTimer(only={"f16x2_halo"}); K.CONV_TIMER = timer
    t0 = time.perf_counter()
    for i in range(20):
        with torch.cuda.stream(streams[i % S]), torch.no_grad(): model.inference_batched(batch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    K.CONV_TIMER = None
    fl, ms, n = timer.flops_and_ms("f16x2_halo")
    print("streams %d: %.1f img/s; halo kernel %.1f TF/s, avg launch %.4f ms" % (S, 160 / dt, fl / ms / 1e9, ms / n))
