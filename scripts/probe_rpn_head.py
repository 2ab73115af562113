"""RPN head on the batch's p2 / p3 maps: fused conv + predictors (one launch) vs conv launch + predictor launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K
from lvc_amd.modeling.proposal_generator.rpn import StandardRPNHead
torch.manual_seed(0)
head = StandardRPNHead(in_channels=256, num_anchors=3).to("cuda:0")
with torch.no_grad():
    head.conv.weight.normal_(0, 0.02)
for shape in [(8, 200, 336), (8, 100, 168)]:
    x = torch.randn(*shape, 256, device="cuda:0").abs()
    for fused in (True, False, True, False):
        K.RPN_FUSED_PRED = fused
        with torch.no_grad():
            for _ in range(3):
                head.forward_nhwc([x])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                head.forward_nhwc([x])
            e1.record(); torch.cuda.synchronize()
        print(shape, "fused" if fused else "plain", "%.3f ms" % (e0.elapsed_time(e1) / 10))
