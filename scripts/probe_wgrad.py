"""Per-layer timing of the weight-gradient kernel on the trainable R50-FPN layers (batch 2, 800x1344)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd import kernels as K
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
SPLIT = os.environ.get("WGRAD_SPLIT")   # f16x2: the fp16 MFMA form
L = [  # name, H, W (input), C, K, R, stride
    ("res3.0.conv1 s2", 200, 336, 256, 128, 1, 2), ("res3.conv2", 100, 168, 128, 128, 3, 1), ("res3.conv3", 100, 168, 128, 512, 1, 1),
    ("res3.conv1", 100, 168, 512, 128, 1, 1), ("res4.conv2", 50, 84, 256, 256, 3, 1), ("res4.conv3", 50, 84, 256, 1024, 1, 1),
    ("res4.conv1", 50, 84, 1024, 256, 1, 1), ("res5.conv2", 25, 42, 512, 512, 3, 1), ("res5.conv3", 25, 42, 512, 2048, 1, 1),
    ("fpn_lateral2", 200, 336, 256, 256, 1, 1), ("fpn_output2", 200, 336, 256, 256, 3, 1), ("fpn_output3", 100, 168, 256, 256, 3, 1),
    ("rpn.conv p2", 200, 336, 256, 256, 3, 1), ("rpn.pred p2", 200, 336, 256, 16, 1, 1), ("fc1", 1, 1, 12544, 1024, 1, 1),
]
tot = 0
for name, H, W, C, Kc, R, st in L:
    n = 1024 if name == "fc1" else N
    pad = 1 if R == 3 else 0
    Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    x = torch.randn(n, H, W, C, device=dev)
    dy = torch.randn(n, Ho, Wo, Kc, device=dev)
    for _ in range(2):
        K.conv_wgrad(x, dy, None, R, R, st, pad, split=SPLIT)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        K.conv_wgrad(x, dy, None, R, R, st, pad, split=SPLIT)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * n * Ho * Wo * Kc * C * R * R
    by = (x.numel() + dy.numel()) * 4
    print("%-16s %7.3f ms  %6.1f TF/s   algorithmic %6.1f MB -> %5.2f TB/s" % (name, ms, fl / ms / 1e9, by / 1e6, by / ms / 1e9))
    tot += ms
print("sum %.2f ms" % tot)
