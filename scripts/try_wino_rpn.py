"""Would the RPN head's 3x3 conv pass the post-trunk chain test (tests/test_gpu_chain.py: logits within 2 x the oracle's own fp32-vs-fp64
error, exact discrete decisions, final boxes / scores within 1e-3) on the Winograd kernel instead of the two-accumulator direct one?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lvc_amd import kernels as K
from lvc_amd.modeling.proposal_generator import rpn as R
import test_gpu_chain as T

mode = sys.argv[1] if len(sys.argv) > 1 else "wino"
if mode != "default":
    R.FUSE_PREDICTOR = False
    K._WINO_MIN_TILES = 1
    orig = T._r50
    def patched():
        m = orig()
        m.proposal_generator.rpn_head.conv.two_acc = False      # read by Conv2d.packed() when the layer is (re)packed: the model is fresh
        return m
    T._r50 = patched
for seeds, hw in (((1, 2, 3, 4, 5, 6, 7, 8), (800, 1333)), ((3, 4), (320, 480))):
    T.test_post_trunk_chain_exact_decisions_and_1e3(seeds, hw, 50)
print("PASSED", mode)
