"""Run smoke() / bench.py / pytest with a deliberate HOST-side perturbation of the product path (test infrastructure; the library variants
of scripts/build_variant.sh perturb the kernels): every parity gate must fail under it (scripts/perturbed_build_check.sh).

  python scripts/perturb_run.py post_topk950 smoke
  python scripts/perturb_run.py post_topk950 bench.py --steps 5 --warmup 2 --no-extras --no-live-pmc
  python scripts/perturb_run.py post_topk950 pytest tests/test_gpu_e2e.py -q -x

post_topk950: find_top_rpn_proposals keeps 950 instead of 1000 proposals per image (reference proposal_utils.py:13-118 with another
POST_NMS_TOPK_TEST): ~5 % of the proposals are dropped -- a regression of the kind "loses a few percent of the detections"."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
what, target, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
from lvc_amd import kernels as K

if what == "post_topk950":
    _orig = K.rpn_proposals

    def _fewer(logits, deltas, cell_anchors, strides, image_sizes, pre_nms_topk, post_nms_topk, *a, **kw):
        return _orig(logits, deltas, cell_anchors, strides, image_sizes, pre_nms_topk, (post_nms_topk * 95) // 100, *a, **kw)

    K.rpn_proposals = _fewer
else:
    raise SystemExit("unknown perturbation %r" % what)
if target == "smoke":
    import __graft_entry__ as g

    g.smoke()
elif target == "pytest":
    import pytest

    sys.exit(pytest.main(rest))
else:
    sys.argv = [target] + rest
    runpy.run_path(os.path.join(ROOT, target), run_name="__main__")
