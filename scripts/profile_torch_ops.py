"""Which torch ops (plumbing) run inside one inference_batched step, with source lines."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
model = build_model(base_rcnn_fpn()).eval()
syn.conditioned_r50_fpn_(model)
dev = torch.device("cuda:0")
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]
for _ in range(3):
    with torch.no_grad(): model.inference_batched(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    with torch.no_grad(): model.inference_batched(batch)
    torch.cuda.synchronize()
import collections
agg = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::empty_strided", "aten::as_strided", "aten::view", "aten::select", "aten::slice", "aten::permute", "aten::reshape", "aten::_unsafe_view", "aten::unsqueeze", "aten::expand", "aten::t", "aten::transpose", "aten::stride", "aten::size", "aten::is_contiguous", "aten::contiguous", "aten::alias", "aten::detach", "aten::lift_fresh", "aten::resolve_conj", "aten::resolve_neg", "aten::item", "aten::_local_scalar_dense", "aten::to", "aten::empty_like", "aten::narrow", "aten::squeeze", "aten::flatten", "aten::unbind", "aten::split", "aten::chunk"):
        st = [s for s in (ev.stack or []) if "lvc_amd" in s]
        agg[(ev.name, st[0] if st else "?")] += 1
for (name, where), n in agg.most_common(40):
    print("%3d  %-28s %s" % (n, name, where[-110:]))
