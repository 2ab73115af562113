import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
model = build_model(base_rcnn_fpn()).eval()
syn.conditioned_r50_fpn_(model)
dev = torch.device("cuda:0")
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]
def step():
    with torch.no_grad():
        return model.inference_batched(batch)
for _ in range(3): out = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): out = step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) * 100)
ref = [o.clone() for o in out]
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        for _ in range(2): step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            gout = step()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print("graph ms/step", (time.perf_counter() - t0) * 100)
    print("same outputs:", all(torch.equal(a, b) for a, b in zip(ref, gout)))
except Exception as e:
    import traceback; traceback.print_exc()
