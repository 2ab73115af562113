import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lvc_amd.modeling.vit import seeded_state_dict_, vit_small
d = torch.device("cuda:0")
m = vit_small(8); seeded_state_dict_(m, 0); m = m.to(d).eval()
x = torch.rand(64, 3, 224, 224, device=d) * 255
mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
with torch.no_grad():
    for _ in range(2): m(x, mean, std)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m(x, mean, std)
    torch.cuda.synchronize(); print("ms per batch", (time.perf_counter() - t0) / 5 * 1e3)
