import os, sys, warnings, collections, traceback
sys.path.insert(0, os.getcwd())
import torch, bench
c = bench._setup()
import lvc_amd
from lvc_amd.utils.events import EventStorage
# build the cfg5 model as train_leg does, via a tiny copy of its setup
import types
src = bench.train_leg
# monkeypatch: capture step function by running train_leg with 1 step but sync debug on in the timed part
cnt = collections.Counter()
orig_warn = warnings.showwarning
def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        st = traceback.extract_stack()
        fr = [f for f in st if "/lvc_amd/" in f.filename or f.filename.endswith("bench.py")]
        key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1])
        cnt[key] += 1
warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
out = bench.train_leg(c, 2, 1, 2, "cfg5", phases=False)
torch.cuda.set_sync_debug_mode("default")
tot = sum(cnt.values())
print("syncs over 3 steps:", tot)
for k, v in cnt.most_common(40):
    print(v, k)
