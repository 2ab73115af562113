import sys, os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys; sys.path.insert(0, %r)
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
for (N,C,H,W,K,R,s,p,name) in [(8,256,200,336,256,3,1,1,"p2 3x3"), (8,64,200,336,256,1,1,0,"res2 1x1 64->256")]:
    x = torch.randn(N,H,W,C, device=d); w = torch.randn(K,C,R,R, device=d)*0.02
    pc = k.pack_conv(w, stride=s, pad=p); y = k.conv2d_nhwc(x, pc, relu=True); torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): k.conv2d_nhwc(x, pc, relu=True, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/5
    print("   %%-18s %%7.3f ms %%7.1f TF/s" %% (name, ms, 2.0*N*H*W*K*C*R*R/ms/1e9))
''' % ROOT
so = os.path.join(ROOT, "lvc_amd", "liblvc_amd.so")
shutil.copy(so, so + ".orig")
try:
    for a in sys.argv[1:]:
        shutil.copy(os.path.join(ROOT, "build", "ablate", "lib_%s.so" % a), so)
        for per in ("2",):
            print("ABLATE=%s workers/CU=%s" % (a, per), flush=True)
            subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LVC_CONV_WORKERS_PER_CU=per))
finally:
    shutil.copy(so + ".orig", so)
