"""Race hunt for csrc/conv_bneck.hip: the same launch repeated must give bit-identical outputs (ring slots, the scratch inside t1's
space, the software pipelines' register reuse are all timing-sensitive if a wait or a barrier is missing), under a second stream's
traffic as well."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
torch.manual_seed(0)
def bn(c): return (torch.rand(c, device=d) + 0.5, torch.randn(c, device=d) * 0.1, torch.randn(c, device=d) * 0.1, torch.rand(c, device=d) + 0.5)
bad = 0
for proj, cin in ((False, 256), (True, 64)):
    w1 = torch.randn(64, cin, 1, 1, device=d) * (2.0 / cin) ** 0.5
    w2 = torch.randn(64, 64, 3, 3, device=d) * (2.0 / 576) ** 0.5
    w3 = torch.randn(256, 128 if proj else 64, 1, 1, device=d) * (2.0 / 64) ** 0.5
    bk = k.pack_bottleneck(k.pack_conv(w1, bn=bn(64)), k.pack_conv(w2, bn=bn(64), pad=1), k.pack_conv(w3, bn=None if proj else bn(256)), proj)
    for shape in ((8, 200, 336), (3, 77, 131), (1, 800, 1344 // 4)):
        x = torch.randn(*shape, cin, device=d).relu_()
        ref = k.bottleneck_fused(x, bk).clone()
        side = torch.cuda.Stream()
        junk = torch.randn(64 << 20, device=d)
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
        for i in range(n):
            if i % 2:
                with torch.cuda.stream(side):
                    junk.mul_(1.0001)          # HBM traffic next to the launch
            y = k.bottleneck_fused(x, bk)
            if not torch.equal(y, ref):
                bad += 1
                print("MISMATCH proj=%s shape=%s iter %d: %d elements, max |d| %.3e" % (proj, shape, i, int((y != ref).sum()), float((y - ref).abs().max())), flush=True)
        torch.cuda.synchronize()
        print("proj=%s shape=%s: %d launches bit-identical" % (proj, shape, n), flush=True)
print("conv error word", k.conv_error_word(d), "| mismatching launches:", bad)
sys.exit(1 if bad else 0)
