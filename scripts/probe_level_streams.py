"""Would the FPN output convs / the RPN head gain from running the small pyramid levels on a second HIP stream beside p2?
Times conv3x3 256->256 on p2 followed by [p3, p4, p5] (the bench's maps, batch 8) on one stream, and with the small levels on a side
stream (fork / join by stream waits).  python scripts/probe_level_streams.py  (GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k

d = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
N = 8
maps = [(200, 336), (100, 168), (50, 84), (25, 42)]
xs = [torch.randn(N, h, w, 256, generator=g).to(d) for h, w in maps]
pcs = [k.pack_conv((torch.randn(256, 256, 3, 3, generator=g) * (2.0 / 2304) ** 0.5).to(d), bias=torch.zeros(256, device=d), pad=1) for _ in maps]
outs = [torch.empty(N, h, w, 256, device=d) for h, w in maps]
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def seq():
    k.conv2d_nhwc(xs[0], pcs[0], out=outs[0])
    k.conv3x3_levels(xs[1:], pcs[1:], outs=outs[1:])


def par():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        k.conv3x3_levels(xs[1:], pcs[1:], outs=outs[1:])
    k.conv2d_nhwc(xs[0], pcs[0], out=outs[0])
    main.wait_stream(side)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000.0


seq()
ref = [o.clone() for o in outs]
par()
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(ref, outs))
a = timeit(lambda: k.conv2d_nhwc(xs[0], pcs[0], out=outs[0]))
b = timeit(lambda: k.conv3x3_levels(xs[1:], pcs[1:], outs=outs[1:]))
for _ in range(3):
    s, p = timeit(seq), timeit(par)
    print("p2 alone %.1f us, p3..p5 alone %.1f us | one stream %.1f us, two streams %.1f us (%.1f %%), bit-identical %s" % (a, b, s, p, 100 * (p / s - 1), same))
