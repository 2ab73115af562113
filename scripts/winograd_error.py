"""what Winograd would cost in accuracy on the 256-channel 3x3 layers -- measured, on the CPU, in the arithmetic the
kernels would use (fp32 transforms, fp32-accurate products accumulated in fp32).

  direct     : the 3x3 convolution evaluated in fp32 (torch CPU: the reference path's arithmetic)
  F(2,3) 1-D : Winograd along x only (4 positions per 2 output pixels, the three filter rows accumulated directly): 6 products per
               output instead of 9 -- MFMA count / 1.5
  F(2x2,3x3) : Winograd in both directions (16 positions per 2x2 outputs): 4 products per output -- MFMA count / 2.25
Every variant against the fp64 evaluation of the same operands; errors relative to the output's largest magnitude.  python scripts/winograd_error.py"""
import torch

torch.manual_seed(0)
torch.set_num_threads(8)
N, C, K, H, W = 2, 256, 256, 64, 96
x = torch.randn(N, C, H, W).relu_() * 1.3
w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
scale = float(ref.abs().max())


def report(name, y):
    e = (y.double() - ref).abs()
    print("%-12s max %.2e  rms %.2e  (of the output scale %.3g)" % (name, float(e.max()) / scale, float(e.pow(2).mean().sqrt()) / scale, scale))
    return float(e.pow(2).mean().sqrt()) / scale


d = report("direct fp32", torch.nn.functional.conv2d(x, w, padding=1))

# ---- F(2,3) along x: y[.., 2t + {0,1}] from d = x[.., 2t-1 .. 2t+2]
xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
d0, d1, d2, d3 = xp[..., 0:W:2], xp[..., 1:W + 1:2], xp[..., 2:W + 2:2], xp[..., 3:W + 3:2]      # [N,C,H+2,W/2] each
V = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
g0, g1, g2 = w[..., 0].double(), w[..., 1].double(), w[..., 2].double()                               # [K,C,3(rows)]
U = [g0.float(), ((g0 + g1 + g2) / 2).float(), ((g0 - g1 + g2) / 2).float(), g2.float()]
M = [torch.nn.functional.conv2d(V[p], U[p].unsqueeze(-1)) for p in range(4)]                          # rows by a (3 x 1) kernel: [N,K,H,W/2]
y = torch.empty(N, K, H, W)
y[..., 0::2] = M[0] + M[1] + M[2]
y[..., 1::2] = M[1] - M[2] - M[3]
e1 = report("F(2,3) 1-D", y)

# ---- F(2x2,3x3)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                              # [N,C,H/2,W/2,4,4]
Vt = torch.einsum("ij,ncabjk,lk->ncabil", Bt, tiles, Bt)                # fp32 transform
Ut = torch.einsum("ij,kcjl,ml->kcim", G, w.double(), G).float()         # exact transform, one rounding
Mt = torch.einsum("ncabil,kcil->nkabil", Vt, Ut)                         # fp32 accumulation over channels
Yt = torch.einsum("ij,nkabjl,ml->nkabim", At, Mt, At)                    # [N,K,H/2,W/2,2,2]
y2 = Yt.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)
e2 = report("F(2x2,3x3)", y2)
print("rms error relative to the direct fp32 evaluation: F(2,3) 1-D x %.2f, F(2x2,3x3) x %.2f" % (e1 / d, e2 / d))
