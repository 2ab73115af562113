"""CPU emulation of the operand-split schemes of the conv/GEMM kernels against fp64 on layer-shaped data (K = 2304,
post-ReLU activations): fp32 with one rounding per 16-deep block (what an MFMA chain does), the six- and three-term bf16
splits, and the two-way fp16 split with 2^11 scaling (Ootomo & Yokota 2022) with two / three accumulators."""
import numpy as np, torch
torch.manual_seed(0)
M,N,K=96,96,2304
def run(xs, ws):
    x=(torch.randn(M,K)*xs).clamp_min(0)   # post-ReLU activations
    w=torch.randn(N,K)*ws
    ref=(x.double()@w.double().t())
    scale=ref.abs().max().item()
    def blocked_acc(parts_list):
        # parts_list: list of (A,B,scale) float64 exact products summed per 16-deep block into fp32 accumulator
        acc=torch.zeros(M,N,dtype=torch.float32)
        return acc
    def mfma(A_list,B_list):  # emulate: for each 16-block, acc = fp32(acc + sum over listed plane pairs of exact products)
        acc=torch.zeros(M,N,dtype=torch.float32)
        for k0 in range(0,K,16):
            for (A,B) in zip(A_list,B_list):
                p=(A[:,k0:k0+16].double()@B[:,k0:k0+16].double().t())
                acc=(acc.double()+p).float()
        return acc
    out={}
    # fp32 sequential-ish (torch CPU matmul)
    out['torch fp32 matmul']=(x@w.t())
    # fp32 blocked 16 (like fp32 MFMA chain)
    out['fp32, 1 rounding per 16-block']=mfma([x],[w])
    # bf16x3, 6 terms
    def split_bf(a):
        h=a.bfloat16().float(); r=a-h; m=r.bfloat16().float(); l=(r-m).bfloat16().float(); return h,m,l
    a1,a2,a3=split_bf(x); b1,b2,b3=split_bf(w)
    out['bf16x3 6 terms']=mfma([a1,a3,a2,a1,a2,a1],[b3,b1,b2,b2,b1,b1])
    out['bf16x3 3 terms']=mfma([a1,a2,a1],[b2,b1,b1])
    # fp16x2 with 2^11 scaling, 2 accumulators
    def split_h(a):
        h=a.half().float(); r=(a-h)*2048.0; m=r.half().float(); return h,m
    h1,h2=split_h(x); g1,g2=split_h(w)
    main=mfma([h1],[g1]); cross=mfma([h1,h2],[g2,g1])
    out['fp16x2 3 terms (2 acc)']=(main.double()+cross.double()/2048).float()
    low=mfma([h2],[g2])
    out['fp16x2 4 terms (3 acc)']=(main.double()+cross.double()/2048+low.double()/2048/2048).float()
    # fp16x2 without scaling, single accumulator (shows the subnormal problem)
    def split_h0(a):
        h=a.half().float(); m=(a-h).half().float(); return h,m
    u1,u2=split_h0(x); v1,v2=split_h0(w)
    out['fp16x2 unscaled 3 terms']=mfma([u1,u2,u1],[v2,v1,v1])
    print("x scale %.3g w scale %.3g  | ref max %.3g" % (xs, ws, scale))
    for k,v in out.items():
        e=(v.double()-ref).abs()
        print("   %-34s max err/scale %.2e   rms err/scale %.2e" % (k, e.max().item()/scale, e.pow(2).mean().sqrt().item()/scale))
run(1.0, 0.03)
run(30.0, 0.002)
run(0.01, 0.5)
