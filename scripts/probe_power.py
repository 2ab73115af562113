"""Sample socket power and sclk (rocm-smi) while one conv shape runs in a loop: is the kernel power-limited?"""
import sys, os, subprocess, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as k
d = torch.device("cuda:0")
N, C, H, W, K, R, s, p = 8, 256, 200, 336, 256, 3, 1, 1
x = torch.randn(N, H, W, C, device=d)
w = torch.randn(K, C, R, R, device=d) * 0.02
pc = k.pack_conv(w, stride=s, pad=p)
samples = []
stop = False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        pw = re.search(r"Power \(W\): ([\d.]+)", out); sc = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        samples.append((time.time(), float(pw.group(1)) if pw else -1, int(sc.group(1)) if sc else -1))
for eng in sys.argv[1:] or ["bf16x3", "mfma_f32"]:
    k.CONV_ENGINE = eng
    y = k.conv2d_nhwc(x, pc, relu=True); torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 8.0:
        for _ in range(50): k.conv2d_nhwc(x, pc, relu=True, out=y)
        torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    stop = True; th.join()
    fl = 2.0 * N * H * W * K * C * R * R
    print("%s: %.3f ms/launch %.1f TF/s" % (eng, dt / n * 1e3, fl * n / dt / 1e12))
    print("   power W:", [int(s[1]) for s in samples]); print("   sclk MHz:", [s[2] for s in samples])
