"""Random-shape stress of K11 (csrc/conv.hip): forward and input gradient against the fp64 convolution, k_linear against the fp64 product.
    python tools/k11_fuzz.py [n_cases]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd import hip_ops as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
random.seed(11)
worst = 0.0
worst_lib = 0.0
for case in range(n):
    B = random.choice([1, 2, 3, 5, 8, 17, 64])
    Ci = 16 * random.randint(1, 8)
    Co = random.choice([8, 16, 24, 32, 48, 64, 96, 128, 160, 256])
    Hh, Ww = random.randint(1, 20), random.randint(1, 20)
    pad = random.choice([0, 1, 2])
    if Hh + 2 * pad - 2 <= 0 or Ww + 2 * pad - 2 <= 0:
        continue
    g = torch.Generator(device="cuda").manual_seed(case)
    x = torch.randn(B, Ci, Hh, Ww, device="cuda", generator=g, requires_grad=True)
    w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * 0.1
    os.environ["AURPPO_K11_DGRAD16"] = "1"          # K11 for every input gradient here, whatever its width
    z = H.conv3x3(x, w, pad)
    zr = torch.nn.functional.conv2d(x.detach().double(), w.double(), None, padding=pad)
    mag = torch.nn.functional.conv2d(x.detach().abs().double(), w.abs().double(), None, padding=pad).clamp_min(1e-30)
    e1 = ((z.double() - zr).abs() / mag).max().item()
    zt = torch.nn.functional.conv2d(x.detach(), w, None, padding=pad)          # the library's own fp32 convolution on the same metric
    worst_lib = max(worst_lib, ((zt.double() - zr).abs() / mag).max().item())
    e2 = 0.0
    if Co % 16 == 0:
        gz = torch.randn(z.shape, device="cuda", generator=g)
        z.backward(gz)
        xd = x.detach().double().requires_grad_(True)
        torch.nn.functional.conv2d(xd, w.double(), None, padding=pad).backward(gz.double())
        magx = torch.nn.functional.conv_transpose2d(gz.abs().double(), w.abs().double(), None, padding=pad).clamp_min(1e-30)
        e2 = ((x.grad.double() - xd.grad).abs() / magx).max().item()
    worst = max(worst, e1, e2)
    assert e1 <= 1e-6 and e2 <= 1e-6, (case, B, Ci, Co, Hh, Ww, pad, e1, e2)
for case in range(n // 2):
    M = random.choice([1, 31, 257, 4096, 40001])
    K = 16 * random.randint(1, 20)
    N = random.choice([1, 6, 16, 33, 64, 100, 256])
    g = torch.Generator(device="cuda").manual_seed(1000 + case)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.1
    y = H.linear_nobias(x, w, 0)
    e = ((y.double() - x.double() @ w.double().t()).abs() / (x.abs().double() @ w.abs().double().t()).clamp_min(1e-30)).max().item()
    worst = max(worst, e)
    assert e <= 1e-6, (case, M, K, N, e)
torch.cuda.synchronize()
print(f"k11_fuzz: {n} convolution cases + {n // 2} linear cases ok; worst error {worst:.3e} of sum|ab| (torch's fp32 conv2d on the same cases: {worst_lib:.3e})")
