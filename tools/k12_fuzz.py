"""Random-shape stress of K12 (csrc/conv.hip::k_conv3x3_wgrad) and k_linear_wgrad against fp64: any channel counts, maps whose width is
/ is not a multiple of 4 (both load paths), batches from one image to several slices of pixels, every padding; the library's own fp32
weight gradient on the same metric beside it.
    python tools/k12_fuzz.py [n_cases]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd import hip_ops as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
random.seed(12)
worst = worst_lib = 0.0
done = 0
for case in range(n):
    B = random.choice([1, 2, 3, 7, 16, 33, 128, 300])
    Ci = random.choice([1, 3, 8, 16, 24, 32, 64, 100, 128])
    Co = random.choice([1, 5, 16, 32, 64, 65, 96, 128, 200, 256])
    Hh, Ww = random.choice([3, 4, 5, 8, 10, 12, 16, 21]), random.choice([3, 4, 5, 8, 10, 12, 16, 21, 32])
    pad = random.choice([0, 1, 1, 2])
    Ho, Wo = Hh + 2 * pad - 2, Ww + 2 * pad - 2
    if Ho <= 0 or Wo <= 0 or B * Ci * Hh * Ww > 40_000_000 or B * Co * Ho * Wo > 40_000_000:
        continue
    g = torch.Generator(device="cuda").manual_seed(case)
    x = torch.randn(B, Ci, Hh, Ww, device="cuda", generator=g)
    gz = torch.randn(B, Co, Ho, Wo, device="cuda", generator=g)
    dw = H.conv3x3_wgrad(gz, x, Co, pad)
    def w64(xx, gg):
        wd = torch.zeros(Co, Ci, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv2d(xx.double(), wd, None, padding=pad).backward(gg.double())
        return wd.grad
    ref, mag = w64(x, gz), w64(x.abs(), gz.abs()).clamp_min(1e-30)
    e = ((dw.double() - ref).abs() / mag).max().item()
    lib = torch.ops.aten.convolution_backward(gz, x, torch.zeros(Co, Ci, 3, 3, device="cuda"), None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1]
    worst_lib = max(worst_lib, ((lib.double() - ref).abs() / mag).max().item())
    worst = max(worst, e)
    assert e <= 1e-6, (case, B, Ci, Co, Hh, Ww, pad, e)
    assert torch.equal(H.conv3x3_wgrad(gz, x, Co, pad), dw), ("not deterministic", case)
    done += 1
nl = 0
for case in range(n // 3):
    M = random.choice([1, 31, 257, 4096, 40001, 131072])
    N, K = 4 * random.randint(1, 80), 4 * random.randint(1, 80)
    g = torch.Generator(device="cuda").manual_seed(2000 + case)
    gy, x = torch.randn(M, N, device="cuda", generator=g), torch.randn(M, K, device="cuda", generator=g)
    dw = H.linear_wgrad(gy, x)
    e = ((dw.double() - gy.double().t() @ x.double()).abs() / (gy.abs().double().t() @ x.abs().double()).clamp_min(1e-30)).max().item()
    worst = max(worst, e)
    assert e <= 1e-6, (case, M, N, K, e)
    nl += 1
torch.cuda.synchronize()
print(f"k12_fuzz: {done} convolution weight-gradient cases + {nl} linear cases ok; worst error {worst:.3e} of sum|ab| (the library's fp32 "
      f"weight gradient on the same convolution cases: {worst_lib:.3e})")
