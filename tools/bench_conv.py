"""K11 (csrc/conv.hip) against torch's conv2d (MIOpen's choice) at the hidden blocks of the robot encoder, minibatch 8192 (config 3)
and 4096 (config 5's shard): forward and input gradient, ms per call and TFLOP/s of direct-convolution FLOPs.
    python tools/bench_conv.py [--batch 8192]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd import _lib
if os.environ.get("AURPPO_LIB"):
    _lib.LIB_PATH = os.environ["AURPPO_LIB"]
from aur_ppo_amd import hip_ops as H
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--size", type=int, default=128, choices=(128, 84))
args = ap.parse_args()
B = args.batch
shapes = ([(16, 32, 64, 1), (32, 64, 32, 1), (64, 128, 16, 1), (128, 256, 8, 1), (256, 256, 8, 0)] if args.size == 128 else
          [(16, 32, 42, 1), (32, 64, 21, 1), (64, 128, 10, 1), (128, 256, 5, 0)])
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rows = []
for Ci, Co, S, pad in shapes:
    x = torch.rand(B, Ci, S, S, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
    So = S + 2 * pad - 2
    g = torch.randn(B, Co, So, So, device="cuda")
    flops = 2.0 * B * So * So * Co * Ci * 9
    t_k = timed(lambda: H.conv3x3(x, w, pad))
    t_t = timed(lambda: torch.nn.functional.conv2d(x, w, None, padding=pad))
    xr = x.clone().requires_grad_(True)
    def dgrad_k():
        z = H._Conv3x3.apply(xr, w, pad)
        return torch.autograd.grad(z, xr, g)[0]
    def dgrad_t():
        z = torch.nn.functional.conv2d(xr, w, None, padding=pad)
        return torch.autograd.grad(z, xr, g)[0]
    t_kb, t_tb = timed(dgrad_k) - t_k, timed(dgrad_t) - t_t
    rows.append(dict(Ci=Ci, Co=Co, size=S, pad=pad, gflop=round(flops / 1e9, 1), k11_fwd_ms=round(t_k, 3), torch_fwd_ms=round(t_t, 3),
                     k11_fwd_tflops=round(flops / t_k / 1e9, 1), torch_fwd_tflops=round(flops / t_t / 1e9, 1),
                     k11_dgrad_ms=round(t_kb, 3), torch_dgrad_ms=round(t_tb, 3)))
    print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps(dict(batch=B, size=args.size, rows=rows)))
