"""K12 (csrc/conv.hip::k_conv3x3_wgrad) against the library's weight-gradient kernels (torch.ops.aten.convolution_backward) at the
hidden blocks of the robot encoder and at the C4-equivariant policy's expanded filter banks; k_linear_wgrad against torch's
split-batch bmm.  ms per call and TFLOP/s of direct-convolution FLOPs.
    python tools/bench_wgrad.py [--batch 8192] [--size 128|84] [--equiv]"""
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
ap.add_argument("--equiv", action="store_true", help="the C4 policy's expanded banks (4x the channels) at --size 84")
ap.add_argument("--linear", action="store_true")
args = ap.parse_args()
B = args.batch
shapes = ([(16, 32, 64, 1), (32, 64, 32, 1), (64, 128, 16, 1), (128, 256, 8, 1), (256, 256, 8, 0)] if args.size == 128 else
          [(16, 32, 42, 1), (32, 64, 21, 1), (64, 128, 10, 1), (128, 256, 5, 0)])
if args.equiv:
    shapes = [(64, 128, 42, 1), (128, 256, 21, 1), (256, 512, 10, 1), (512, 512, 5, 0)]
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rows = []
if args.linear:
    for M, N, K in [(131072, 256, 256), (131072, 256, 64), (131072, 512, 512)]:
        gy, x = torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")
        flops = 2.0 * M * N * K
        t_k = timed(lambda: H.linear_wgrad(gy, x))
        t_t = timed(lambda: torch.bmm(gy.view(128, M // 128, -1).transpose(1, 2), x.view(128, M // 128, -1)).sum(0))
        rows.append(dict(M=M, N=N, K=K, k_ms=round(t_k, 3), torch_ms=round(t_t, 3), k_tflops=round(flops / t_k / 1e9, 1),
                         torch_tflops=round(flops / t_t / 1e9, 1)))
        print(rows[-1], file=sys.stderr, flush=True)
else:
    for Ci, Co, S, pad in shapes:
        x = torch.rand(B, Ci, S, S, device="cuda")
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        So = S + 2 * pad - 2
        g = torch.randn(B, Co, So, So, device="cuda")
        flops = 2.0 * B * So * So * Co * Ci * 9
        t_k = timed(lambda: H.conv3x3_wgrad(g, x, Co, pad))
        t_t = timed(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                                [False, True, False])[1])
        rows.append(dict(Ci=Ci, Co=Co, size=S, pad=pad, gflop=round(flops / 1e9, 1), k12_ms=round(t_k, 3), torch_ms=round(t_t, 3),
                         k12_tflops=round(flops / t_k / 1e9, 1), torch_tflops=round(flops / t_t / 1e9, 1)))
        print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps(dict(batch=B, size=args.size, equiv=args.equiv, rows=rows)))
