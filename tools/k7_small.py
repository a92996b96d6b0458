"""K7 launch duration versus minibatch size (hipEvent pair inside the library around the kernel): the intercept is
the fixed cost of a launch -- weight staging, first tile's latency, accumulator hand-over and slab write."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_mlp_fused import _setup
H, pol, bucket, obs, act, rec = _setup(128, 4096, 64, 6)
lay = H.mlp_layout(pol, bucket)
g = torch.zeros_like(bucket.flat_grad)
for M in (64, 2048, 15872, 31744, 63488, 131072):
    idx = torch.randperm(obs.shape[0], device="cuda")[:M].int()
    ts = []
    for _ in range(8):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        H.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, g, 0.2, 0.0, 0.5, events=ev)
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    ts.sort()
    print(f"M={M:7d} tiles={M // 32:5d}  K7 median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us")
