import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd import hip_ops as H
n, E = 524288, 4
rng = H.MT19937(1, n)
out = torch.empty((E, n), dtype=torch.int32, device="cuda")
for _ in range(3):
    rng.shuffle_epochs(n, E, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    rng.shuffle_epochs(n, E, out=out)
torch.cuda.synchronize()
print("ms per 4-epoch shuffle:", (time.perf_counter() - t0) / 10 * 1e3)
