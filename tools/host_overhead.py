"""Host-side cost of the per-minibatch calls of the multi-process path (mlp_ppo_grad + mlp_ppo_apply), measured by
issuing them on a tiny minibatch without synchronising: the GPU work is shorter than the calls, so the loop runs at the
host's pace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_mlp_fused import _setup
H, pol, bucket, obs, act, rec = _setup(8, 64, 64, 6)
lay = H.mlp_layout(pol, bucket)
nb = bucket.flat_param.numel()
m, v, g = (torch.zeros(nb, device="cuda") for _ in range(3))
lr, t = torch.full((1,), 3e-4, device="cuda"), torch.zeros(1, device="cuda")
sc, nrm = torch.zeros(9, device="cuda"), torch.zeros(1, device="cuda")
idx = torch.randperm(512, device="cuda")[:64].int()
def pair():
    H.mlp_ppo_grad(obs, act, rec, idx, bucket.flat_param, lay, g, 0.2, 0.0, 0.5, True, 1, sc, t, chained=True)
    H.mlp_ppo_apply(bucket.flat_param, g, m, v, lay, lr, t, 0.5, (0.9, 0.999), 1e-5, nrm, grad_scale=0.5, rec=rec, next_idx=idx)
for _ in range(50):
    pair()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    pair()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host: {(t1 - t0) / 2000 * 1e6:.1f} us per grad+apply pair; with the final drain {(t2 - t0) / 2000 * 1e6:.1f} us")
