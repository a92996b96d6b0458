"""Time of the optimizer tail of a chained minibatch (k_adam_chain) with and without the next minibatch's advantage
partial sums riding in the same launch, inside a hipGraph of 64 back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_mlp_fused import _setup
H, pol, bucket, obs, act, rec = _setup(128, 4096, 64, 6)
lay = H.mlp_layout(pol, bucket)
rec64 = H.pack_records(rec, act)
n = bucket.flat_param.numel()
g = torch.randn(n, device="cuda") * 1e-3
m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
lr, t = torch.tensor([3e-4], device="cuda"), torch.ones(1, device="cuda")
norm = torch.zeros(1, device="cuda")
idx = torch.randperm(obs.shape[0], device="cuda")[:131072].int()
def run(with_stats):
    def body():
        for _ in range(64):
            H.mlp_ppo_apply(bucket.flat_param, g, m, v, lay, lr, t, 0.5, (0.9, 0.999), 1e-5, norm,
                            rec=rec64 if with_stats else None, next_idx=idx if with_stats else None)
    body(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 64)
    return min(ts)
print(f"k_adam_chain alone {run(False):.2f} us per launch; with the next slice's advantage sums {run(True):.2f} us")
