"""k_adam_chain with and without the next slice's statistics workgroups (next_idx named or not): run under tools/prof_stats.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_mlp_fused import _setup
Hh, pol, bucket, obs, act, rec = _setup(128, 4096, 64, 6)
lay = Hh.mlp_layout(pol, bucket)
M = 131072
perm = torch.randperm(obs.shape[0], device="cuda").int()
sl = [perm[k * M:(k + 1) * M] for k in range(4)]
rec64 = Hh.pack_records(rec, act)
nb = bucket.flat_param.numel()
m, v = torch.zeros(nb, device="cuda"), torch.zeros(nb, device="cuda")
lr, t = torch.full((1,), 3e-4, device="cuda"), torch.zeros(1, device="cuda")
sc, nrm = torch.zeros(9, device="cuda"), torch.zeros(1, device="cuda")
with_next = os.environ.get("WITH_NEXT", "1") == "1"
for rep in range(10):
    for k, idx in enumerate(sl):
        nxt = sl[(k + 1) % 4] if with_next else None
        Hh.mlp_ppo_minibatch(obs, None, rec64, idx, bucket.flat_param, lay, bucket.flat_grad, 0.2, 0.0, 0.5, True, 1, sc, m, v, lr, t,
                             0.5, (0.9, 0.999), 1e-5, nrm, next_idx=nxt, chained=with_next and (rep > 0 or k > 0))
torch.cuda.synchronize()
print("done", with_next)
