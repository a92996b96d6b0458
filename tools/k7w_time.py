"""Time the K7w kernel alone (hipEvents around the main kernel) at the headline minibatch for one net shape:
    K7W_SHAPE="128,3,64" (hidden, layers, state_dim; default) [AURPPO_LIB=<other build>] python tools/k7w_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from aur_ppo_amd import _lib
if os.environ.get("AURPPO_LIB"):
    _lib.LIB_PATH = os.environ["AURPPO_LIB"]
from aur_ppo_amd import hip_ops as H
from aur_ppo_amd.actor_critic import actor_critic
from aur_ppo_amd.flat import FlatBucket
hidden, layers, D = (int(x) for x in os.environ.get("K7W_SHAPE", "128,3,64").split(","))
T, N, A, M = 128, 4096, 6, int(os.environ.get("K7_M", 131072))
torch.manual_seed(0)
pol = actor_critic(D, (A,), hidden, layers, 0.0, True).cuda()
bucket = FlatBucket(pol.parameters())
lay = H.mlp_layout(pol, bucket)
g = torch.Generator(device="cuda").manual_seed(1)
B = T * N
obs = torch.randn(B, D, device="cuda", generator=g)
act = torch.randn(B, A, device="cuda", generator=g)
rec = torch.stack([-4 + 0.2 * torch.randn(B, device="cuda", generator=g), 2 * torch.randn(B, device="cuda", generator=g),
                   torch.randn(B, device="cuda", generator=g), torch.randn(B, device="cuda", generator=g)], 1).contiguous()
rec64 = H.pack_records(rec, act)
idx = torch.randperm(B, device="cuda")[:M].int()
ts = []
for it in range(20):
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    H.mlp_ppo_step(obs, None, rec64, idx, bucket.flat_param, lay, bucket.flat_grad, 0.2, 0.0, 0.5, events=ev)
    torch.cuda.synchronize()
    if it >= 4:
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
print(f"{layers}x{hidden} D={D} wide={lay['wide']} M={M}: main kernel {np.median(ts):.1f} us (min {min(ts):.1f}, max {max(ts):.1f})")
