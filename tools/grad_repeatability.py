import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/aur_ppo_amd") else os.getcwd())
from aur_ppo_amd import hip_ops as H
from aur_ppo_amd.actor_critic import actor_critic
from aur_ppo_amd.flat import FlatBucket
hidden, layers = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
D, A, B, M = 64, 6, 524288, 131072
pol = actor_critic(D, (A,), hidden, layers, 0.0, True).cuda()
bucket = FlatBucket(pol.parameters())
lay = H.mlp_layout(pol, bucket)
g = torch.Generator(device="cuda").manual_seed(1)
obs = torch.randn(B, D, device="cuda", generator=g); act = torch.randn(B, A, device="cuda", generator=g)
rec = torch.randn(B, 4, device="cuda", generator=g); rec64 = H.pack_records(rec, act)
idx = torch.randperm(B, device="cuda")[:M].int()
side = torch.cuda.Stream(); junk = torch.zeros(1 << 22, device="cuda")
ref = None; worst = {}
for r in range(40):
    gout = torch.full_like(bucket.flat_grad, float("nan")); sc = torch.empty(9, device="cuda")
    if r % 2:
        with torch.cuda.stream(side):
            for _ in range(30): junk.add_(1.0)
    H.mlp_ppo_step(obs, None, rec64, idx, bucket.flat_param, lay, gout, 0.2, 0.0, 0.5, True, 1, sc)
    torch.cuda.synchronize()
    if ref is None:
        ref = (gout.clone(), sc.clone()); continue
    off = 0
    for name, p in pol.named_parameters():
        k = p.numel()
        d = float((gout[off:off+k] - ref[0][off:off+k]).abs().max()); s_ = float(ref[0][off:off+k].abs().max())
        worst[name] = max(worst.get(name, (0, 0))[0], d), s_
        off += k
    worst["scalars"] = max(worst.get("scalars", (0, 0))[0], float((sc - ref[1]).abs().max())), float(ref[1].abs().max())
for k, (d, s_) in worst.items(): print(f"{k:24s} max |g_r - g_0| = {d:.3e}   (max |g| {s_:.3e})")
