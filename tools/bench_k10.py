"""K10 (the encoder's first block in one kernel, csrc/pool.hip) at BASELINE config 3's minibatch (8192 x (1,128,128) -> 16 channels)
and config 5's shard minibatch (4096 x (3,84,84)), HIP-event timed, against its algorithmic bytes: forward reads the observation
and writes the pooled block + a byte mask per pooled element; backward reads the pooled gradient, the masks and the observation."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd import hip_ops as H
rows = []
for (B, Ci, S) in ((8192, 1, 128), (4096, 3, 84)):
    obs = torch.rand(B, Ci, S, S, device="cuda")
    st = (torch.rand(B, device="cuda") < 0.5).float()
    w = (0.3 * torch.randn(16, Ci + 1, 3, 3, device="cuda")).requires_grad_(True)
    b = torch.zeros(16, device="cuda", requires_grad=True)
    y = H.first_block(obs, st, w, b)
    dy = torch.randn_like(y)
    def fwd():
        with torch.no_grad():
            return H.first_block(obs, st, w, b)
    def bwd():
        return torch.autograd.grad(y, (w, b), dy, retain_graph=True)
    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 10
    P = B * 16 * (S // 2) * (S // 2)
    fb = 4 * obs.numel() + 5 * P            # obs read, y + mask written
    bb = 5 * P + 4 * obs.numel()            # dy + mask read, obs read (partials are negligible)
    rows.append({"shape": [B, Ci, S, S], "fwd_ms": round(res["fwd"], 3), "fwd_GBs": round(fb / res["fwd"] / 1e6, 1),
                 "fwd_gflop": round(2 * P * 4 * 9 * Ci / 1e9, 1), "bwd_ms": round(res["bwd"], 3), "bwd_GBs": round(bb / res["bwd"] / 1e6, 1)})
print(json.dumps(rows))
