"""K9 (bias + state plane + ReLU + 2x2 max-pool, csrc/pool.hip) at the four block shapes of the 128x128 encoder with
BASELINE config 3's minibatch (8192 rows), HIP-event timed, against its algorithmic bytes: forward reads X (+ the L2-resident
plane) and writes X/4 floats + X/4 mask bytes; backward reads dY (X/4 floats) + the mask and writes dX."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd import hip_ops as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rows = []
for (C, S, plane) in ((16, 128, True), (32, 64, False), (64, 32, False), (128, 16, False)):
    x = torch.randn(B, C, S, S, device="cuda")
    bias = torch.randn(C, device="cuda")
    sc = (torch.rand(B, device="cuda") < 0.5).float() if plane else None
    pl = torch.randn(1, C, S, S, device="cuda") if plane else None
    xg = x.requires_grad_(True)
    y = H.bias_relu_pool2(xg, bias, sc, pl)
    dy = torch.randn_like(y)
    def fwd():
        return H.bias_relu_pool2(x.detach(), bias, sc, pl)
    def bwd():
        (g,) = torch.autograd.grad(y, xg, dy, retain_graph=True)
        return g
    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 10
    X = B * C * S * S
    fb, bb = 4 * X + 5 * X / 4, 5 * X / 4 + 4 * X
    rows.append({"shape": [B, C, S, S], "fwd_ms": round(res["fwd"], 3), "fwd_GBs": round(fb / res["fwd"] / 1e6, 1),
                 "bwd_ms": round(res["bwd"], 3), "bwd_GBs": round(bb / res["bwd"] / 1e6, 1),
                 "note": "bwd of the first block includes k_weighted_batch_sum (reads dX once more)" if plane else ""})
    del x, xg, y, dy
    torch.cuda.empty_cache()
print(json.dumps(rows))
