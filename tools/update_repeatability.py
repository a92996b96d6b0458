"""How far do two runs of the SAME update from the SAME state drift apart on the GPU?  K7 / K7w hand tiles to workgroups
through a counter, so the order in which a slab's sums are formed is not fixed from launch to launch; the gradients then
differ in their last bits, and Adam (sensitivity lr / eps = 30 to a gradient element far below eps) turns that into
weight differences.  Prints, over R repeats, the max |w_i - w_0| per parameter tensor and the gradient magnitude there.
    python tools/update_repeatability.py [--hidden-dim H --num-layers L --repeats R --busy]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # noqa: E402
from aur_ppo_amd.ppo import ppo                # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hidden-dim", type=int, default=64)
ap.add_argument("--num-layers", type=int, default=2)
ap.add_argument("--repeats", type=int, default=12)
ap.add_argument("--busy", action="store_true", help="keep another stream busy with small kernels during the update")
a = ap.parse_args()
args = argparse.Namespace(num_steps=128, envs_per_gpu=4096, epochs=4, minibatches=4, obs_dim=64, act_dim=6,
                          hidden_dim=a.hidden_dim, num_layers=a.num_layers)
hp = bench.hyper(args, 1)
hp["hip_graph"] = False
torch.manual_seed(1)
agent = ppo(hp)
data = bench.synth_buffers(128, 4096, 64, 6, 1234)
for k in ("states", "actions", "values", "rewards", "terminals"):
    getattr(agent.buffer, k).copy_(data[k])
with torch.no_grad():
    _, lp, _, _ = agent.policy.evaluate(agent.buffer.states.view(-1, 64), agent.buffer.actions.view(-1, 6))
    agent.buffer.log_probs.copy_(lp.view(128, 4096))
p0 = agent.bucket.flat_param.clone()
outs, grads = [], None
side = torch.cuda.Stream()
junk = torch.zeros(1 << 20, device="cuda")
for r in range(a.repeats):
    with torch.no_grad():
        agent.bucket.flat_param.copy_(p0)
        for t in (agent._adam_m, agent._adam_v, agent._adam_t):
            if t is not None:
                t.zero_()
    agent.seed_all(1)
    ret, adv = agent.advantages(data["next_obs"].cuda(), data["next_done"].cuda())
    if a.busy:
        with torch.cuda.stream(side):
            for _ in range(400 * (r % 3)):
                junk.add_(1.0)
    n = agent.update(ret, adv)
    torch.cuda.synchronize()
    outs.append(agent.bucket.flat_param.detach().cpu().clone())
    if grads is None:
        grads = agent.bucket.flat_grad.detach().cpu().clone()
off = 0
worst = 0.0
for name, p in agent.policy.named_parameters():
    k = p.numel()
    d = torch.stack([(o[off:off + k] - outs[0][off:off + k]).abs() for o in outs[1:]]).max(0).values
    i = int(d.argmax())
    worst = max(worst, float(d.max()))
    print(f"{name:24s} max |w_r - w_0| = {float(d.max()):.3e}  there: w = {float(outs[0][off + i]):+.3e}, last-step grad = {float(grads[off + i]):+.3e}")
    off += k
print(f"worst over {a.repeats} repeats: {worst:.3e}   (Adam: lr {hp['learning_rate']}, eps 1e-5, 16 steps)")
