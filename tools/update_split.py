"""Unprofiled split of one bench step on the main stream: advantage kernels (bootstrap value, GAE, record pack) versus
the update (permutation hand-over + graph replay), from events on the main stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
sys.argv = [sys.argv[0], "--cpu-baseline-updates", "0"]
args = bench.parse()
from aur_ppo_amd.ppo import ppo
hp = bench.hyper(args, 1)
hp["device"] = torch.device("cuda", 0)
agent = ppo(hp)
T, N = args.num_steps, agent.num_envs
data = bench.synth_buffers(T, N, args.obs_dim, args.act_dim, 1234)
for k in ("states", "actions", "values", "rewards", "terminals"):
    getattr(agent.buffer, k).copy_(data[k])
with torch.no_grad():
    _, lp, _, _ = agent.policy.evaluate(agent.buffer.states.view(-1, args.obs_dim), agent.buffer.actions.view(-1, args.act_dim))
    agent.buffer.log_probs.copy_(lp.view(T, N))
agent.seed_all(1)
nobs, ndone = data["next_obs"].cuda(), data["next_done"].cuda()
ev = []
for u in range(34):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    ret, adv = agent.advantages(nobs, ndone)
    e[1].record()
    agent.update(ret, adv)
    e[2].record()
    if u >= 4:
        ev.append(e)
torch.cuda.synchronize()
a = np.array([e[0].elapsed_time(e[1]) for e in ev]); b = np.array([e[1].elapsed_time(e[2]) for e in ev])
p = np.array([x[0].elapsed_time(y[0]) for x, y in zip(ev, ev[1:])])
print("period %.3f ms = advantages %.3f ms + update %.3f ms (medians); update / 16 minibatches = %.1f us" %
      (np.median(p), np.median(a), np.median(b), np.median(b) * 1e3 / 16))
