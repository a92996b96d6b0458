"""Which pipeline bounds the update in an unprofiled run: per update, when the side stream finished the NEXT update's
permutations (K2) relative to when the main stream finished THIS update.  Positive slack = the shuffles were ready
before they were needed."""
import os, sys, time
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
base = torch.cuda.Event(enable_timing=True)
rows = []
for u in range(14):
    if u == 4:
        base.record()
    ret, adv = agent.advantages(nobs, ndone)
    agent.update(ret, adv)                       # enqueues the prefetch of update u+1's permutations on the side stream
    e_main = torch.cuda.Event(enable_timing=True); e_main.record()
    e_side = torch.cuda.Event(enable_timing=True); e_side.record(agent._perm_stream)
    if u >= 4:
        rows.append((e_main, e_side))
torch.cuda.synchronize()
tm = np.array([base.elapsed_time(a) for a, _ in rows]); ts = np.array([base.elapsed_time(b) for _, b in rows])
print("update period (main stream) %.3f ms; permutations of the next update ready %.3f ms (median) before the main stream ends the current one"
      % (np.median(np.diff(tm)), np.median(tm - ts)))
