"""The bench step with ONE frozen set of permutations (the shuffle streams idle) -- run it under tools/prof_stats.sh and compare
the main stream's kernel durations with profiles/rNN/bench_kernel_stats.csv to see which of them the side streams slow down."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
def step():
    ret, adv = agent.advantages(nobs, ndone)
    agent.update(ret, adv)
for _ in range(4): step()
frozen = agent._take_perms().clone()
torch.cuda.synchronize()
agent._take_perms = lambda: frozen
for _ in range(5): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize()
print(f"frozen permutations: {(time.perf_counter() - t) / 100 * 1e3:.4f} ms per step")
