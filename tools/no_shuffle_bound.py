"""Upper bound of what the side stream's shuffle kernels cost the main stream: the bench step with the permutations
of the first update reused (nothing running beside K7) against the shipped step."""
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
def timed(n=40):
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
a = timed()
frozen = agent._take_perms().clone()
torch.cuda.synchronize()
agent._take_perms = lambda: frozen
b = timed()
print(f"shipped {a:.4f} ms per step; with the shuffle stream idle {b:.4f} ms")
