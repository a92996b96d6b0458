"""Experiment: does replaying ONE hipGraphExec back to back cost a bubble per replay?  The update captured twice and the
two instances replayed alternately, against the shipped single instance."""
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
def shipped():
    ret, adv = agent.advantages(nobs, ndone)
    agent.update(ret, adv)
def timed(fn, n=40):
    for _ in range(6): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
a = timed(shipped)
graphs = []
for _ in range(2):
    perms = agent._take_perms(); agent._perm_static.copy_(perms)
    ret, adv = agent.advantages(nobs, ndone)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        agent._update_body(ret, adv, agent._perm_static, True)
    graphs.append(g)
k = [0]
def alternating():
    ret, adv = agent.advantages(nobs, ndone)
    perms = agent._take_perms(); agent._perm_static.copy_(perms)
    graphs[k[0] & 1].replay(); k[0] += 1
b = timed(alternating)
def single():
    ret, adv = agent.advantages(nobs, ndone)
    perms = agent._take_perms(); agent._perm_static.copy_(perms)
    graphs[0].replay()
c = timed(single)
print(f"shipped {a:.4f} ms; two graph instances alternating {b:.4f} ms; one of them alone {c:.4f} ms")
