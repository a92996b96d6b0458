"""How many streams the process can keep busy before K7's stream shares a hardware queue: the bench step with 0..3 extra
streams that each run a trickle of small kernels.  Run with and without GPU_MAX_HW_QUEUES=8 in the environment."""
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
extra = [torch.cuda.Stream() for _ in range(3)]
bufs = [torch.zeros(1 << 20, device="cuda") for _ in range(3)]
def step(n_extra):
    ret, adv = agent.advantages(nobs, ndone)
    agent.update(ret, adv)
    for k in range(n_extra):
        with torch.cuda.stream(extra[k]):
            for _ in range(20):
                bufs[k].add_(1.0)
def timed(n_extra, n=30):
    for _ in range(5): step(n_extra)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step(n_extra)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
for k in (0, 1, 2, 3, 0):
    print(f"  {k} extra busy streams: {timed(k):.3f} ms per step")
