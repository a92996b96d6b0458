"""Soak: SOAK_UPDATES whole updates at the BASELINE size (K7 on the main stream, the shuffle pipeline for the next update beside it),
the permutations every update used compared with numpy's stream -- the accept relay under the contention it really runs in."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from aur_ppo_amd.ppo import ppo
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda:0")
hp = bench.hyper(args, 1)
hp["device"] = dev
hp["fused_mlp"] = True
hp["force_dp"] = False
torch.manual_seed(1)
agent = ppo(hp)
T, N, Dm, A = args.num_steps, agent.num_envs, args.obs_dim, args.act_dim
data = bench.synth_buffers(T, N, Dm, A, 1234)
for k in ("states", "actions", "values", "rewards", "terminals"):
    getattr(agent.buffer, k).copy_(data[k])
next_obs, next_done = data["next_obs"].to(dev), data["next_done"].to(dev)
with torch.no_grad():
    _, lp, _, _ = agent.policy.evaluate(agent.buffer.states.view(-1, Dm), agent.buffer.actions.view(-1, A))
    agent.buffer.log_probs.copy_(lp.view(T, N))
agent.seed_all(1)
rs = np.random.RandomState(1)
B, E = T * N, args.epochs
n_upd = int(os.environ.get("SOAK_UPDATES", "150"))
t0 = time.time()
for u in range(n_upd):
    returns, advantages = agent.advantages(next_obs, next_done)
    agent.update(returns, advantages)
    got = agent._last_perms.cpu().numpy()
    ref = np.arange(B)
    for e in range(E):
        rs.shuffle(ref)
        assert np.array_equal(got[e], ref), (u, e)
    if u % 25 == 0:
        print(f"update {u}: permutations ok ({time.time() - t0:.0f} s)", flush=True)
st = torch.zeros(1, device=dev)
agent.rng.status_into(st)
assert float(st) == 0.0 and torch.isfinite(agent.bucket.flat_param).all()
print(f"{n_upd} updates: every permutation numpy's, status clean, weights finite")
