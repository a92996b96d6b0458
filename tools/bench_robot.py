"""BASELINE config 3's shape with the non-equivariant CNN actor-critic (the equivariant one needs e2cnn): robot_ppo's GAE +
update (src/robot_ppo.py:329-408) on synthetic image rollouts, N envs x T steps of (1,128,128) observations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd.robot_ppo import robot_ppo
from aur_ppo_amd.robot_run import build_parser, params_from_args
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
E = int(sys.argv[3]) if len(sys.argv) > 3 else 4
p = params_from_args(build_parser().parse_args([]))
p.update(gym_id="Synthetic-arm", num_envs=N, num_steps=T, total_timesteps=N * T * 4, num_update_epochs=E, num_minibatches=4,
         do_pretraining=False, log=False)
torch.manual_seed(1)
a = robot_ppo(p)
g = torch.Generator(device="cuda").manual_seed(3)
b = a.buffer
b.states.copy_((torch.rand(T, N, device="cuda", generator=g) < 0.5).float())
b.observations.copy_(torch.rand(T, N, 1, 128, 128, device="cuda", generator=g))
b.actions.copy_(0.3 * torch.randn(T, N, 5, device="cuda", generator=g))
b.rewards.copy_((torch.rand(T, N, device="cuda", generator=g) < 0.3).float())
b.terminals.copy_((torch.rand(T, N, device="cuda", generator=g) < 0.02).float())
with torch.no_grad():
    for t in range(T):
        _, _, lp, _, v = a.policy.evaluate(b.states[t], b.observations[t], b.actions[t])
        b.log_probs[t].copy_(lp); b.values[t].copy_(v.flatten())
ns, no, nd = b.states[0].clone(), b.observations[0].clone(), torch.zeros(N, device="cuda")
a.seed_all(1)
def step():
    ret, adv = a.advantages(ns, no, nd, b, T)
    a.update(b.flatten(ret, adv), E, a.batch_size, a.minibatch_size, [])
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
K = 3
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"robot_ppo CNN policy, N={N} T={T} E={E}, 4 minibatches of {a.minibatch_size}: {dt * 1e3:.1f} ms per GAE+update -> {N * T / dt / 1e6:.3f} M env-steps/s")
