"""End-to-end train() throughput (rollout + GAE + update), the reference's charts/SPS metric, synthetic env."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd.ppo import ppo
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fused = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
U = int(sys.argv[3]) if len(sys.argv) > 3 else 60
hp = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=128, gae=True, total_timesteps=128 * N * U, anneal_lr=True, gae_lambda=0.95,
          num_update_epochs=4, num_envs=N, num_minibatches=4, entropy_coeff=0.0, value_coeff=0.5, clip_coeff=0.2, clip_vloss=True,
          max_grad_norm=0.5, target_kl=None, norm_adv=True, capture_video=False, hidden_dim=64, continuous=True,
          learning_rate=3e-4, exp_name="bench", num_layers=2, dropout=0.0, gamma=0.99, track=False, log=False, save=False,
          obs_dim=64, act_dim=6, fused_mlp=fused)
a = ppo(hp)
# time the steady state: train() logs wall-clock per update through _log_update; here the first updates (eager warm-up and
# the two graph captures) are excluded by timing the last U - 10 updates
marks = []
orig = a._log_update
def _mark(*args, **kw):
    torch.cuda.synchronize()
    marks.append(time.perf_counter())
    return orig(*args, **kw)
a._log_update = _mark
a.train()
torch.cuda.synchronize()
skip = min(10, U - 2)
dt = marks[-1] - marks[skip - 1]
n = len(marks) - skip
print(f"N={N} fused={fused}: {n} steady-state updates in {dt:.3f} s -> SPS {128 * N * n / dt / 1e6:.2f} M env-steps/s "
      f"({dt / n * 1e3:.2f} ms per update incl. rollout); first {skip} updates (warm-up, graph captures): {marks[skip - 1] - marks[0]:.3f} s")
