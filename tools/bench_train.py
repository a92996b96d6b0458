"""End-to-end train() throughput (rollout + GAE + update), the reference's charts/SPS metric, synthetic env."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd.ppo import ppo
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fused = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
U = 12
hp = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=128, gae=True, total_timesteps=128 * N * U, anneal_lr=True, gae_lambda=0.95,
          num_update_epochs=4, num_envs=N, num_minibatches=4, entropy_coeff=0.0, value_coeff=0.5, clip_coeff=0.2, clip_vloss=True,
          max_grad_norm=0.5, target_kl=None, norm_adv=True, capture_video=False, hidden_dim=64, continuous=True,
          learning_rate=3e-4, exp_name="bench", num_layers=2, dropout=0.0, gamma=0.99, track=False, log=False, save=False,
          obs_dim=64, act_dim=6, fused_mlp=fused)
a = ppo(hp)
t0 = time.perf_counter()
a.train()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"N={N} fused={fused}: {U} updates in {dt:.3f} s -> SPS {128 * N * U / dt / 1e6:.2f} M env-steps/s ({dt / U * 1e3:.1f} ms per update incl. rollout)")
