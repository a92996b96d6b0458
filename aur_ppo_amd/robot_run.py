"""Command line of the BulletArm trainer -- flags, aliases and defaults of src/robot_run.py:40-123
(bool flags parse as upstream: ``type=bool`` ones treat any non-empty string as True, the
``str2bool`` ones parse yes/no words)."""
from __future__ import annotations

import argparse


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("-id", "--gym_id", type=str, default="close_loop_block_reaching")
    p.add_argument("-s", "--seed", type=float, default=1.0)
    p.add_argument("-gae", "--gae", type=bool, default=True)
    p.add_argument("-ns", "--num_steps", type=int, default=1024)
    p.add_argument("-t", "--total_timesteps", type=int, default=50000)
    p.add_argument("-ue", "--num_update_epochs", type=int, default=10)
    p.add_argument("-pte", "--pretrain_episodes", type=int, default=100)
    p.add_argument("-pts", "--pretrain_steps", type=int, default=1000)
    p.add_argument("-ptb", "--pretrain_batch_size", type=int, default=8)
    p.add_argument("-cf", "--clip_coeff", type=float, default=0.2)
    p.add_argument("-lr", "--learning_rate", type=float, default=3e-4)
    p.add_argument("-ec", "--entropy_coeff", type=float, default=0.01)
    p.add_argument("-vf", "--value_coeff", type=float, default=0.5)
    p.add_argument("-nm", "--num_minibatches", type=int, default=4)
    p.add_argument("-expw", "--expert_weight", type=float, default=0.9)
    p.add_argument("-al", "--anneal_lr", type=bool, default=True)
    p.add_argument("-gl", "--gae_lambda", type=float, default=0.95)
    p.add_argument("-ne", "--num_envs", type=int, default=5)
    p.add_argument("-cvl", "--clip_vloss", type=bool, default=True)
    p.add_argument("-mgn", "--max_grad_norm", type=float, default=0.5)
    p.add_argument("-tkl", "--target_kl", type=float, default=None)
    p.add_argument("-na", "--norm_adv", type=bool, default=True)
    p.add_argument("-p", "--capture_video", type=bool, default=False)
    p.add_argument("-d", "--hidden_dim", type=int, default=64)
    p.add_argument("-c", "--continuous", type=str2bool, default=True, nargs="?", const=True)
    p.add_argument("-exp", "--exp_name", type=str, default="close_loop_block_pulling")
    p.add_argument("-nl", "--num_layers", type=int, default=2)
    p.add_argument("-do", "--dropout", type=float, default=0.0)
    p.add_argument("-g", "--gamma", type=float, default=0.99)
    p.add_argument("-tr", "--track", type=bool, default=False)
    p.add_argument("-tri", "--trials", type=int, default=1)
    p.add_argument("-eq", "--equivariant", type=bool, default=False)
    p.add_argument("-anexp", "--anneal_exp", type=bool, default=False)
    p.add_argument("-sfp", "--save_file_path", type=str, default=None)
    p.add_argument("-render", "--render", type=str2bool, default=False, nargs="?", const=False)
    p.add_argument("-dpr", "--do_pretraining", type=str2bool, default=True, nargs="?", const=True)
    # extras (not upstream; defaults reproduce upstream): image shape of the Synthetic-arm workloads, resume file
    p.add_argument("--obs_size", type=int, default=128)
    p.add_argument("--obs_channels", type=int, default=1)
    p.add_argument("--resume", type=str, default=None, help="continue from a checkpoint written by --checkpoint_path")
    p.add_argument("--checkpoint_path", type=str, default=None, help="where to write mid-run checkpoints")
    p.add_argument("--checkpoint_every", type=int, default=0, help="updates between mid-run checkpoints (0 = none)")
    return p


PARAM_KEYS = ("gym_id", "seed", "num_steps", "gae", "total_timesteps", "anneal_lr", "gae_lambda", "num_update_epochs",
              "num_envs", "num_minibatches", "entropy_coeff", "value_coeff", "clip_coeff", "clip_vloss", "max_grad_norm",
              "target_kl", "norm_adv", "capture_video", "hidden_dim", "continuous", "learning_rate", "exp_name",
              "num_layers", "dropout", "gamma", "track", "pretrain_episodes", "pretrain_steps", "pretrain_batch_size",
              "expert_weight", "equivariant", "anneal_exp", "save_file_path", "render", "do_pretraining",
              "obs_size", "obs_channels", "resume", "checkpoint_path", "checkpoint_every")


def params_from_args(args):
    return {k: getattr(args, k) for k in PARAM_KEYS}


def main(argv=None):
    args = build_parser().parse_args(argv)
    from . import dist as D
    D.init_from_env()
    from .robot_ppo import robot_ppo
    return robot_ppo(params_from_args(args)).train()


if __name__ == "__main__":
    main()
