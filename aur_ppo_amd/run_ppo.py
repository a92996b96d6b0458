"""Command line of the gym PPO trainer -- the flag set, short aliases and defaults of the
reference's src/run_ppo.py:14-41 (the script its README calls ``run.py``), including
the silent hyper-parameter override applied when ``--continuous`` is true (src/run_ppo.py:44-51).

Extra flags (not upstream; all optional): ``--keep_hparams`` opts out of that override so that
many-env continuous configurations can be launched from the CLI; ``--obs_dim/--act_dim`` size the
built-in ``Synthetic-v0`` environment.  Under ``torch.distributed.run`` the process group is
initialised from the environment and ``--num_envs`` is the global env count.
"""
from __future__ import annotations

import argparse


def strtobool(v) -> bool:
    s = str(v).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return True
    if s in ("n", "no", "f", "false", "off", "0"):
        return False
    raise ValueError(f"invalid truth value {v!r}")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    sb = lambda x: bool(strtobool(x))  # noqa: E731
    p.add_argument("-id", "--gym_id", type=str, default="CartPole-v1", help="Id of the environment that we will use")
    p.add_argument("-rb", "--robot", type=sb, default=False, nargs="?", const=False)
    p.add_argument("-s", "--seed", type=float, default=1.0, help="Seed for experiment")
    p.add_argument("-ns", "--num_steps", type=int, default=128, help="Number of steps that the environment should take")
    p.add_argument("-gae", "--gae", type=sb, default=True, nargs="?", const=True, help="Generalized Advantage Estimation flag")
    p.add_argument("-t", "--total_timesteps", type=int, default=500000, help="Total number of timesteps that we will take")
    p.add_argument("-al", "--anneal_lr", type=sb, default=True, nargs="?", const=True, help="How to anneal our learning rate")
    p.add_argument("-gl", "--gae_lambda", type=float, default=0.95, help="the lambda for the general advantage estimation")
    p.add_argument("-ue", "--num_update_epochs", type=int, default=4, help="The  number of update epochs for the policy")
    p.add_argument("-ne", "--num_envs", type=int, default=4, help="Number of environments to run in our vectorized setup")
    p.add_argument("-nm", "--num_minibatches", type=int, default=4, help="Number of minibatches")
    p.add_argument("-ec", "--entropy_coeff", type=float, default=0.01, help="Coefficient for entropy")
    p.add_argument("-vf", "--value_coeff", type=float, default=0.5, help="Coefficient for values")
    p.add_argument("-cf", "--clip_coeff", type=float, default=0.2, help="the surrogate clipping coefficient")
    p.add_argument("-cvl", "--clip_vloss", type=sb, default=True, nargs="?", const=True, help="Clip the value loss")
    p.add_argument("-mgn", "--max_grad_norm", type=float, default=0.5, help="the maximum norm for the gradient clipping")
    p.add_argument("-tkl", "--target_kl", type=float, default=None, help="The KL divergence that we will not exceed")
    # the next three are `type=bool` upstream: ANY non-empty string parses as True (src/run_ppo.py:31-32,40)
    p.add_argument("-na", "--norm_adv", type=bool, default=True, help="Normalize advantage estimates")
    p.add_argument("-p", "--capture_video", type=bool, default=False, help="Whether to capture the video or not")
    p.add_argument("-d", "--hidden_dim", type=int, default=64, help="Hidden dimension of the neural networks in the actor critic")
    p.add_argument("-c", "--continuous", type=sb, default=False, nargs="?", const=False)
    p.add_argument("-lr", "--learning_rate", type=float, default=2.5e-4, help="Learning rate for our agent")
    p.add_argument("-exp", "--exp_name", type=str, default="CartPole PPO", help="Experiment name")
    p.add_argument("-nl", "--num_layers", type=int, default=2, help="The number of layers in our actor and critic")
    p.add_argument("-do", "--dropout", type=float, default=0.0, help="Dropout in our actor and critic")
    p.add_argument("-g", "--gamma", type=float, default=0.99, help="Discount value for rewards")
    p.add_argument("-tr", "--track", type=bool, default=False, help="Track the performance of the environment")
    p.add_argument("-tri", "--trials", type=int, default=1, help="Number of trials to run")
    # --- not upstream
    p.add_argument("--keep_hparams", action="store_true", help="do not apply the --continuous hyper-parameter override")
    p.add_argument("--obs_dim", type=int, default=None, help="Synthetic-v0 observation width")
    p.add_argument("--act_dim", type=int, default=None, help="Synthetic-v0 action width / count")
    p.add_argument("--resume", type=str, default=None, help="continue from a checkpoint written by --checkpoint_path")
    p.add_argument("--checkpoint_path", type=str, default=None, help="where to write mid-run checkpoints")
    p.add_argument("--checkpoint_every", type=int, default=0, help="updates between mid-run checkpoints (0 = none)")
    return p


def params_from_args(args) -> dict:
    if args.continuous and not args.keep_hparams:      # src/run_ppo.py:44-51
        args.learning_rate = 3e-4
        args.num_envs = 1
        args.total_timesteps = 2000000
        args.num_steps = 2048
        args.num_minibatches = 32
        args.num_update_epochs = 10
        args.entropy_coeff = 0
    keys = ("gym_id", "seed", "num_steps", "gae", "total_timesteps", "anneal_lr", "gae_lambda", "num_update_epochs",
            "num_envs", "num_minibatches", "entropy_coeff", "value_coeff", "clip_coeff", "clip_vloss", "max_grad_norm",
            "target_kl", "norm_adv", "capture_video", "hidden_dim", "continuous", "learning_rate", "exp_name",
            "num_layers", "dropout", "gamma", "track")       # the params dict of src/run_ppo.py:53-81
    params = {k: getattr(args, k) for k in keys}
    for k in ("obs_dim", "act_dim", "resume", "checkpoint_path"):
        if getattr(args, k) is not None:
            params[k] = getattr(args, k)
    if args.checkpoint_every:
        params["checkpoint_every"] = args.checkpoint_every
    return params


def main(argv=None):
    args = build_parser().parse_args(argv)
    params = params_from_args(args)
    from . import dist as D
    D.init_from_env()
    from .ppo import ppo
    to_run = ppo(params)
    return to_run.train()


if __name__ == "__main__":
    main()
