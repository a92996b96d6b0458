"""CPU: the CLI reproduces the reference's flags, aliases, defaults and continuous override."""
from aur_ppo_amd.run_ppo import build_parser, params_from_args

REF_DEFAULTS = dict(gym_id="CartPole-v1", seed=1.0, num_steps=128, gae=True, total_timesteps=500000, anneal_lr=True,
                    gae_lambda=0.95, num_update_epochs=4, num_envs=4, num_minibatches=4, entropy_coeff=0.01,
                    value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True,
                    capture_video=False, hidden_dim=64, continuous=False, learning_rate=2.5e-4,
                    exp_name="CartPole PPO", num_layers=2, dropout=0.0, gamma=0.99, track=False)


def test_defaults_and_param_keys_match_reference():
    p = params_from_args(build_parser().parse_args([]))
    assert p == REF_DEFAULTS                       # src/run_ppo.py:14-41,53-81


def test_short_aliases():
    argv = ("-id X -s 3 -ns 64 -gae False -t 1000 -al False -gl 0.9 -ue 2 -ne 8 -nm 2 -ec 0.0 -vf 0.25 -cf 0.1 "
            "-cvl False -mgn 1.0 -tkl 0.02 -d 32 -lr 0.001 -exp e -nl 3 -do 0.1 -g 0.9 -tri 2 -rb False").split()
    a = build_parser().parse_args(argv)
    p = params_from_args(a)
    assert (p["gym_id"], p["seed"], p["num_steps"], p["gae"], p["total_timesteps"], p["anneal_lr"]) == ("X", 3.0, 64, False, 1000, False)
    assert (p["gae_lambda"], p["num_update_epochs"], p["num_envs"], p["num_minibatches"]) == (0.9, 2, 8, 2)
    assert (p["clip_vloss"], p["max_grad_norm"], p["target_kl"], p["hidden_dim"], p["num_layers"]) == (False, 1.0, 0.02, 32, 3)
    assert a.trials == 2 and a.robot is False


def test_continuous_override_and_opt_out():
    p = params_from_args(build_parser().parse_args(["-c", "True", "-ne", "1024"]))
    assert (p["learning_rate"], p["num_envs"], p["total_timesteps"], p["num_steps"], p["num_minibatches"],
            p["num_update_epochs"], p["entropy_coeff"]) == (3e-4, 1, 2000000, 2048, 32, 10, 0)   # src/run_ppo.py:44-51
    p = params_from_args(build_parser().parse_args(["-c", "True", "-ne", "1024", "--keep_hparams", "--obs_dim", "64"]))
    assert p["num_envs"] == 1024 and p["obs_dim"] == 64 and p["num_steps"] == 128


def test_type_bool_flags_parse_like_upstream():
    # `type=bool`: any non-empty string is True, even "False" (upstream quirk, kept)
    p = params_from_args(build_parser().parse_args(["-na", "False", "-tr", "0"]))
    assert p["norm_adv"] is True and p["track"] is True
