"""robot_ppo's GAE + update (src/robot_ppo.py:329-408) on synthetic image rollouts with the plain-CNN actor-critic (the
equivariant one needs e2cnn): BASELINE config 3's shape (N=256, T=128, (1,128,128)) and config 5's per-GPU shard
(N=2048/8=256, T=64, (3,84,84), build-defined encoder).  Prints one JSON line per run with the conv-FLOP bound.

    python tools/bench_robot.py --config 3 [--channels-last] [--miopen-find] [--updates 2]

``run()`` is what ``bench.py --workload robot3|robot5`` calls.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
T0 = time.time()
def note(msg):
    print(f"[bench_robot {time.time() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)
import threading
_hb = []
def _heartbeat():      # MIOpen compiles / searches its solvers on first use of a shape: minutes of silence on a fresh box
    while True:
        time.sleep(60)
        note("still running")
def start_heartbeat():
    if not _hb:
        _hb.append(threading.Thread(target=_heartbeat, daemon=True))
        _hb[0].start()

# forward conv FLOPs of ONE encoder per sample (2 * out_pixels * out_ch * in_ch * 9), two encoders (actor, critic), fwd + bwd ~ 3x
def conv_flops(size, cin):
    f, c_in, s = 0, cin, size
    for c_out in (16, 32, 64, 128):
        f += 2 * s * s * c_out * c_in * 9; c_in = c_out; s //= 2
    if size == 128:
        f += 2 * s * s * 256 * 128 * 9; s -= 2; f += 2 * s * s * 256 * 256 * 9; s //= 2; s -= 2; f += 2 * s * s * 128 * 256 * 9
    else:
        s -= 2; f += 2 * s * s * 256 * 128 * 9; s -= 2; f += 2 * s * s * 128 * 256 * 9
    return f
def equiv_flops(size, cin, n):      # expanded C4 filter banks: 4 channels per regular field
    if size == 128:
        spec = [(cin, n // 8 * 4, 1, 2), (n // 8 * 4, n // 4 * 4, 1, 2), (n // 4 * 4, n // 2 * 4, 1, 2), (n // 2 * 4, n * 4, 1, 2),
                (n * 4, 2 * n * 4, 1, 0), (2 * n * 4, n * 4, 0, 2), (n * 4, n * 4, 0, 0)]
    else:
        spec = [(cin, n // 8 * 4, 1, 2), (n // 8 * 4, n // 4 * 4, 1, 2), (n // 4 * 4, n // 2 * 4, 1, 3), (n // 2 * 4, n * 4, 0, 0),
                (n * 4, n * 4, 0, 0), (n * 4, n * 4, 0, 0)]
    f, s = 0, size
    for ci, co, pad, pool in spec:
        s = s if pad else s - 2
        f += 2 * s * s * co * ci * 9
        s = s // pool if pool else s
    return f

CONV_MARKS = ("miopen", "MIOpen", "winograd", "Winograd", "igemm", "Igemm", "conv", "Conv", "gemm", "Cijk", "naive", "transpose", "Transpose", "batched_transpose", "SubTensor", "Op2d", "Op3d", "Op5d")


def run(config=3, envs=0, steps=0, epochs=4, minibatches=4, updates=2, warmup=1, channels_last=False, equivariant=False,
        equiv_hidden=128, kernel_table=False, miopen_find=False, device=None):
    """Time ``updates`` steady-state GAE + update passes of robot_ppo at the config's shape; returns a dict (see the keys at
    the end).  ``kernel_table``: one more update under torch.profiler -- per-kernel GPU time, from which the convolution
    kernels' share and their aggregate FLOP rate are formed."""
    from aur_ppo_amd.robot_ppo import robot_ppo
    from aur_ppo_amd.robot_run import build_parser, params_from_args
    start_heartbeat()
    C, S = (1, 128) if config == 3 else (3, 84)
    N = envs or 256
    T = steps or (128 if config == 3 else 64)
    E = epochs
    torch.backends.cudnn.benchmark = bool(miopen_find)
    p = params_from_args(build_parser().parse_args([]))
    p.update(gym_id="Synthetic-arm", num_envs=N, num_steps=T, total_timesteps=N * T * 4, num_update_epochs=E, num_minibatches=minibatches,
             do_pretraining=False, log=False, obs_size=S, obs_channels=C, channels_last=channels_last,
             equivariant=equivariant, equiv_hidden=equiv_hidden)
    if device is not None:
        p["device"] = device
    torch.manual_seed(1)
    a = robot_ppo(p)
    note("trainer built")
    g = torch.Generator(device="cuda").manual_seed(3)
    b = a.buffer
    b.states.copy_((torch.rand(T, N, device="cuda", generator=g) < 0.5).float())
    for t in range(T):
        b.observations[t].copy_(torch.rand(N, C, S, S, device="cuda", generator=g))
    b.actions.copy_(0.3 * torch.randn(T, N, 5, device="cuda", generator=g))
    b.rewards.copy_((torch.rand(T, N, device="cuda", generator=g) < 0.3).float())
    b.terminals.copy_((torch.rand(T, N, device="cuda", generator=g) < 0.02).float())
    with torch.no_grad():
        for t in range(T):
            _, _, lp, _, v = a.policy.evaluate(b.states[t], b.observations[t], b.actions[t])
            b.log_probs[t].copy_(lp); b.values[t].copy_(v.flatten())
    note("rollout values filled (T forward passes at N rows: MIOpen picks/compiles its solvers on first use of a shape)")
    ns, no, nd = b.states[0].clone(), b.observations[0].clone(), torch.zeros(N, device="cuda")
    a.seed_all(1)
    def step():
        ret, adv = a.advantages(ns, no, nd, b, T)
        a.update(b.flatten(ret, adv), E, a.batch_size, a.minibatch_size, [])
    for _ in range(max(1, warmup)):
        step()
    torch.cuda.synchronize()
    note("warm-up update(s) done (solvers for the minibatch shapes chosen)")
    t0 = time.perf_counter()
    for _ in range(updates): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / updates
    note(f"timed: {dt * 1e3:.1f} ms per update")
    flops = 3 * 2 * (equiv_flops(S, C + 1, equiv_hidden) if equivariant else conv_flops(S, C + 1)) * N * T * E
    out = {"policy": "C4-equivariant (build-defined)" if equivariant else "plain CNN",
           "workload": f"robot_ppo, config {config}: N={N} T={T} obs=({C},{S},{S}) E={E}, {minibatches} minibatches of {a.minibatch_size}",
           "N": N, "T": T, "channels_last": channels_last, "miopen_find": miopen_find, "updates_timed": updates,
           "ms_per_update": round(dt * 1e3, 1), "env_steps_per_s": round(N * T / dt, 1), "conv_tflop_per_update": round(flops / 1e12, 1),
           "conv_flops_per_update": flops, "conv_bound_ms_at_157_tflops": round(flops / 157.3e12 * 1e3, 1),
           "frac_of_fp32_mfma_peak": round(flops / dt / 157.3e12, 3)}
    if kernel_table:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step(); torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
        tot = sum(e.device_time_total for e in rows)
        note(f"kernel table of one steady-state update ({tot / 1e3:.1f} ms of GPU time):")
        for e in rows[:30]:
            print(f"  {e.device_time_total / 1e3:9.1f} ms {100 * e.device_time_total / tot:5.1f}%  x{e.count:<6d} {e.key[:120]}", file=sys.stderr)
        ours = ("k_brp", "k_first_block", "k_gae", "k_gather", "k_loss", "k_adv", "k_clip", "k_sqnorm", "k_fy", "k_mt", "k_weighted", "k_pack")
        # every convolution kernel of the two encoders (K11 / K12 are named k_conv3x3*; the rest is the library's, its layout transposes included)
        conv = [e for e in rows if not e.key.startswith(ours) and any(m in e.key for m in CONV_MARKS)]
        lib_us = sum(e.device_time_total for e in conv)
        hand_us = sum(e.device_time_total for e in conv if "k_conv3x3" in e.key or "k_fold_slices" in e.key)
        out["kernel_table"] = {"gpu_ms_per_update": round(tot / 1e3, 1),
                               "library_conv_ms": round(lib_us / 1e3, 1), "library_conv_share": round(lib_us / tot, 3),
                               "library_conv_tflops": round(flops / (lib_us * 1e-6) / 1e12, 1) if lib_us else None,
                               "hand_written_conv_ms": round(hand_us / 1e3, 1), "miopen_conv_ms": round((lib_us - hand_us) / 1e3, 1),
                               "top": [{"kernel": e.key[:100], "ms": round(e.device_time_total / 1e3, 2), "share": round(e.device_time_total / tot, 4),
                                        "launches": e.count, "avg_us": round(e.device_time_total / max(e.count, 1), 1)} for e in rows[:12]],
                               "how": "torch.profiler (CUDA activity) over one more steady-state update in this process"}
    a_ref = a
    out["_agent"] = a_ref
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3, choices=(3, 5))
    ap.add_argument("--envs", type=int, default=0)
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--minibatches", type=int, default=4)
    ap.add_argument("--updates", type=int, default=2)
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--equivariant", action="store_true", help="the build-defined C4-equivariant actor / critic (aur_ppo_amd/equiv.py)")
    ap.add_argument("--equiv-hidden", type=int, default=128)
    ap.add_argument("--kernel-table", action="store_true", help="after the timed updates, one more under torch.profiler: top kernels by GPU time")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark: let MIOpen time its solvers per shape "
                    "(on a fresh box the search for the 8192-row minibatch shapes alone ran past 7 minutes: not used)")
    args = ap.parse_args()
    res = run(args.config, args.envs, args.steps, args.epochs, args.minibatches, args.updates, 1, args.channels_last, args.equivariant,
              args.equiv_hidden, args.kernel_table, args.miopen_find)
    res.pop("_agent", None)
    res.pop("conv_flops_per_update", None)
    print(json.dumps(res))
