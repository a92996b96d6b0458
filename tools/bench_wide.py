"""K7w / K8w against the per-op path (K3 gather -> torch evaluate -> K5 loss -> autograd) at the headline minibatch size,
for the MLP shapes src/run_ppo.py's -d / -nl can ask for.  Prints one JSON object; run on the GPU box:
    python tools/bench_wide.py > gpurun_out/wide_bench.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aur_ppo_amd import hip_ops as H                       # noqa: E402
from aur_ppo_amd.actor_critic import actor_critic         # noqa: E402
from aur_ppo_amd.flat import FlatBucket                   # noqa: E402

PEAK = 157.3e12


def timed(fn, reps):
    """Median of three batches of ``reps`` calls, in us per call."""
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(out)[1]


def main():
    T, N, A, M = 128, 4096, 6, 131072
    rows = []
    for hidden, layers, D in [(64, 2, 64), (64, 1, 64), (64, 3, 64), (128, 2, 64), (128, 3, 64), (128, 3, 128), (32, 2, 64), (96, 2, 64)]:
        torch.manual_seed(0)
        pol = actor_critic(D, (A,), hidden, layers, 0.0, True).cuda()
        bucket = FlatBucket(pol.parameters())
        lay = H.mlp_layout(pol, bucket)
        B = T * N
        g = torch.Generator(device="cuda").manual_seed(1)
        obs = torch.randn(B, D, device="cuda", generator=g)
        act = torch.randn(B, A, device="cuda", generator=g)
        rec = torch.randn(B, 4, device="cuda", generator=g)
        rec64 = H.pack_records(rec, act)
        idx = torch.randperm(B, device="cuda")[:M].int()
        gout = torch.empty_like(bucket.flat_grad)
        sc = torch.empty(9, device="cuda")
        ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]

        def fused():
            H.mlp_ppo_step(obs, None, rec64, idx, bucket.flat_param, lay, gout, 0.2, 0.0, 0.5, True, 1, sc)

        def per_op():
            mb = H.gather(idx, [obs, act, rec])
            _, nlp, ent, nv = pol.evaluate(mb[0], mb[1])
            loss = H.ppo_loss_packed(nlp, nv, ent, mb[2], 0.2, 0.0, 0.5, True, 1, sc)
            bucket.zero_grad()
            loss.backward()

        # main kernel alone (events inside the library call)
        ks = []
        for _ in range(5):
            H.mlp_ppo_step(obs, None, rec64, idx, bucket.flat_param, lay, gout, 0.2, 0.0, 0.5, True, 1, sc, events=ev)
            torch.cuda.synchronize()
            ks.append(ev[0].elapsed_time(ev[1]) * 1e3)
        t_f = timed(fused, 20)
        t_p = timed(per_op, 5)
        flops = H.mlp_step_flops(lay, M)
        o1 = torch.randn(N, D, device="cuda", generator=g)
        nz = torch.randn(N, A, device="cuda", generator=g)
        t_act = timed(lambda: H.mlp_act(o1, nz, bucket.flat_param, lay), 50)
        with torch.no_grad():
            t_act_t = timed(lambda: pol.evaluate(o1), 20)
        rows.append(dict(hidden=hidden, layers=layers, D=D, kernel=f"K7 (k_mlp_step{H.k7_variant()})" if not lay["wide"] else f"K7w ({H.k7w_kernel_name(hidden, D, layers)})",
                         n_params=lay["n_params"], gflop=round(flops / 1e9, 2), main_kernel_us=round(sorted(ks)[2], 1),
                         step_us=round(t_f, 1), per_op_us=round(t_p, 1), speedup=round(t_p / t_f, 1),
                         frac_of_fp32_mfma_peak=round(flops / (sorted(ks)[2] * 1e-6) / PEAK, 3),
                         act_us=round(t_act, 1), act_torch_us=round(t_act_t, 1)))
        print(rows[-1], file=sys.stderr, flush=True)
    print(json.dumps(dict(M=M, A=A, rows=rows, note="step_us = stats/prep + main kernel + slab reduce; per_op = gather + torch "
                          "evaluate + K5 + autograd; act = one rollout step at N = 4096"), indent=1))


if __name__ == "__main__":
    main()
