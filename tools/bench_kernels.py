"""Micro-benchmarks of the custom kernels at BASELINE sizes (HIP-event timing on the current
stream).  'cold' rotates over several buffer sets larger than the 256 MiB Infinity Cache."""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd import hip_ops as H

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="gather,gae,loss,clip,shuffle")
ap.add_argument("--N", type=int, default=4096)
ap.add_argument("--T", type=int, default=128)
ap.add_argument("--D", type=int, default=64)
ap.add_argument("--A", type=int, default=6)
ap.add_argument("--iters", type=int, default=40)
args = ap.parse_args()
dev = torch.device("cuda")
T, N, D, A = args.T, args.N, args.D, args.A
B, M = T * N, T * N // 4


def timeit(fn, iters=args.iters, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    evs = []
    for i in range(iters):
        b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b.record(); fn(i); e.record()
        evs.append((b, e))
    torch.cuda.synchronize()
    ts = sorted(b.elapsed_time(e) * 1e3 for b, e in evs)
    return ts[len(ts) // 2], ts[0]

out = {}
what = args.what.split(",")
if "gather" in what:
    nset = 4
    sets = [[torch.randn(B, D, device=dev), torch.randn(B, A, device=dev)] + [torch.randn(B, device=dev) for _ in range(4)]
            for _ in range(nset)]
    outs = [[torch.empty((M,) + tuple(s.shape[1:]), device=dev) for s in sets[0]] for _ in range(2)]
    rng = H.MT19937(1, B)
    perms = rng.shuffle_epochs(B, 4)
    nbytes = M * (8 * D + 8 * A + 36)
    for label, rot in (("hot", 1), ("cold", nset)):
        med, mn = timeit(lambda i: H.gather(perms[i % 4][(i % 4) * M:(i % 4 + 1) * M] if False else perms[i % 4][:M], sets[i % rot], outs[i % 2]))
        out[f"gather_{label}"] = dict(us=med, min_us=mn, GBs=nbytes / med / 1e3, frac=nbytes / med / 1e3 / 8000)
    # packed layout: obs + actions + one (B,4) record instead of four scalar streams
    psets = [[st[0], st[1], torch.randn(B, 4, device=dev)] for st in sets]
    pouts = [[torch.empty((M,) + tuple(s.shape[1:]), device=dev) for s in psets[0]] for _ in range(2)]
    for label, rot in (("hot", 1), ("cold", nset)):
        med, mn = timeit(lambda i: H.gather(perms[i % 4][:M], psets[i % rot], pouts[i % 2]))
        out[f"gather_packed_{label}"] = dict(us=med, min_us=mn, GBs=nbytes / med / 1e3, frac=nbytes / med / 1e3 / 8000)
    # obs-only, to separate the wide stream from the narrow ones
    med, mn = timeit(lambda i: H.gather(perms[i % 4][:M], sets[i % nset][:1], outs[i % 2][:1]))
    out["gather_obs_only_cold"] = dict(us=med, GBs=M * (8 * D + 4) / med / 1e3)
    med, mn = timeit(lambda i: H.gather(perms[i % 4][:M], sets[i % nset][2:], outs[i % 2][2:]))
    out["gather_scalars_only_cold"] = dict(us=med)
    # reference points: torch index_select of the obs rows, and a plain copy of the same bytes
    idx64 = perms[0][:M].long()
    med, _ = timeit(lambda i: torch.index_select(sets[i % nset][0], 0, idx64, out=outs[i % 2][0]))
    out["torch_index_select_obs_cold"] = dict(us=med, GBs=M * (8 * D + 8) / med / 1e3)
    med, _ = timeit(lambda i: outs[i % 2][0].copy_(sets[i % nset][0][:M]))
    out["copy_same_bytes_obs"] = dict(us=med, GBs=M * 8 * D / med / 1e3)
if "gae" in what:
    r, v = torch.randn(T, N, device=dev), torch.randn(T, N, device=dev)
    d = (torch.rand(T, N, device=dev) < 0.02).float()
    nv, nd = torch.randn(N, device=dev), torch.zeros(N, device=dev)
    o = (torch.empty_like(r), torch.empty_like(r))
    med, mn = timeit(lambda i: H.gae(r, v, d, nv, nd, 0.99, 0.95, out=o))
    out["gae"] = dict(us=med, min_us=mn, GBs=B * 20 / med / 1e3, frac=B * 20 / med / 1e3 / 8000)
if "loss" in what:
    a = [torch.randn(M, device=dev) for _ in range(7)]
    med, mn = timeit(lambda i: H.loss_fwd_bwd(*a, 0.2, 0.0, 0.5, True, 1))
    out["loss_fwd_bwd"] = dict(us=med, min_us=mn, GBs=M * 40 / med / 1e3)
if "clip" in what:
    g = torch.randn(17104, device=dev)
    med, mn = timeit(lambda i: H.grad_norm_clip_(g, 0.5))
    out["clip_17k"] = dict(us=med, min_us=mn)
if "shuffle" in what:
    for n in (65536, 131072, 524288):
        rng = H.MT19937(1, n)
        o = torch.empty((1, n), dtype=torch.int32, device=dev)
        med, mn = timeit(lambda i: rng.shuffle_epochs(n, 1, out=o), iters=8, warm=2)
        out[f"shuffle_{n}"] = dict(us=med, min_us=mn)
for k, v in out.items():
    print(k, json.dumps({kk: round(vv, 2) if isinstance(vv, float) else vv for kk, vv in v.items()}))
