"""Stress: K7 (both builds) and the chained minibatch on random shapes -- tiny and ragged minibatches, every head width,
odd observation widths -- checked against the per-op autograd path.  Looks for hangs (run it under `timeout`) and
for shape-dependent errors the parametrised tests do not reach."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_mlp_fused import _setup
random.seed(int(os.environ.get("FUZZ_SEED", "1")))
n_cases = int(os.environ.get("FUZZ_CASES", "120"))
worst = 0.0
only = int(os.environ.get("FUZZ_ONLY", "-1"))   # re-run one case of a seed (the draws of the others are still consumed)
for case in range(n_cases):
    cont = random.random() < 0.6
    D = random.choice([1, 2, 3, 4, 5, 6, 8, 10, 11, 16, 17, 30, 32, 33, 48, 62, 63, 64])
    A = random.randint(1, 16) if cont else random.randint(2, 16)
    T, N = random.choice([(4, 32), (8, 64), (16, 64), (32, 128)])
    B = T * N
    M = random.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 1000, B // 4, B // 2, B])
    M = max(1, min(M, B))
    os.environ["AURPPO_K7_VARIANT"] = random.choice(["2", "3", "3"])
    norm_adv = random.random() < 0.7 and M > 1
    vmode = random.choice([0, 1, 2])
    packed = (A if cont else 1) <= 12 and random.random() < 0.5
    if only >= 0 and case != only:
        continue
    H, pol, bucket, obs, act, rec = _setup(T, N, D, A, seed=case, cont=cont)
    lay = H.mlp_layout(pol, bucket)
    idx = torch.randperm(B, device="cuda")[:M].int()
    mb = H.gather(idx, [obs, act, rec])
    _, nlp, ent, nv = pol.evaluate(mb[0], mb[1])
    sc_ref = torch.empty(9, device="cuda")
    loss = H.ppo_loss_packed(nlp, nv, ent, mb[2], 0.2, 0.01, 0.5, norm_adv, vmode, sc_ref)
    bucket.zero_grad()
    loss.backward()
    g_ref = bucket.flat_grad[:lay["n_params"]].clone()
    g_out = torch.full_like(bucket.flat_grad, float("nan"))
    aw = A if cont else 1
    if packed:      # packed 64-byte records
        sc = H.mlp_ppo_step(obs, None, H.pack_records(rec, act.reshape(B, -1)), idx, bucket.flat_param, lay, g_out, 0.2, 0.01, 0.5,
                            norm_adv, vmode)
    else:
        sc = H.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, g_out, 0.2, 0.01, 0.5, norm_adv, vmode)
    torch.cuda.synchronize()
    g = g_out[:lay["n_params"]]
    assert torch.isfinite(g).all(), (case, "non-finite gradient")
    scale = float(g_ref.abs().max()) + 1e-12
    err = float((g - g_ref).abs().max()) / scale
    if err >= 1e-4:
        # a sample whose ratio sits on a clip edge (or whose value step sits on +-clip) makes max()/clamp() pick either
        # side depending on the last bit of the log-prob: both derivatives are valid, the case proves nothing
        with torch.no_grad():
            ratio = (nlp - mb[2][:, 0]).double().exp()
            edge = float(torch.minimum((ratio - 1.2).abs(), (ratio - 0.8).abs()).min())
            vedge = float(((nv.reshape(-1) - mb[2][:, 3]).double().abs() - 0.2).abs().min()) if vmode == 1 else 1.0
        if min(edge, vedge) < 2e-6:
            print(f"case {case}: skipped, a sample sits {min(edge, vedge):.1e} from a clip edge", flush=True)
            continue
        names = (["actor_logstd"] if cont else []) + [f"{n}.{k}" for n in ("actor", "critic") for k in ("w1", "b1", "w2", "b2", "w3", "b3")]
        off = 0
        for p_, nm in zip(bucket.params, names):
            k = p_.numel()
            print(f"   {nm:14s} max |err| {float((g[off:off + k] - g_ref[off:off + k]).abs().max()):.3e}  max |ref| {float(g_ref[off:off + k].abs().max()):.3e}")
            off += k
        print("   variant", os.environ["AURPPO_K7_VARIANT"], "packed", packed, "T,N", T, N)
    assert err < 1e-4, (case, cont, D, A, M, norm_adv, vmode, err)
    worst = max(worst, err)
    assert torch.allclose(sc, sc_ref, rtol=5e-5, atol=5e-6, equal_nan=True), (case, sc, sc_ref)
    if case % 20 == 0:
        print(f"case {case}: ok (cont={cont} D={D} A={A} B={B} M={M}), worst relative gradient error so far {worst:.2e}", flush=True)
print(f"{n_cases} cases ok, worst relative gradient error {worst:.2e}")
