"""Stress: K7w (every build of k_mlpw_step / k_mlpw3_step) on random net shapes -- hidden 8..128, 1..3 layers, state widths that do
and do not fill float4 rows or k-steps, tiny and ragged minibatches, both heads -- against the per-op autograd path.  Run it under
`timeout` (it also looks for hangs).    FUZZ_CASES=100 FUZZ_SEED=1 python tools/k7w_fuzz.py"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_mlp_wide import _setup
random.seed(int(os.environ.get("FUZZ_SEED", "1")))
n_cases = int(os.environ.get("FUZZ_CASES", "100"))
worst = 0.0
kernels = set()
for case in range(n_cases):
    cont = random.random() < 0.6
    hidden = random.choice([8, 24, 32, 48, 64, 65, 80, 96, 100, 112, 128, 128, 128])
    layers = random.choice([1, 2, 3])
    D = random.choice([1, 3, 4, 5, 8, 11, 16, 17, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128])
    A = random.randint(1, 16) if cont else random.randint(2, 16)
    T, N = random.choice([(4, 32), (8, 64), (16, 64), (32, 128)])
    B = T * N
    M = max(1, min(random.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 1000, B // 4, B // 2, B]), B))
    norm_adv = random.random() < 0.7 and M > 1
    vmode = random.choice([0, 1, 2])
    packed = (A if cont else 1) <= 12 and random.random() < 0.5
    H, pol, bucket, obs, act, rec = _setup(T, N, D, A, hidden, layers, seed=case, cont=cont)
    lay = H.mlp_layout(pol, bucket)
    if lay is None or not lay["wide"]:
        continue                       # (2 x 64 over <= 64 state floats is K7's)
    idx = torch.randperm(B, device="cuda")[:M].int()
    mb = H.gather(idx, [obs, act, rec])
    _, nlp, ent, nv = pol.evaluate(mb[0], mb[1])
    sc_ref = torch.empty(9, device="cuda")
    loss = H.ppo_loss_packed(nlp, nv, ent, mb[2], 0.2, 0.01, 0.5, norm_adv, vmode, sc_ref)
    bucket.zero_grad()
    loss.backward()
    g_ref = bucket.flat_grad[:lay["n_params"]].clone()
    g_out = torch.full_like(bucket.flat_grad, float("nan"))
    if packed:
        sc = H.mlp_ppo_step(obs, None, H.pack_records(rec, act.reshape(B, -1)), idx, bucket.flat_param, lay, g_out, 0.2, 0.01, 0.5, norm_adv, vmode)
    else:
        sc = H.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, g_out, 0.2, 0.01, 0.5, norm_adv, vmode)
    torch.cuda.synchronize()
    g = g_out[:lay["n_params"]]
    assert torch.isfinite(g).all(), (case, "non-finite gradient", hidden, layers, D, A, M)
    scale = float(g_ref.abs().max()) + 1e-12
    err = float((g - g_ref).abs().max()) / scale
    if err >= 1e-4:
        with torch.no_grad():           # a sample on a clip edge makes max() / clamp() pick either side: the case proves nothing
            ratio = (nlp - mb[2][:, 0]).double().exp()
            edge = float(torch.minimum((ratio - 1.2).abs(), (ratio - 0.8).abs()).min())
            vedge = float(((nv.reshape(-1) - mb[2][:, 3]).double().abs() - 0.2).abs().min()) if vmode == 1 else 1.0
        if min(edge, vedge) < 2e-6:
            print(f"case {case}: skipped, a sample sits {min(edge, vedge):.1e} from a clip edge", flush=True)
            continue
    assert err < 1e-4, (case, cont, hidden, layers, D, A, M, norm_adv, vmode, packed, err)
    worst = max(worst, err)
    assert torch.allclose(sc, sc_ref, rtol=5e-5, atol=5e-6, equal_nan=True), (case, sc, sc_ref)
    kernels.add((hidden == 128, hidden <= 64 and D <= 64, layers))
    if case % 20 == 0:
        print(f"case {case}: ok (cont={cont} {layers}x{hidden} D={D} A={A} B={B} M={M}), worst relative gradient error so far {worst:.2e}", flush=True)
print(f"k7w_fuzz: {n_cases} cases ok over {len(kernels)} kernel builds, worst relative gradient error {worst:.2e}")
