"""Stress: the shuffle pipeline (default accept kernel, random relay widths) against numpy over random sizes and seeds, several
shuffles per handle so the stream carries over -- looks for the rare paths of k_fy_accept3 (a true index outside the window, a list
that overflows, octave edges on chunk boundaries, the tail octaves) and for hangs (run it under `timeout`)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AURPPO_TEST_KNOBS"] = "1"
import numpy as np, torch
from aur_ppo_amd import hip_ops as H
random.seed(int(os.environ.get("FUZZ_SEED", "1")))
n_cases = int(os.environ.get("FUZZ_CASES", "150"))
for case in range(n_cases):
    kind = random.random()
    if kind < 0.3:
        n = random.randint(2, 3000)
    elif kind < 0.6:
        n = random.choice([2 ** k for k in range(4, 21)]) + random.randint(-3, 3)
    else:
        n = random.randint(3000, 1200000)
    n = max(n, 2)
    seed = random.randint(0, 2 ** 32 - 1)
    os.environ["AURPPO_K2_ACCEPT3_WGS"] = str(random.choice([1, 2, 3, 5, 6, 8]))
    reps = random.randint(1, 3)
    rng = H.MT19937(seed, n)
    got = rng.shuffle_epochs(n, reps).cpu().numpy()
    rs = np.random.RandomState(seed)
    idx = np.arange(n)
    for e in range(reps):
        rs.shuffle(idx)
        assert np.array_equal(got[e], idx), (case, n, seed, e, os.environ["AURPPO_K2_ACCEPT3_WGS"])
    key, pos = rng.get_state()
    st = rs.get_state()
    assert np.array_equal(key, st[1]) and pos == st[2], (case, n, seed)
    status = torch.zeros(1, device="cuda")
    rng.status_into(status)
    assert float(status) == 0.0
    if case % 25 == 0:
        print(f"case {case}: ok (n={n}, {reps} shuffles, {os.environ['AURPPO_K2_ACCEPT3_WGS']} workgroups)", flush=True)
print(f"{n_cases} cases ok")
