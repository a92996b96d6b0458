"""Shared helpers for the parity tests (tests may import oracle/; the product may not)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def gae_big_inputs(ci, T, N):
    """Same generator as oracle/gen_golden.py::gae_big_inputs (numpy legacy stream, version-stable)."""
    rs = np.random.RandomState(1000 + ci)
    r = rs.standard_normal((T, N)).astype(np.float32)
    v = rs.standard_normal((T, N)).astype(np.float32)
    d = (rs.random_sample((T, N)) < 0.02).astype(np.float32)
    nv = rs.standard_normal(N).astype(np.float32)
    nd = (rs.random_sample(N) < 0.02).astype(np.float32)
    return r, v, d, nv, nd


def synth_rollout(T, N, D, A, seed=1234, continuous=True, p_done=0.02):
    """Synthetic rollout tensors of SURVEY section 8d (numpy legacy stream so CPU and GPU box agree)."""
    rs = np.random.RandomState(seed)
    out = dict(
        states=rs.standard_normal((T, N, D)).astype(np.float32),
        actions=(rs.standard_normal((T, N, A)).astype(np.float32) if continuous
                 else rs.randint(0, A, size=(T, N)).astype(np.float32)),
        values=rs.standard_normal((T, N)).astype(np.float32),
        rewards=rs.standard_normal((T, N)).astype(np.float32),
        terminals=(rs.random_sample((T, N)) < p_done).astype(np.float32),
        log_probs=(-1.0 + 0.5 * rs.standard_normal((T, N))).astype(np.float32),
        next_obs=rs.standard_normal((N, D)).astype(np.float32),
        next_value=rs.standard_normal(N).astype(np.float32),
        next_done=(rs.random_sample(N) < p_done).astype(np.float32),
    )
    return out
