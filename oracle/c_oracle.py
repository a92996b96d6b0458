"""ctypes view of oracle/_build/libaurppo_oracle.so (TEST INFRASTRUCTURE -- see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# AURPPO_ORACLE_SO selects another build of the same source (the ASan/UBSan one of `make -C oracle asan`)
_SO = os.environ.get("AURPPO_ORACLE_SO") or os.path.join(_HERE, "_build", "libaurppo_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_mt_sizeof.restype = C.c_size_t
    return _lib


def _p(a, ct=C.c_float):
    return a.ctypes.data_as(C.POINTER(ct))


def gae(r, v, d, nv, nd, gamma, lam, mode=0):
    r, v, d, nv, nd = (np.ascontiguousarray(x, dtype=np.float32) for x in (r, v, d, nv, nd))
    T, N = r.shape
    adv, ret = np.empty_like(r), np.empty_like(r)
    lib().oracle_gae_f32(_p(r), _p(v), _p(d), _p(nv), _p(nd), _p(adv), _p(ret), T, N, C.c_double(gamma),
                         C.c_double(lam), mode)
    return ret, adv


class MT:
    def __init__(self, seed):
        self.buf = C.create_string_buffer(lib().oracle_mt_sizeof())
        lib().oracle_mt_seed(self.buf, C.c_uint32(seed))

    def shuffle(self, x):
        assert x.dtype == np.int32 and x.flags.c_contiguous
        lib().oracle_mt_shuffle_i32(self.buf, _p(x, C.c_int32), len(x))

    def get_state(self):
        key = np.empty(624, np.uint32)
        pos = C.c_int32()
        lib().oracle_mt_get_state(self.buf, _p(key, C.c_uint32), C.byref(pos))
        return key, pos.value


def gather(idx, src):
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    src = np.ascontiguousarray(src, dtype=np.float32)
    re = int(np.prod(src.shape[1:])) if src.ndim > 1 else 1
    dst = np.empty((len(idx),) + src.shape[1:], np.float32)
    lib().oracle_gather_f32(_p(idx, C.c_int32), len(idx), _p(src), _p(dst), re)
    return dst


def ppo_loss(newlogp, oldlogp, adv, newv, oldv, ret, entropy, clip, ent_coef, vf_coef, norm_adv=True, vloss_mode=1):
    a = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1) for x in (newlogp, oldlogp, adv, newv, oldv, ret, entropy)]
    M = len(a[0])
    out = np.empty(9, np.float32)
    g = [np.empty(M, np.float32) for _ in range(3)]
    lib().oracle_loss_fwd_bwd_f32(*[_p(x) for x in a], M, C.c_double(clip), C.c_double(ent_coef), C.c_double(vf_coef),
                                  int(bool(norm_adv)), int(vloss_mode), _p(out), _p(g[0]), _p(g[1]), _p(g[2]))
    return out, g[0], g[1], g[2]


def grad_norm_clip(g, max_norm):
    g = np.array(g, dtype=np.float32, copy=True)
    n = C.c_float()
    lib().oracle_grad_norm_clip_f32(_p(g), C.c_int64(g.size), C.c_double(max_norm), C.byref(n))
    return g, n.value
