"""Env-sharded data parallelism: one process per GPU, ``torch.distributed`` backend "nccl"
(= RCCL over xGMI).  New relative to the reference, which is single-process; semantics are
SURVEY section 8e:

* rank r owns envs ``[r*N/W, (r+1)*N/W)`` -- GAE, shuffle, gather, adv-norm and loss are local;
* exactly one exchange per optimizer step: SUM all-reduce of the flat gradient bucket, divided
  by W, then the identical clip + Adam on every rank (norm taken on the reduced gradient);
* with ``target_kl`` set, one extra 1-float all-reduce (mean) per epoch so ranks stop together.

xGMI is point-to-point (7 links/GPU): the 68 KB MLP bucket is latency-bound, so it goes out as a
single message; nothing here is bucketed or ring-tuned.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*.
    Returns (rank, local_rank, world).  No-op for WORLD_SIZE=1."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_envs(num_envs, rank_, world):
    """Contiguous env shard [lo, hi) of rank_; requires num_envs % world == 0 so every rank runs
    the same number of minibatch steps (collectives stay matched)."""
    if num_envs % world != 0:
        raise ValueError(f"num_envs={num_envs} is not divisible by world size {world}")
    per = num_envs // world
    return rank_ * per, (rank_ + 1) * per


def allreduce_mean_(flat, world=None, force=False):
    """In-place mean over ranks of one flat tensor (the gradient bucket).  ``force``: issue the collective even in a
    process group of one (rehearsal of the launch path, as ``allreduce_sum_``)."""
    w = world_size() if world is None else world
    if w > 1 or (force and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if w > 1:
            flat.mul_(1.0 / w)
    return flat


def allreduce_sum_(flat, world=None, force=False):
    """In-place SUM over ranks; the caller folds the 1/W into its next kernel (``hip_ops.mlp_ppo_apply``).
    ``force``: issue the collective even in a process group of one (rehearsal of the launch path)."""
    w = world_size() if world is None else world
    if w > 1 or (force and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_choice():
    """Which exchange carries the MLP policy's gradient bucket between ranks: ``"rccl"`` (default: ``torch.distributed``'s
    all-reduce, a ring / tree over xGMI) or ``"p2p"`` (``AURPPO_DP_ALLREDUCE=p2p``: the one-shot exchange over HIP-IPC peer
    memory of csrc/p2p.hip -- one launch, one hop, no collective library on the path; SURVEY 8e's plan B for a 68 KB
    message).  Buckets beyond the one-shot kernel's reach (CNN policies) stay on the process group's all-reduce either way."""
    v = os.environ.get("AURPPO_DP_ALLREDUCE", "rccl").strip().lower()
    if v not in ("rccl", "p2p"):
        raise ValueError(f"AURPPO_DP_ALLREDUCE={v!r}: expected 'rccl' or 'p2p'")
    return v


def make_p2p_exchange(ops, max_floats, device):
    """A ``P2PExchange`` over the default process group's ranks: the IPC handles travel through one ``all_gather_object`` at
    set-up (whatever the group's backend is); nothing after that touches ``torch.distributed``."""
    def exchange(mine):
        if world_size() == 1:
            return [mine]
        out = [None] * world_size()
        dist.all_gather_object(out, mine)
        return out
    # A rank whose set-up fails (peer mapping refused, no IPC support) must not leave the others waiting in the next collective:
    # every rank reports, and all of them raise together if any of them could not open its peers.
    x, err = None, None
    try:
        x = ops.P2PExchange(rank(), world_size(), max_floats, device, exchange)
    except Exception as e:       # noqa: BLE001 -- reported below, on every rank
        err = f"rank {rank()}: {type(e).__name__}: {e}"
    if world_size() > 1:
        errs = [None] * world_size()
        dist.all_gather_object(errs, err)
        bad = [e for e in errs if e]
        if bad:
            if x is not None:
                x.close()
            raise RuntimeError("one-shot gradient exchange could not be set up: " + " | ".join(bad)[:600])
    elif err:
        raise RuntimeError(err)
    return x


def collectives_capturable():
    """True when the default group's all-reduce can be recorded into a hipGraph: RCCL enqueues device work only;
    gloo stages through host memory and cannot be captured.  A process without a group has nothing to capture."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_backend() == "nccl"


def barrier():
    if world_size() > 1:
        dist.barrier()


def shutdown():
    """Tear the default group down (a clean exit: RCCL warns about leaked communicators otherwise)."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
