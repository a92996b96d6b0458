"""aur_ppo_amd -- MI355X-native GAE -> shuffle -> gather -> clipped-surrogate update path of
biirving/aur_ppo (src/ppo.py, src/robot_ppo.py), behind the reference's own Python API.

The non-network arithmetic runs in hand-written HIP kernels (``csrc/``, C ABI in
``include/aurppo.h``); PyTorch-ROCm provides device memory, streams, autograd for the policy /
value networks and ``torch.distributed`` (RCCL).  There is no CPU fallback: importing
``aur_ppo_amd.hip_ops`` without the built library, or calling it without a GPU, raises.
"""
__version__ = "0.1.0"
