"""mcmc_gpu_amd -- MI355X-native many-chain large-scale-chain sampler (drop-in for one hot path of
tylerrleee/mcmc-gpu: gstatsMCMC.MCMC_gpu.chain_crf_gpu.run and its proposal / likelihood / accept step).

Python here is host plumbing; the compute is hand-written HIP for gfx950 in csrc/, reached through the
C ABI of include/gsm.h (libgsm_hip.so).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
