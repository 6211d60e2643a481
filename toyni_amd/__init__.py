"""toyni_amd -- MI355X (gfx950) backend for Toyni's BabyBear NTT and FRI fold.

Only the hot path lives here: hand-written HIP kernels behind the C ABI of
include/toyni_hip.h (csrc/), and a thin host-side mirror of the reference's `src/ntt.rs::cuda`,
`src/math/fri.rs` and `BabyBearDomain` call surface (ntt.py, fri.py, domain.py) that drives the ABI
through ctypes.  There is no CPU implementation in this package: importing it without the built
library raises.
"""
from . import _lib  # noqa: F401  (fails loudly if libtoyni_hip.so is missing)
from .ntt import (  # noqa: F401
    CudaBuffer, GpuBuffer, NttContext, PinnedArray, cuda_available, gpu_available, intt_cuda, intt_gpu, ntt_cuda, ntt_gpu,
    ntt_host_multi_gpu, ntt_slab_multi_gpu_device, ntt_slab_multi_gpu_host,
)
from .fri import fri_fold, fri_fold_device, fri_fold_ext, fri_fold_ext_device, fri_fold_layers_device, fri_fold_xs_device  # noqa: F401
from .domain import BabyBearDomain  # noqa: F401
from .merkle import MerkleTree, merkle_commit_device  # noqa: F401
from . import prover  # noqa: F401

P = 2013265921
