"""FRI pairwise fold on the GPU.  Mirrors `fri_fold(evals, xs, beta)` of src/math/fri.rs:27-48 and the
prover's fold loop (src/fibonacci.rs:220-245)."""
import numpy as np

from ._lib import check, lib


def fri_fold(evals, xs, beta: int) -> np.ndarray:
    """src/math/fri.rs:27-48: host slices (u64 elements) in, new array of len/2 out."""
    e = np.ascontiguousarray(evals, dtype=np.uint64)
    x = np.ascontiguousarray(xs, dtype=np.uint64)
    assert e.size % 2 == 0, "Evaluations length must be even"  # src/math/fri.rs:28
    assert x.size >= e.size // 2
    out = np.empty(e.size // 2, dtype=np.uint64)
    st = lib.toyni_fri_fold_host(out.ctypes.data, e.ctypes.data, e.size, x.ctypes.data, int(beta))
    assert st != 10005, "Cannot invert zero"  # src/babybear.rs:112
    check(st, "GPU FRI fold failed")
    return out


def fri_fold_device(ctx, d_evals: int, d_out: int, m: int, beta: int, x0: int, stream: int = 0) -> None:
    """Device-resident fold of one layer of size m on the points x0 * w_m^i (packed u32 pointers)."""
    check(lib.toyni_fri_fold_device(ctx.handle, d_evals, d_out, m, beta, x0, stream or None), "GPU FRI fold failed")


def fri_fold_layers_device(ctx, d_evals: int, d_layers: int, betas, shift: int, stream: int = 0) -> None:
    """The prover's fold loop: len(betas) layers of a size-ctx.n codeword on shift * <w_n>, back to back in d_layers."""
    b = np.ascontiguousarray(betas, dtype=np.uint32)
    check(lib.toyni_fri_fold_layers_device(ctx.handle, d_evals, d_layers, b.ctypes.data, b.size, shift, stream or None),
          "GPU FRI fold failed")
