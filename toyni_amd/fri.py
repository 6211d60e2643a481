"""FRI pairwise fold on the GPU.  Mirrors `fri_fold(evals, xs, beta)` of src/math/fri.rs:27-48 and the
prover's fold loop (src/fibonacci.rs:220-245)."""
import numpy as np

from ._lib import check, lib


def fri_fold(evals, xs, beta: int, out: np.ndarray = None) -> np.ndarray:
    """src/math/fri.rs:27-48: host slices (u64 elements) in, new array of len/2 out (or `out`, len/2 contiguous u64, reused)."""
    e = np.ascontiguousarray(evals, dtype=np.uint64)
    x = np.ascontiguousarray(xs, dtype=np.uint64)
    assert e.size % 2 == 0, "Evaluations length must be even"  # src/math/fri.rs:28
    assert x.size >= e.size // 2
    if out is None:
        out = np.empty(e.size // 2, dtype=np.uint64)
    assert out.dtype == np.uint64 and out.size == e.size // 2 and out.flags.c_contiguous, "out: len/2 contiguous u64"
    st = lib.toyni_fri_fold_host(out.ctypes.data, e.ctypes.data, e.size, x.ctypes.data, int(beta))
    assert st != 10005, "Cannot invert zero"  # src/babybear.rs:112
    check(st, "GPU FRI fold failed")
    return out


def fri_fold_device(ctx, d_evals: int, d_out: int, m: int, beta: int, x0: int, stream: int = 0) -> None:
    """Device-resident fold of one layer of size m on the points x0 * w_m^i (packed u32 pointers)."""
    check(lib.toyni_fri_fold_device(ctx.handle, d_evals, d_out, m, beta, x0, stream or None), "GPU FRI fold failed")


def fri_fold_xs_device(d_evals: int, d_xs: int, d_out: int, m: int, beta: int, stream: int = 0) -> None:
    """Device-resident form of the reference's own signature fri_fold(evals, xs, beta) (src/math/fri.rs:27-48): explicit points,
    only xs[0 .. m/2) is read (packed u32 pointers)."""
    check(lib.toyni_fri_fold_xs_device(d_evals, d_xs, d_out, m, beta, stream or None), "GPU FRI fold (explicit points) failed")


def fri_fold_layers_device(ctx, d_evals: int, d_layers: int, betas, shift: int, stream: int = 0) -> None:
    """The prover's fold loop: len(betas) layers of a size-ctx.n codeword on shift * <w_n>, back to back in d_layers."""
    b = np.ascontiguousarray(betas, dtype=np.uint32)
    check(lib.toyni_fri_fold_layers_device(ctx.handle, d_evals, d_layers, b.ctypes.data, b.size, shift, stream or None),
          "GPU FRI fold failed")


def fri_fold_ext(evals4, xs, beta4) -> np.ndarray:
    """src/math/fri.rs:7-25: Ext values [len, 4] and beta (4 coordinates), base-field points; returns [len/2, 4]."""
    e = np.ascontiguousarray(evals4, dtype=np.uint64).reshape(-1, 4)
    x = np.ascontiguousarray(xs, dtype=np.uint64)
    b = np.ascontiguousarray(beta4, dtype=np.uint64)
    assert e.shape[0] % 2 == 0, "Evaluations length must be even"  # src/math/fri.rs:8
    assert x.size >= e.shape[0] // 2 and b.size == 4
    out = np.empty((e.shape[0] // 2, 4), dtype=np.uint64)
    st = lib.toyni_fri_fold_ext_host(out.ctypes.data, e.ctypes.data, e.shape[0], x.ctypes.data, b.ctypes.data)
    assert st != 10005, "Cannot invert zero"
    check(st, "GPU FRI fold (Ext) failed")
    return out


def fri_fold_ext_device(ctx, d_evals: int, d_out: int, m: int, beta4, x0: int, stream: int = 0) -> None:
    """Device-resident Ext fold of one layer of m elements (4 packed u32 each) on the points x0 * w_m^i."""
    b = np.ascontiguousarray(beta4, dtype=np.uint32)
    assert b.size == 4
    check(lib.toyni_fri_fold_ext_device(ctx.handle, d_evals, d_out, m, b.ctypes.data, x0, stream or None), "GPU FRI fold (Ext) failed")
