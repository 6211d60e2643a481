"""Device-resident steps of the Fibonacci prover around the hot path (include/toyni_hip.h 3c): one FRI round with its
commitment, constraint / quotient / DEEP evaluation on the LDE coset, polynomial evaluation at the out-of-domain points, and
Merkle openings.  Mirrors the corresponding lines of `StarkProver::generate_proof` (src/fibonacci.rs:133-150,186-198,222-245,
366-375) as calls on packed-u32 device pointers; no host arithmetic, no CPU path."""
import ctypes

import numpy as np

from ._lib import check, lib


def fri_fold_commit_device(ctx, d_evals: int, d_out: int, m: int, beta: int, x0: int, d_salts: int, d_levels: int, stream: int = 0) -> None:
    """fold a layer of m values and commit the folded layer (all tree levels to d_levels) in one call."""
    check(lib.toyni_fri_fold_commit_device(ctx.handle, d_evals, d_out, m, beta, x0, d_salts or None, d_levels, stream or None),
          "GPU fold + commit failed")


def fri_commit_phase_device(ctx, d_layer0: int, m0: int, x0: int, final_size: int, d_salts: int, challenge, d_layers: int, d_levels: int,
                            stream: int = 0):
    """The fold loop of the commit phase (src/fibonacci.rs:222-245) in one call.  `challenge(round, prev_root: bytes | None,
    want_beta: bool) -> int` is the caller's transcript: absorb prev_root when it is not None, return the squeezed beta when
    want_beta (anything otherwise).  Returns the list of roots (bytes), one per round."""
    from ._lib import FRI_CHALLENGE_FN
    failure = []

    def _cb(_user, rnd, root_ptr, beta_ptr):
        try:
            root = bytes(root_ptr[:32]) if root_ptr else None
            beta = challenge(int(rnd), root, bool(beta_ptr))
            if beta_ptr:
                beta_ptr[0] = int(beta)
            return 0
        except BaseException as e:  # noqa: BLE001  (nothing may unwind through the C frames -- KeyboardInterrupt included: ctypes
            failure.append(e)       # would print and swallow it, return 0, and the C loop would go on with beta = 0; re-raised below)
            return 1

    cb = FRI_CHALLENGE_FN(_cb)
    rounds = ctypes.c_uint(0)
    max_rounds = max(0, (m0 // max(final_size, 1)).bit_length() - 1)
    roots = np.zeros(max(max_rounds, 1) * 32, dtype=np.uint8)
    rc = lib.toyni_fri_commit_phase_device(ctx.handle, d_layer0, m0, x0, final_size, d_salts or None, ctypes.cast(cb, ctypes.c_void_p), None,
                                           d_layers, d_levels, roots.ctypes.data, ctypes.byref(rounds), stream or None)
    if failure:
        raise failure[0]
    check(rc, "GPU FRI commit phase failed")
    return [roots[32 * k:32 * k + 32].tobytes() for k in range(rounds.value)]


def fib_quotient_device(ctx, d_trace_lde: int, d_c_evals: int, d_q_evals: int, log_blowup: int, shift: int, stream: int = 0) -> None:
    check(lib.toyni_fib_quotient_device(ctx.handle, d_trace_lde, d_c_evals or None, d_q_evals, log_blowup, shift, stream or None),
          "GPU constraint / quotient evaluation failed")


def fib_deep_device(ctx, d_trace_lde: int, d_q_evals: int, d_out: int, log_blowup: int, shift: int, z: int, ood, stream: int = 0) -> None:
    o = np.ascontiguousarray(ood, dtype=np.uint32)
    assert o.size == 4
    check(lib.toyni_fib_deep_device(ctx.handle, d_trace_lde, d_q_evals, d_out, log_blowup, shift, z, o.ctypes.data, stream or None),
          "GPU DEEP evaluation failed")


def poly_eval_device(ctx, d_coeffs: int, ncoeffs: int, points, d_out: int, stream: int = 0) -> None:
    p = np.ascontiguousarray(points, dtype=np.uint32)
    check(lib.toyni_poly_eval_device(ctx.handle, d_coeffs, ncoeffs, p.ctypes.data, p.size, d_out, stream or None), "GPU polynomial evaluation failed")


def merkle_open_record_bytes(n: int) -> int:
    return lib.toyni_merkle_open_record_bytes(n)


def merkle_open_device(d_levels: int, n: int, d_values: int, d_salts: int, d_indices: int, nidx: int, d_out: int, stream: int = 0) -> None:
    check(lib.toyni_merkle_open_device(d_levels, n, d_values, d_salts or None, d_indices, nidx, d_out, stream or None), "GPU Merkle opening failed")


class _OpenGroup(ctypes.Structure):   # toyni_merkle_open_group (include/toyni_hip.h 3c)
    _fields_ = [("d_levels", ctypes.c_void_p), ("n", ctypes.c_size_t), ("d_values", ctypes.c_void_p), ("d_salts", ctypes.c_void_p),
                ("d_indices", ctypes.c_void_p), ("nidx", ctypes.c_size_t), ("d_out", ctypes.c_void_p)]


def merkle_open_groups_device(groups, stream: int = 0) -> None:
    """The openings of several trees in one launch per 32 trees.  groups: iterable of (d_levels, n, d_values, d_salts, d_indices, nidx,
    d_out), every field as in merkle_open_device."""
    arr = (_OpenGroup * len(groups))(*[_OpenGroup(lv, n, v, s or None, ix, k, o) for lv, n, v, s, ix, k, o in groups])
    check(lib.toyni_merkle_open_groups_device(arr, len(groups), stream or None), "GPU Merkle openings failed")


def parse_openings(raw: np.ndarray, n: int, indices, salted: bool):
    """Records of toyni_merkle_open_device -> the fields of MerkleOpening (src/fibonacci.rs:366-375)."""
    depth = 0
    m = n
    while m > 1:
        m = (m + 1) // 2
        depth += 1
    rec = merkle_open_record_bytes(n)
    raw = np.asarray(raw, dtype=np.uint8).reshape(len(indices), rec)
    out = []
    for k, index in enumerate(indices):
        r = raw[k]
        path = [r[32 * l:32 * (l + 1)].tobytes() for l in range(depth)]
        salt = r[32 * depth:32 * depth + 16].tobytes() if salted else b""
        value = int.from_bytes(r[32 * depth + 16:32 * depth + 24].tobytes(), "little")
        position = [bool(b) for b in r[32 * depth + 24:32 * depth + 24 + depth]]
        out.append({"index": int(index), "value": value, "path": path, "position": position, "salt": salt})
    return out
