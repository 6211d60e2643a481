"""ctypes binding of the CPU restatement in toyni_oracle.c.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under toyni_amd/ imports this package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libtoyni_oracle.so")

P = 2013265921  # src/babybear.rs:8


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "toyni_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "clean", "all"], stdout=subprocess.DEVNULL)
    return _SO


def _load():
    try:
        build()
        return ctypes.CDLL(_SO)
    except OSError:
        build(force=True)
        return ctypes.CDLL(_SO)


_lib = _load()
_u64 = ctypes.c_uint64
_sz = ctypes.c_size_t
_p64 = ctypes.POINTER(ctypes.c_uint64)

for _name, _res, _args in [
    ("orc_bb_new", _u64, [_u64]),
    ("orc_bb_add", _u64, [_u64, _u64]),
    ("orc_bb_sub", _u64, [_u64, _u64]),
    ("orc_bb_mul", _u64, [_u64, _u64]),
    ("orc_bb_neg", _u64, [_u64]),
    ("orc_bb_pow", _u64, [_u64, _u64]),
    ("orc_bb_inverse", _u64, [_u64]),
    ("orc_bb_div", _u64, [_u64, _u64]),
    ("orc_bb_root_of_unity", _u64, [ctypes.c_uint32]),
    ("orc_ntt", ctypes.c_int, [_p64, _sz, _u64]),
    ("orc_intt", ctypes.c_int, [_p64, _sz, _u64]),
    ("orc_ntt_canonical", ctypes.c_int, [_p64, _sz]),
    ("orc_intt_canonical", ctypes.c_int, [_p64, _sz]),
    ("orc_roots_of_unity_domain", ctypes.c_int, [_p64, _sz]),
    ("orc_domain_elements", ctypes.c_int, [_p64, _sz, _u64]),
    ("orc_domain_fft", ctypes.c_int, [_p64, _sz, _p64, _sz, _u64]),
    ("orc_domain_ifft", ctypes.c_int, [_p64, _sz, _u64]),
    ("orc_fri_fold", ctypes.c_int, [_p64, _p64, _sz, _p64, _u64]),
    ("orc_fri_fold_layers", ctypes.c_int, [_p64, _p64, _sz, _u64, _p64, ctypes.c_uint]),
    ("orc_fri_fold_ext", ctypes.c_int, [_p64, _p64, _sz, _p64, _p64]),
    ("orc_sha256", None, [ctypes.c_void_p, ctypes.c_void_p, _sz]),
    ("orc_hash_leaf", None, [ctypes.c_void_p, ctypes.c_void_p, _sz]),
    ("orc_hash_node", None, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("orc_merkle_total_digests", _sz, [_sz]),
    ("orc_merkle_levels", None, [ctypes.c_void_p, ctypes.c_void_p, _sz, _sz]),
    ("orc_merkle_commit_values", None, [ctypes.c_void_p, _p64, ctypes.c_void_p, _sz]),
    ("orc_poly_eval", _u64, [_p64, _sz, _u64]),
    ("orc_fib_quotient", ctypes.c_int, [ctypes.c_void_p, _p64, _p64, _sz, _sz, _u64]),
    ("orc_fib_deep", ctypes.c_int, [_p64, _p64, _p64, _sz, _sz, _u64, _u64, _u64, _u64, _u64, _u64]),
    ("orc_merkle_get_proof", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _sz, _sz]),
    ("orc_fill_pattern_7i3", None, [_p64, _sz]),
    ("orc_fill_splitmix", None, [_p64, _sz, _u64]),
]:
    _f = getattr(_lib, _name)
    _f.restype = _res
    _f.argtypes = _args


def _ptr(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p64)


def _arr(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


# ---- field (src/babybear.rs) ----
def bb_new(v): return int(_lib.orc_bb_new(int(v) & (2**64 - 1)))
def bb_add(a, b): return int(_lib.orc_bb_add(a, b))
def bb_sub(a, b): return int(_lib.orc_bb_sub(a, b))
def bb_mul(a, b): return int(_lib.orc_bb_mul(a, b))
def bb_neg(a): return int(_lib.orc_bb_neg(a))
def bb_pow(a, e): return int(_lib.orc_bb_pow(a, e))
def bb_inverse(a):
    assert a != 0, "Cannot invert zero"  # src/babybear.rs:112
    return int(_lib.orc_bb_inverse(a))
def bb_div(a, b): return int(_lib.orc_bb_div(a, b))
def root_of_unity(log_n):
    assert log_n <= 27, "BabyBear only supports NTT up to 2^27"  # src/babybear.rs:119
    return int(_lib.orc_bb_root_of_unity(log_n))


# ---- transforms (src/ntt.rs) ----
def ntt(values, omega=None) -> np.ndarray:
    v = _arr(values).copy()
    n = v.size
    assert n and n & (n - 1) == 0, "NTT size must be power of 2"  # src/ntt.rs:26
    rc = _lib.orc_ntt_canonical(_ptr(v), n) if omega is None else _lib.orc_ntt(_ptr(v), n, omega)
    assert rc == 0
    return v


def intt(values, omega=None) -> np.ndarray:
    v = _arr(values).copy()
    n = v.size
    assert n and n & (n - 1) == 0, "NTT size must be power of 2"
    rc = _lib.orc_intt_canonical(_ptr(v), n) if omega is None else _lib.orc_intt(_ptr(v), n, omega)
    assert rc == 0
    return v


def roots_of_unity_domain(n) -> np.ndarray:
    out = np.empty(n, dtype=np.uint64)
    assert _lib.orc_roots_of_unity_domain(_ptr(out), n) == 0
    return out


# ---- domain (src/math/domain.rs) ----
def domain_elements(n, shift=1) -> np.ndarray:
    out = np.empty(n, dtype=np.uint64)
    assert _lib.orc_domain_elements(_ptr(out), n, shift) == 0
    return out


def domain_fft(coeffs, size, shift=1) -> np.ndarray:
    c = _arr(coeffs)
    out = np.empty(size, dtype=np.uint64)
    assert _lib.orc_domain_fft(_ptr(out), size, _ptr(c), c.size, shift) == 0
    return out


def domain_ifft(evals, shift=1) -> np.ndarray:
    v = _arr(evals).copy()
    assert _lib.orc_domain_ifft(_ptr(v), v.size, shift) == 0
    return v


# ---- FRI fold (src/math/fri.rs) ----
def fri_fold(evals, xs, beta) -> np.ndarray:
    e = _arr(evals)
    x = _arr(xs)
    assert e.size % 2 == 0, "Evaluations length must be even"  # src/math/fri.rs:28
    assert x.size >= e.size // 2
    out = np.empty(e.size // 2, dtype=np.uint64)
    assert _lib.orc_fri_fold(_ptr(out), _ptr(e), e.size, _ptr(x), beta) == 0
    return out


def fri_fold_layers(evals, shift, betas):
    e = _arr(evals)
    b = _arr(betas)
    n = e.size
    total = sum(n >> (k + 1) for k in range(b.size))
    out = np.empty(total, dtype=np.uint64)
    assert _lib.orc_fri_fold_layers(_ptr(out), _ptr(e), n, shift, _ptr(b), b.size) == 0
    layers, off = [], 0
    for k in range(b.size):
        m = n >> (k + 1)
        layers.append(out[off:off + m].copy())
        off += m
    return layers


def fri_fold_ext(evals, xs, beta) -> np.ndarray:
    e = _arr(evals).reshape(-1, 4)
    x = _arr(xs)
    b = _arr(beta)
    assert e.shape[0] % 2 == 0 and b.size == 4
    out = np.empty((e.shape[0] // 2, 4), dtype=np.uint64)
    assert _lib.orc_fri_fold_ext(_ptr(out), _ptr(e), e.shape[0], _ptr(x), _ptr(b)) == 0
    return out


# ---- synthetic inputs ----
def pattern_7i3(n) -> np.ndarray:
    out = np.empty(n, dtype=np.uint64)
    _lib.orc_fill_pattern_7i3(_ptr(out), n)
    return out


def splitmix(n, seed=0x70796E69) -> np.ndarray:
    out = np.empty(n, dtype=np.uint64)
    _lib.orc_fill_splitmix(_ptr(out), n, seed)
    return out


# ---- Merkle commitment (src/merkle.rs, src/fibonacci.rs:340-361) ----
def sha256(msg: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    _lib.orc_sha256(out, msg, len(msg))
    return out.raw


def hash_leaf(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    _lib.orc_hash_leaf(out, data, len(data))
    return out.raw


def hash_node(left: bytes, right: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    _lib.orc_hash_node(out, left, right)
    return out.raw


def merkle_level_sizes(n: int):
    sizes = []
    while n >= 1:
        sizes.append(n)
        if n == 1:
            break
        n = (n + 1) // 2
    return sizes


def _split_levels(flat: np.ndarray, n: int):
    out, off = [], 0
    for m in merkle_level_sizes(n):
        out.append(flat[off:off + m])
        off += m
    return out


def merkle_levels(leaves) -> list:
    """MerkleTree::new(leaves).levels for equal-length byte-string leaves: list of [count, 32] uint8 arrays."""
    n = len(leaves)
    ln = len(leaves[0])
    assert all(len(l) == ln for l in leaves)
    flat = np.zeros((int(_lib.orc_merkle_total_digests(n)), 32), dtype=np.uint8)
    buf = b"".join(leaves)
    _lib.orc_merkle_levels(flat.ctypes.data, buf, ln, n)
    return _split_levels(flat, n)


def merkle_commit_values(values, salts=None) -> list:
    """build_merkle_tree / build_unsalted_tree: levels of the tree over leaf = salt(16) || value(8, LE) or value alone."""
    v = _arr(values)
    n = v.size
    flat = np.zeros((int(_lib.orc_merkle_total_digests(n)), 32), dtype=np.uint8)
    sp = None
    if salts is not None:
        s = np.ascontiguousarray(salts, dtype=np.uint8).reshape(n, 16)
        sp = s.ctypes.data
    _lib.orc_merkle_commit_values(flat.ctypes.data, _ptr(v), sp, n)
    return _split_levels(flat, n)


# ---- pointwise prover steps (src/fibonacci.rs:133-150,186-198; src/math/polynomial.rs:134-144; src/merkle.rs:50-80) ----
def poly_eval(coeffs, x) -> int:
    c = _arr(coeffs)
    return int(_lib.orc_poly_eval(_ptr(c), c.size, int(x)))


def fib_quotient(trace_lde, n, shift):
    """(c_evals, q_evals) on the coset shift * <w_N>, N = len(trace_lde)."""
    t = _arr(trace_lde)
    c = np.empty(t.size, dtype=np.uint64)
    q = np.empty(t.size, dtype=np.uint64)
    rc = _lib.orc_fib_quotient(c.ctypes.data, _ptr(q), _ptr(t), t.size, n, shift)
    assert rc != -3, "Cannot invert zero"
    assert rc == 0
    return c, q


def fib_deep(trace_lde, q_evals, n, shift, z, t_z, t_gz, t_ggz, q_z) -> np.ndarray:
    t, q = _arr(trace_lde), _arr(q_evals)
    out = np.empty(t.size, dtype=np.uint64)
    rc = _lib.orc_fib_deep(_ptr(out), _ptr(t), _ptr(q), t.size, n, shift, z, t_z, t_gz, t_ggz, q_z)
    assert rc != -3, "Cannot invert zero"
    assert rc == 0
    return out


def merkle_get_proof(levels, index):
    """MerkleTree::get_proof on the levels of merkle_levels / merkle_commit_values: (path digests, position flags) or None."""
    n = len(levels[0])
    flat = np.ascontiguousarray(np.concatenate(levels), dtype=np.uint8)
    depth = len(levels) - 1
    path = np.zeros((max(depth, 1), 32), dtype=np.uint8)
    pos = np.zeros(max(depth, 1), dtype=np.uint8)
    d = _lib.orc_merkle_get_proof(path.ctypes.data, pos.ctypes.data, flat.ctypes.data, n, index)
    if d < 0:
        return None
    return [path[l].tobytes() for l in range(d)], [bool(pos[l]) for l in range(d)]
