"""Randomised differential test of the device entry points against the oracle: sizes, batches, directions, coset
shifts, in place / out of place and chunking drawn from a seeded generator, so every run checks the same few hundred
cases -- chosen to wander over the dispatch table's seams (1 / 2 / 3-pass plans, 8 / 16 / 32-wide tiles of the
1024-point passes, ragged single-pass row tiles, batches that are not multiples of anything).  Bit-exact."""
import os

import numpy as np
import pytest

import oracle
from oracle import P
from test_gpu_parity import DevBuf, ta  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _oracle_rows(x, n, inverse, shift):
    rows = x.reshape(-1, n).astype(np.uint64)
    out = np.empty_like(rows)
    for b, row in enumerate(rows):
        if shift == 1:
            out[b] = oracle.intt(row) if inverse else oracle.ntt(row)
        else:
            out[b] = oracle.domain_ifft(row, shift) if inverse else oracle.domain_fft(row, n, shift)
    return out.reshape(-1)


def test_random_shapes_against_oracle(ta):
    # TOYNI_FUZZ_SEED / TOYNI_FUZZ_CASES: soak runs over other seeds and more cases (defaults: the fixed 220 of every run)
    rng = np.random.default_rng(int(os.environ.get("TOYNI_FUZZ_SEED", str(0x70796E69)), 0))
    budget = 1 << 23                      # elements per case (oracle time stays in the tens of milliseconds)
    cases = 0
    ncases = int(os.environ.get("TOYNI_FUZZ_CASES", "220"))
    for _ in range(ncases):
        log_n = int(rng.integers(0, 21))
        n = 1 << log_n
        max_batch = max(1, min(70, budget >> log_n))
        batch = int(rng.integers(1, max_batch + 1))
        inverse = bool(rng.integers(0, 2))
        shift = 1 if rng.integers(0, 3) else int(rng.integers(2, P))
        inplace = bool(rng.integers(0, 2))
        chunk = None if rng.integers(0, 4) else int(n * rng.integers(1, batch + 1))
        x = rng.integers(0, P, size=n * batch, dtype=np.uint32)
        want = _oracle_rows(x, n, inverse, shift)
        ctx = ta.ntt.get_or_create_ctx(n)
        if chunk is not None:
            ctx.set_chunk(chunk)
        a = DevBuf(ta, x.nbytes)
        b = a if inplace else DevBuf(ta, x.nbytes)
        try:
            a.upload(x)
            ctx.run_device(a.ptr, b.ptr, batch, inverse, shift=shift)
            ctx.synchronize()
            got = b.download(np.uint32, x.size)
            if not inplace:
                assert (a.download(np.uint32, x.size) == x).all(), "out-of-place transform modified its input"
        finally:
            a.free()
            if b is not a:
                b.free()
            ctx.set_chunk(0)
        assert (got.astype(np.uint64) == want).all(), \
            f"log_n={log_n} batch={batch} inverse={inverse} shift={shift} inplace={inplace} chunk={chunk}"
        cases += 1
    assert cases == ncases


def test_random_folds_against_oracle(ta):
    rng = np.random.default_rng(int(os.environ.get("TOYNI_FUZZ_SEED", "0"), 0) ^ 0xF01D)   # default: the fixed 60 of every run
    for case in range(int(os.environ.get("TOYNI_FUZZ_CASES", "60"))):
        log_m = int(rng.integers(1, 19))
        m = 1 << log_m
        shift = int(rng.integers(1, P))
        beta = int(rng.integers(0, P))
        evals = rng.integers(0, P, size=m, dtype=np.uint64)
        xs = oracle.domain_elements(m, shift)
        want = oracle.fri_fold(evals, xs, beta)
        assert (ta.fri_fold(evals, xs, beta) == want).all(), f"explicit points m={m}"
        # structured points on a context at least as large as the layer (layer k of a larger codeword)
        log_ctx = int(rng.integers(log_m, 21))
        ctx = ta.ntt.get_or_create_ctx(1 << log_ctx)
        e32 = evals.astype(np.uint32)
        a, o = DevBuf(ta, e32.nbytes), DevBuf(ta, max(e32.nbytes // 2, 4))
        try:
            a.upload(e32)
            ta.fri_fold_device(ctx, a.ptr, o.ptr, m, beta, shift)
            ctx.synchronize()
            got = o.download(np.uint32, m // 2)
        finally:
            a.free()
            o.free()
        assert (got.astype(np.uint64) == want).all(), f"structured points m={m} ctx=2^{log_ctx} shift={shift}"
        # the reference's signature takes ANY even length (src/math/fri.rs:28): ragged lengths around both explicit-point kernels' gates
        # (half a multiple of 4 or not; below and above 2^18 pairs), a few zero points, arbitrary (non-coset) points
        m2 = 2 * int(rng.integers(1, 1 << int(rng.integers(1, 21))))
        e2 = rng.integers(0, P, size=m2, dtype=np.uint64)
        x2 = rng.integers(1, P, size=m2 // 2, dtype=np.uint64)
        want2 = oracle.fri_fold(e2, x2, beta)
        assert (ta.fri_fold(e2, x2, beta) == want2).all(), f"explicit points, ragged m={m2}"
        if case % 4 == 0:   # whole quads around the 16-per-inversion kernel's gate (2^18 pairs): a ragged last chunk, zero points among the points
            m3 = 8 * int(rng.integers(1 << 15, 1 << 17))
            e3 = rng.integers(0, P, size=m3, dtype=np.uint64)
            x3 = rng.integers(1, P, size=m3 // 2, dtype=np.uint64)
            assert (ta.fri_fold(e3, x3, beta) == oracle.fri_fold(e3, x3, beta)).all(), f"explicit points, whole quads m={m3}"
            m4 = 2 * int(rng.integers(1, 6000))   # Ext values, explicit points (src/math/fri.rs:7-25): ragged over the kernel's 1024-element chunks
            e4 = rng.integers(0, P, size=(m4, 4), dtype=np.uint64)
            x4 = rng.integers(1, P, size=m4 // 2, dtype=np.uint64)
            b4 = rng.integers(0, P, size=4, dtype=np.uint64)
            assert (ta.fri_fold_ext(e4, x4, b4) == oracle.fri_fold_ext(e4, x4, b4)).all(), f"Ext explicit points, ragged m={m4}"


def test_random_low_degree_extensions_against_oracle(ta):
    rng = np.random.default_rng(int(os.environ.get("TOYNI_FUZZ_SEED", "0"), 0) ^ 0x1DE)
    for _ in range(int(os.environ.get("TOYNI_FUZZ_CASES", "80"))):
        log_n = int(rng.integers(1, 21))
        n = 1 << log_n
        log_blowup = int(rng.integers(0, log_n + 1))
        n_in = n >> log_blowup
        batch = int(rng.integers(1, max(2, min(20, (1 << 22) >> log_n) + 1)))
        shift = 7 if rng.integers(0, 2) else int(rng.integers(1, P))
        coeffs = rng.integers(0, P, size=n_in * batch, dtype=np.uint32)
        ctx = ta.ntt.get_or_create_ctx(n)
        a, o = DevBuf(ta, coeffs.nbytes), DevBuf(ta, 4 * n * batch)
        try:
            a.upload(coeffs)
            ctx.lde_device(a.ptr, o.ptr, batch, log_blowup, shift)
            ctx.synchronize()
            got = o.download(np.uint32, n * batch)
        finally:
            a.free()
            o.free()
        for b in range(batch):
            want = oracle.domain_fft(coeffs[b * n_in:(b + 1) * n_in].astype(np.uint64), n, shift)
            assert (got[b * n:(b + 1) * n].astype(np.uint64) == want).all(), f"log_n={log_n} log_blowup={log_blowup} batch={batch} shift={shift} b={b}"


def test_random_ext_shapes_against_oracle(ta):
    """fft_ext / ifft_ext (src/math/domain.rs:129-151) through the interleaved passes: random sizes, vector counts, directions, coset
    shifts, in place / out of place; every coordinate of every vector is the oracle's transform of that coordinate column."""
    rng = np.random.default_rng(int(os.environ.get("TOYNI_FUZZ_SEED", str(0xE87)), 0))
    budget = 1 << 21                      # Ext elements per case
    for _ in range(int(os.environ.get("TOYNI_FUZZ_CASES", "70"))):
        log_n = int(rng.integers(0, 20))
        n = 1 << log_n
        vecs = int(rng.integers(1, max(1, min(40, budget >> log_n)) + 1))
        inverse = bool(rng.integers(0, 2))
        shift = 1 if rng.integers(0, 3) else int(rng.integers(2, P))
        inplace = bool(rng.integers(0, 2))
        chunk = None if rng.integers(0, 4) else int(n * rng.integers(1, 4 * vecs + 1))   # in base-field elements: rounded up to whole vectors
        x = rng.integers(0, P, size=(vecs, n, 4), dtype=np.uint32)
        ctx = ta.ntt.get_or_create_ctx(n)
        if chunk is not None:
            ctx.set_chunk(chunk)
        a = DevBuf(ta, x.nbytes)
        b = a if inplace else DevBuf(ta, x.nbytes)
        try:
            a.upload(x)
            ctx.run_device_ext_batch(a.ptr, b.ptr, vecs, inverse, shift=shift)
            ctx.synchronize()
            got = b.download(np.uint32, x.size).reshape(x.shape)
            if not inplace:
                assert (a.download(np.uint32, x.size).reshape(x.shape) == x).all(), "out-of-place transform modified its input"
        finally:
            a.free()
            if b is not a:
                b.free()
            ctx.set_chunk(0)
        for v in range(vecs):
            want = _oracle_rows(np.ascontiguousarray(x[v].T), n, inverse, shift).reshape(4, n)
            assert (got[v].T.astype(np.uint64) == want).all(), f"log_n={log_n} vecs={vecs} inverse={inverse} shift={shift} inplace={inplace} chunk={chunk} v={v}"
