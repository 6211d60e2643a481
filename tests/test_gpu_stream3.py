"""Round 5: n = 2^21 as TWO sweeps -- 1024-point column pass + the streaming three-step 2048-point closing pass
(Pass3<KIND_ROW_T, 5, 3, 3, 4>, ntt_pass3s_kernel) -- and the low-degree extensions that ride on the same two-pass plans
(2^16 -> 2^21, what the prover and a downstream zkvm call per column: src/math/domain.rs:107-123, src/fibonacci.rs:101-103,125-128;
2^17 -> 2^22 through the 16-wide 2048-point column shape with implied zero padding).

Every transform of every batch is compared with the oracle (src/ntt.rs:24-66 restated), bit-exact; the launched kernel symbols are
asserted, so a silent fall-back to the three-pass plan fails the test."""
import numpy as np
import pytest

import oracle
from oracle import P
from test_gpu_parity import DevBuf, ta  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

STREAM_ROW = "ntt_pass3s_kernelIN5toyni5Pass3ILi1ELi5ELi3ELi3ELi4E"   # closing 2048-point pass, 16-row tiles
STREAM_COL = "ntt_pass3s_kernelIN5toyni5Pass3ILi0ELi5ELi3ELi3ELi4E"   # 2048-point column pass (LDE first pass of n = 2^22)


def _launched(ta, prefix):
    return [k for k in ta._lib.launched_kernels() if prefix in k]


@pytest.mark.parametrize("batch,shift,inplace,chunk_transforms", [
    (4, 1, True, None),        # the smallest launch that takes the streaming shape (2^7 32-wide tiles' worth)
    (5, 7, False, None),       # odd batch, coset (COSET_SHIFT, src/fibonacci.rs:16), out of place
    (13, 1234567891, True, 6),  # chunked: launches of 6, 6 and 1 transforms -- the last one falls back to the latency shapes
])
def test_two_pass_2p21_every_transform_against_the_oracle(ta, batch, shift, inplace, chunk_transforms):
    n = 1 << 21
    rng = np.random.default_rng(0x5721 + batch)
    x = rng.integers(0, P, size=n * batch, dtype=np.uint32)
    x[:n] = (7 * np.arange(n, dtype=np.uint64) + 3) % P          # the reference's own input pattern (src/ntt.rs:272) in transform 0
    ctx = ta.ntt.get_or_create_ctx(n)
    assert ctx.passes == 3 and ctx.passes_for(batch) == 2 and ctx.passes_for(1) == 2
    if chunk_transforms:
        ctx.set_chunk(chunk_transforms * n)
    a = DevBuf(ta, x.nbytes)
    b = a if inplace else DevBuf(ta, x.nbytes)
    try:
        a.upload(x)
        ctx.run_device(a.ptr, b.ptr, batch, False, shift=shift)
        ctx.synchronize()
        y = b.download(np.uint32, x.size)
        for t in range(batch):
            want = oracle.domain_fft(x[t * n:(t + 1) * n].astype(np.uint64), n, shift)
            assert (y[t * n:(t + 1) * n] == want).all(), f"2^21 x{batch} shift {shift}: transform {t} differs from the oracle"
        if not inplace:
            assert (a.download(np.uint32, x.size) == x).all(), "out-of-place transform modified its input"
        ctx.run_device(b.ptr, b.ptr, batch, True, shift=shift)
        ctx.synchronize()
        assert (b.download(np.uint32, x.size) == x).all(), "inverse(forward(x)) != x"
    finally:
        ctx.set_chunk(0)
        a.free()
        if b is not a:
            b.free()
    assert _launched(ta, STREAM_ROW), "the streaming 2048-point closing pass never ran"


@pytest.mark.parametrize("log_n,z,batch", [
    (21, 5, 5),     # the prover's LDE (2^16 -> 2^21, blow-up 32), ragged batch
    (21, 5, 64),    # the batch bench.py times (extras.lde_64x_2^16_to_2^21); sampled vectors
    (21, 2, 4),     # blow-up 4
    (22, 5, 3),     # 2^17 -> 2^22: 16-wide 2048-point column pass with a zero fraction of 2^5
    (22, 1, 2),     # blow-up 2
    (22, 7, 4),     # blow-up 128: beyond five bits (the LZ = 5 variant with its row guard)
])
def test_lde_through_the_two_pass_plans(ta, log_n, z, batch):
    n, n_in = 1 << log_n, (1 << log_n) >> z
    rng = np.random.default_rng(0x1DE0 + log_n * 64 + z)
    c = rng.integers(0, P, size=n_in * batch, dtype=np.uint32)
    ctx = ta.ntt.get_or_create_ctx(n)
    a, b = DevBuf(ta, c.nbytes), DevBuf(ta, 4 * n * batch)
    try:
        a.upload(c)
        ctx.lde_device(a.ptr, b.ptr, batch, z, 7)
        ctx.synchronize()
        ts = range(batch) if batch <= 8 else sorted({0, 1, batch // 2, batch - 1})
        for t in ts:
            got = b.download(np.uint32, n, offset=4 * n * t)
            want = oracle.domain_fft(c[t * n_in:(t + 1) * n_in].astype(np.uint64), n, 7)   # zero-pads, scales by 7^i, transforms
            assert (got == want).all(), f"LDE 2^{log_n - z} -> 2^{log_n} x{batch}: vector {t} differs from the oracle"
    finally:
        a.free()
        b.free()
    assert _launched(ta, STREAM_ROW)
    if log_n == 22:
        assert any(f"EELi{min(z, 5)}EEv" in k for k in _launched(ta, STREAM_COL)), "the zero-fraction variant of the 2048-point column pass never ran"


@pytest.mark.parametrize("vectors,shift", [(1, 1), (3, 7)])
def test_ext_vectors_of_2p21_through_the_two_pass_plan(ta, vectors, shift):
    """fft_ext / ifft_ext (src/math/domain.rs:129-151) at n = 2^21: each coordinate of an AoS vector is the base transform of that
    coordinate -- through the interleaved 1024-point column pass and the interleaved streaming 2048-point closing pass."""
    n = 1 << 21
    rng = np.random.default_rng(0xE57 + vectors)
    x = rng.integers(0, P, size=4 * n * vectors, dtype=np.uint32)
    ctx = ta.ntt.get_or_create_ctx(n)
    a, b = DevBuf(ta, x.nbytes), DevBuf(ta, x.nbytes)
    try:
        a.upload(x)
        ctx.run_device_ext_batch(a.ptr, b.ptr, vectors, False, shift=shift)
        ctx.synchronize()
        y = b.download(np.uint32, x.size).reshape(vectors, n, 4)
        xx = x.reshape(vectors, n, 4)
        for v in range(vectors):
            for k in (range(4) if v == 0 else [v % 4]):
                want = oracle.domain_fft(xx[v, :, k].astype(np.uint64), n, shift)
                assert (y[v, :, k] == want).all(), f"ext 2^21 x{vectors} shift {shift}: vector {v} coordinate {k}"
        ctx.run_device_ext_batch(b.ptr, b.ptr, vectors, True, shift=shift)
        ctx.synchronize()
        assert (b.download(np.uint32, x.size) == x).all(), "ifft_ext(fft_ext(x)) != x"
        # LDE of Ext coefficients: 2^16 -> 2^21 (blow-up 32), the same buffers
        c = x[:4 * (n >> 5) * vectors]
        a.upload(c)
        ctx.lde_ext_device(a.ptr, b.ptr, vectors, 5, 7)
        ctx.synchronize()
        y = b.download(np.uint32, 4 * n * vectors).reshape(vectors, n, 4)
        cc = c.reshape(vectors, n >> 5, 4)
        for v in range(vectors):
            k = (v + 1) % 4
            assert (y[v, :, k] == oracle.domain_fft(cc[v, :, k].astype(np.uint64), n, 7)).all(), f"ext LDE 2^16 -> 2^21: vector {v} coordinate {k}"
    finally:
        a.free()
        b.free()
    assert _launched(ta, "ntt_pass3s_kernelIN5toyni5Pass3ILi1ELi5ELi3ELi3ELi4ELb0ELi2E"), "the interleaved streaming closing pass never ran"


@pytest.mark.parametrize("z,vectors", [(5, 1), (3, 2)])
def test_ext_lde_to_2p22_through_the_interleaved_column_shape(ta, z, vectors):
    """fft_ext of zero-padded Ext coefficient vectors on the LDE coset (src/math/domain.rs:107-123,129-151), 2^(22-z) -> 2^22: the
    interleaved 16-wide 2048-point column shape (zero-fraction variant) + the interleaved streaming closing pass."""
    n, n_in = 1 << 22, (1 << 22) >> z
    rng = np.random.default_rng(0xE22 + z)
    c = rng.integers(0, P, size=(vectors, n_in, 4), dtype=np.uint32)
    ctx = ta.ntt.get_or_create_ctx(n)
    a, b = DevBuf(ta, c.nbytes), DevBuf(ta, 16 * n * vectors)
    try:
        a.upload(c)
        ctx.lde_ext_device(a.ptr, b.ptr, vectors, z, 7)
        ctx.synchronize()
        y = b.download(np.uint32, 4 * n * vectors).reshape(vectors, n, 4)
    finally:
        a.free()
        b.free()
    for v in range(vectors):
        for k in (range(4) if v == 0 else [2]):
            want = oracle.domain_fft(np.ascontiguousarray(c[v, :, k]).astype(np.uint64), n, 7)
            assert (y[v, :, k] == want).all(), f"Ext LDE 2^{22 - z} -> 2^22: vector {v} coordinate {k}"
    assert any(f"EELi{z}EEv" in k for k in _launched(ta, "ntt_pass3s_kernelIN5toyni5Pass3ILi0ELi5ELi3ELi3ELi4ELb0ELi2E")), "interleaved column shape never ran"


def test_plain_2p22_keeps_the_three_pass_plan_and_lone_transforms_the_latency_shapes(ta):
    import os
    if any(k.startswith("TOYNI_") and k not in ("TOYNI_LAUNCH_LOG", "TOYNI_FUZZ_SEED", "TOYNI_FUZZ_CASES") for k in os.environ):
        pytest.skip("the default dispatch's plan choices are asserted only without dispatch knobs (tools/knob_soak.sh)")
    ctx22 = ta.ntt.get_or_create_ctx(1 << 22)
    assert ctx22.passes == 3 and ctx22.passes_for(1) == 2 and ctx22.passes_for(2) == 2 and ctx22.passes_for(64) == 3
    ctx21 = ta.ntt.get_or_create_ctx(1 << 21)
    assert ctx21.passes_for(1) == 2 and ctx21.passes_for(1024) == 2
    ctx13 = ta.ntt.get_or_create_ctx(1 << 13)
    assert ctx13.passes == 2 and ctx13.passes_for(1) == 2 and ctx13.passes_for(1 << 12) == 1    # the single-sweep kernel from 2^25 elements


def test_random_shapes_on_the_two_pass_plans(ta):
    """Seeded differential fuzz over the sizes with a second plan (n = 2^21, 2^22): batches either side of the 2^7-tile gate, both
    directions, coset shifts, in place / out of place, chunked launches, low-degree extensions of every blow-up, Ext vectors at 2^21.
    TOYNI_FUZZ_SEED / TOYNI_FUZZ_CASES: soak runs (default: the fixed 14 cases of every session).  Every transform against the oracle."""
    import os
    rng = np.random.default_rng(int(os.environ.get("TOYNI_FUZZ_SEED", str(0x2B21)), 0))
    for case in range(int(os.environ.get("TOYNI_FUZZ_CASES", "14"))):
        log_n = 21 + int(rng.integers(0, 2))
        n = 1 << log_n
        kind = int(rng.integers(0, 4))                      # 0, 1: plain / coset transform; 2: LDE; 3: Ext (2^21) or LDE (2^22)
        shift = 1 if rng.integers(0, 3) == 0 else int(rng.integers(2, P))
        ctx = ta.ntt.get_or_create_ctx(n)
        if kind <= 1:
            batch = int(rng.integers(1, 7 if log_n == 21 else 4))
            inverse, inplace = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            chunk = None if rng.integers(0, 3) else int(n * rng.integers(1, batch + 1))
            x = rng.integers(0, P, size=n * batch, dtype=np.uint32)
            if chunk:
                ctx.set_chunk(chunk)
            a = DevBuf(ta, x.nbytes)
            b = a if inplace else DevBuf(ta, x.nbytes)
            try:
                a.upload(x)
                ctx.run_device(a.ptr, b.ptr, batch, inverse, shift=shift)
                ctx.synchronize()
                y = b.download(np.uint32, x.size)
            finally:
                ctx.set_chunk(0)
                a.free()
                if b is not a:
                    b.free()
            for t in range(batch):
                row = x[t * n:(t + 1) * n].astype(np.uint64)
                want = oracle.domain_ifft(row, shift) if inverse else oracle.domain_fft(row, n, shift)
                assert (y[t * n:(t + 1) * n] == want).all(), f"case {case}: 2^{log_n} x{batch} inverse={inverse} shift={shift} inplace={inplace} chunk={chunk} t={t}"
        elif kind == 2 or log_n == 22:
            z = int(rng.integers(1, 13))
            batch = int(rng.integers(1, 6 if log_n == 21 else 4))
            n_in = n >> z
            c = rng.integers(0, P, size=n_in * batch, dtype=np.uint32)
            a, b = DevBuf(ta, c.nbytes), DevBuf(ta, 4 * n * batch)
            try:
                a.upload(c)
                ctx.lde_device(a.ptr, b.ptr, batch, z, shift)
                ctx.synchronize()
                y = b.download(np.uint32, n * batch)
            finally:
                a.free()
                b.free()
            for t in range(batch):
                want = oracle.domain_fft(c[t * n_in:(t + 1) * n_in].astype(np.uint64), n, shift)
                assert (y[t * n:(t + 1) * n] == want).all(), f"case {case}: LDE 2^{log_n - z} -> 2^{log_n} x{batch} shift={shift} t={t}"
        else:
            vecs, inverse = int(rng.integers(1, 3)), bool(rng.integers(0, 2))
            x = rng.integers(0, P, size=(vecs, n, 4), dtype=np.uint32)
            a = DevBuf(ta, x.nbytes)
            try:
                a.upload(x)
                ctx.run_device_ext_batch(a.ptr, a.ptr, vecs, inverse, shift=shift)
                ctx.synchronize()
                y = a.download(np.uint32, x.size).reshape(x.shape)
            finally:
                a.free()
            for v in range(vecs):
                k = int(rng.integers(0, 4))
                col = np.ascontiguousarray(x[v, :, k]).astype(np.uint64)
                want = oracle.domain_ifft(col, shift) if inverse else oracle.domain_fft(col, n, shift)
                assert (y[v, :, k] == want).all(), f"case {case}: Ext 2^21 x{vecs} inverse={inverse} shift={shift} vector {v} coordinate {k}"


@pytest.mark.parametrize("batch,shift,inverse,inplace", [
    (2048, 1, False, True),       # the smallest batch that takes the kernel (TOYNI_R2048_MIN_ROWS)
    (5001, 7, False, False),      # ragged: some waves run one row more than others; forward coset, out of place
    (4099, 1234567, True, True),  # inverse coset (n^-1 rides on the output seeds)
    (3000, 1, True, False),       # plain inverse: n^-1 at the store
])
def test_n2p11_one_wave_per_transform(ta, batch, shift, inverse, inplace):
    """n = 2^11 (the reference's own test size: trace 64, blow-up 32 -> 2048 points, src/fibonacci.rs tests) in ONE sweep, one wave per
    transform and no workgroup barrier (Row2048, ntt_row2048_kernel).  EVERY transform of the batch against the oracle."""
    n = 1 << 11
    rng = np.random.default_rng(0x2048 + batch)
    x = rng.integers(0, P, size=n * batch, dtype=np.uint32)
    x[:n] = (7 * np.arange(n, dtype=np.uint64) + 3) % P
    ctx = ta.ntt.get_or_create_ctx(n)
    assert ctx.passes == 2 and ctx.passes_for(batch) == 1 and ctx.passes_for(5) == 2
    a = DevBuf(ta, x.nbytes)
    b = a if inplace else DevBuf(ta, x.nbytes)
    try:
        a.upload(x)
        ctx.run_device(a.ptr, b.ptr, batch, inverse, shift=shift)
        ctx.synchronize()
        y = b.download(np.uint32, x.size)
        if not inplace:
            assert (a.download(np.uint32, x.size) == x).all(), "out-of-place transform modified its input"
    finally:
        a.free()
        if b is not a:
            b.free()
    rows = x.reshape(batch, n).astype(np.uint64)
    for t in range(batch):
        want = oracle.domain_ifft(rows[t], shift) if inverse else oracle.domain_fft(rows[t], n, shift)
        assert (y[t * n:(t + 1) * n] == want).all(), f"2^11 x{batch} shift {shift} inverse {inverse}: transform {t}"
    assert _launched(ta, "ntt_row2048_kernel"), "the one-wave-per-transform kernel never ran"


@pytest.mark.parametrize("batch,shift,inverse,inplace", [
    (2048, 1, False, True),       # the smallest batch that takes the kernel (TOYNI_R4096_MIN_ROWS): 256 tiles of eight rows
    (2051, 7, False, False),      # ragged last tile (three rows of eight): forward coset, out of place
    (2500, 1234567, True, True),  # inverse coset
    (2300, 1, True, False),       # plain inverse
])
def test_n2p12_two_waves_per_transform(ta, batch, shift, inverse, inplace):
    """n = 2^12 in ONE sweep, two waves per transform, eight transforms per workgroup (Row4096, ntt_row4096_kernel).  EVERY transform of
    the batch against the oracle."""
    n = 1 << 12
    rng = np.random.default_rng(0x4096 + batch)
    x = rng.integers(0, P, size=n * batch, dtype=np.uint32)
    ctx = ta.ntt.get_or_create_ctx(n)
    assert ctx.passes == 2 and ctx.passes_for(batch) == 1 and ctx.passes_for(5) == 2
    a = DevBuf(ta, x.nbytes)
    b = a if inplace else DevBuf(ta, x.nbytes)
    try:
        a.upload(x)
        ctx.run_device(a.ptr, b.ptr, batch, inverse, shift=shift)
        ctx.synchronize()
        y = b.download(np.uint32, x.size)
        if not inplace:
            assert (a.download(np.uint32, x.size) == x).all(), "out-of-place transform modified its input"
    finally:
        a.free()
        if b is not a:
            b.free()
    rows = x.reshape(batch, n).astype(np.uint64)
    for t in range(batch):
        want = oracle.domain_ifft(rows[t], shift) if inverse else oracle.domain_fft(rows[t], n, shift)
        assert (y[t * n:(t + 1) * n] == want).all(), f"2^12 x{batch} shift {shift} inverse {inverse}: transform {t}"
    assert _launched(ta, "ntt_row4096_kernel"), "the two-waves-per-transform kernel never ran"
