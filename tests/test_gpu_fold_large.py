"""GPU parity of the SIZE-GATED fold kernels (VERDICT r3 #1 / ADVICE r3): the launcher picks them by layer size, so the ordinary
fold tests (layers up to 2^21) never reach them.

  fri_fold_stream_kernel<true,1024,2>    structured points, m * 4 B >= TOYNI_FOLD_NT_MIN_BYTES (256 MiB)          -> m = 2^26, 2^27
  fri_fold_xs16_kernel<NT,256>           explicit points, half >= 2^18 (toyni_fri_fold_xs_device's gate), whole quads, 16-byte aligned
                                         pointers -> m = 2^19 + 8, 2^22 (+ a ragged last chunk), 2^24; the non-temporal twin
                                         from m * 4 B >= 256 MiB -> m = 2^26
  fri_fold_ext_stream_kernel<true,...>   Ext values, m * 16 B >= 256 MiB                                          -> m = 2^25

Every output of every layer is compared with the oracle's fri_fold / fri_fold_ext (src/math/fri.rs:27-48, :7-25) -- bit-exact.
The oracle folds ~2 * 10^7 elements per second on one thread, so the largest layer costs a few seconds of CPU."""
import ctypes

import numpy as np
import pytest

import oracle
from oracle import P
from test_gpu_parity import DevBuf, ta  # noqa: F401  (fixture)

gpu = pytest.mark.gpu


def _mulmod(a, b):
    return (a * b) % np.uint64(P)      # operands < 2^31: the product fits a u64


def coset_points(x0, log_m, count):
    """x0 * w_m^i for i < count (count a power of two), by doubling: block [2^k, 2^(k+1)) = block [0, 2^k) * w_m^(2^k)."""
    w = oracle.root_of_unity(log_m)
    out = np.empty(count, dtype=np.uint64)
    out[0] = x0
    step, filled = w, 1
    while filled < count:
        out[filled:2 * filled] = _mulmod(out[:filled], np.uint64(step))
        step = oracle.bb_mul(step, step)
        filled *= 2
    return out


def test_point_helper_is_the_oracles_chain():
    # w_m^(i * 2^11) = w_{m / 2^11}^i: the strided helper output is the oracle's domain of the smaller size
    xs = coset_points(49, 27, 1 << 20)
    assert (xs[:: 1 << 11] == oracle.domain_elements(1 << 16, 49)[: 1 << 9]).all()
    assert (coset_points(7, 10, 1 << 9) == oracle.domain_elements(1 << 10, 7)[: 1 << 9]).all()


def _launched(ta, needle):
    return [k for k in ta._lib.launched_kernels() if needle in k]


@gpu
@pytest.mark.parametrize("log_m,layer", [(26, 1), (27, 0)])
def test_structured_fold_stream_kernels_vs_oracle(ta, log_m, layer):
    # layer `layer` of a 2^27 codeword on the coset 7 * <w>: points (7 w^i)^(2^layer) = 7^(2^layer) * w_m^i
    N, m = 1 << 27, 1 << log_m
    assert m == N >> layer
    ctx = ta.ntt.get_or_create_ctx(N)
    x0 = oracle.bb_pow(7, 1 << layer)
    beta = 1234567891
    rng = np.random.default_rng(100 + log_m)
    e32 = rng.integers(0, P, size=m, dtype=np.uint32)
    a, o = DevBuf(ta, m * 4), DevBuf(ta, m * 2)
    try:
        a.upload(e32)
        ta.fri_fold_device(ctx, a.ptr, o.ptr, m, beta, x0)
        ctx.synchronize()
        got = o.download(np.uint32, m // 2)
        assert (a.download(np.uint32, 1 << 16) == e32[: 1 << 16]).all()      # the input layer is untouched
    finally:
        a.free(); o.free()
    want = oracle.fri_fold(e32.astype(np.uint64), coset_points(x0, log_m, m // 2), beta)
    assert (got == want).all(), np.flatnonzero(got != want)[:8]
    assert _launched(ta, "fri_fold_stream_kernelILb1ELi1024ELi2EE"), "the launcher did not take the shaped-stream kernel this test is about"


@gpu
@pytest.mark.parametrize("m", [1 << 19, (1 << 19) + 8, 1 << 22, (1 << 22) + 8 * 1000, 1 << 24, 1 << 26])
def test_explicit_point_fold_xs16_vs_oracle(ta, m):
    # (2^19: the smallest layer the launcher gives to this kernel; 2^19 + 8: one quad in the last chunk; 2^22 + 8000: half is a
    # multiple of 4 but not of the 4096 pairs a workgroup covers per iteration -- a ragged last chunk)
    lib = ta._lib.lib
    half = m // 2
    rng = np.random.default_rng(200 + m % 1000 + m.bit_length())
    e = rng.integers(0, P, size=m, dtype=np.uint64)
    xs = rng.integers(1, P, size=half, dtype=np.uint64)            # arbitrary nonzero points, not a coset
    # zero points: a thread's 16 points are quad tid of each of the four 256-quad runs of its 1024-quad chunk, so 5 / 5 + 1024 /
    # 5 + 3072 share an inversion with one another; first, middle and last chunk
    zeros = [5, 5 + 1024, 5 + 3072, 4096 * 50 + 4095, half - 16, half - 1]
    xs[zeros] = 0
    beta = 987654321
    want = oracle.fri_fold(e, np.where(xs == 0, np.uint64(1), xs), beta)
    for i in zeros:                                              # x^-1 := 0 = pow(0, p - 2): only the average survives
        want[i] = oracle.bb_mul(oracle.bb_add(int(e[i]), int(e[i + half])), (P + 1) // 2)
    # one allocation with slack so that the same data can be handed over 4-byte MISaligned as well
    pad = 4
    de, dx, do = DevBuf(ta, m * 4 + 16), DevBuf(ta, half * 4 + 16), DevBuf(ta, half * 4 + 16)
    try:
        de.upload(e.astype(np.uint32)); dx.upload(xs.astype(np.uint32))
        ta._lib.check(lib.toyni_fri_fold_xs_device(de.ptr, dx.ptr, do.ptr, m, beta, None), "fold")
        assert lib.toyni_stream_synchronize(None, None) == 0      # the call ran on the null stream
        got = do.download(np.uint32, half)
        assert (got == want).all(), np.flatnonzero(got != want)[:8]
        twin = "fri_fold_xs16_kernelILb1ELi256EE" if m * 4 >= (256 << 20) else "fri_fold_xs16_kernelILb0ELi256EE"
        assert _launched(ta, twin), "the launcher did not take the 16-per-inversion kernel this size is about"
        # the same layer behind pointers that are only 4-byte aligned: must fall back to the 4-per-inversion kernel and agree
        de.upload(e.astype(np.uint32), offset=pad); dx.upload(xs.astype(np.uint32), offset=pad)
        ta._lib.check(lib.toyni_fri_fold_xs_device(de.ptr + pad, dx.ptr + pad, do.ptr + pad, m, beta, None), "fold")
        assert lib.toyni_stream_synchronize(None, None) == 0
        got = do.download(np.uint32, half, offset=pad)
        assert (got == want).all(), np.flatnonzero(got != want)[:8]
        assert _launched(ta, "fri_fold_xs_kernel")
    finally:
        de.free(); dx.free(); do.free()


@gpu
def test_ext_fold_stream_kernel_vs_oracle(ta):
    # 2^25 Ext elements = 512 MiB of input: the shaped-stream Ext kernel; layer 2 of a 2^27 context, points 7^4 * w_m^i
    log_m, N = 25, 1 << 27
    m = 1 << log_m
    ctx = ta.ntt.get_or_create_ctx(N)
    x0 = oracle.bb_pow(7, 4)
    rng = np.random.default_rng(325)
    beta = [int(v) for v in rng.integers(0, P, size=4)]
    e32 = rng.integers(0, P, size=(m, 4), dtype=np.uint32)
    a, o = DevBuf(ta, m * 16), DevBuf(ta, m * 8)
    try:
        a.upload(e32)
        ta.fri_fold_ext_device(ctx, a.ptr, o.ptr, m, beta, x0)
        ctx.synchronize()
        got = o.download(np.uint32, (m // 2) * 4).reshape(-1, 4)
    finally:
        a.free(); o.free()
    xs = coset_points(x0, log_m, m // 2)
    # the oracle in slices (keeps the u64 copies small): pair (i, i + m/2) of the layer = pair (j, j + L) of a 2L-element slice
    L = 1 << 21
    for s0 in range(0, m // 2, L):
        part = np.concatenate([e32[s0:s0 + L], e32[m // 2 + s0:m // 2 + s0 + L]]).astype(np.uint64)
        want = oracle.fri_fold_ext(part, xs[s0:s0 + L], beta)
        assert (got[s0:s0 + L] == want).all(), (s0, np.flatnonzero((got[s0:s0 + L] != want).any(axis=1))[:8])
    assert _launched(ta, "fri_fold_ext_stream_kernel")
