"""A fixed matrix of device-resident calls against the oracle, built to walk the launcher's DISPATCH TABLE rather than the API:
for every transform size the batches that select each tile width (three-step latency shapes, 8 / 16 / 32-wide, 64-wide), every zero
fraction of the low-degree extension's first pass, the interleaved (Ext, AoS) forms of all of them, and one fold.

Used twice by tests/test_gpu_dispatch_matrix.py: in process with the default knobs, and as a child program (`python
tests/dispatch_matrix.py <profile>`) under the dispatch knobs that reach the variants no default call selects at test sizes
(TOYNI_NT_MIN_BYTES=0: the non-temporal twins, which a default call takes only from 512 MiB up; TOYNI_P3_TILES=-1: the two-step
shapes on small launches; TOYNI_WIDE_TILES=0: 64-wide tiles; the single-sweep kernel's other workgroup shapes).  Every child
dumps the kernels it launched into $TOYNI_LAUNCH_LOG at exit (tests/_hooks/sitecustomize.py), which is what
tests/test_zz_kernel_coverage.py judges.

Checks per case, bit-exact: sampled transforms of the batch (first, last, one in the middle) against the oracle's
domain_fft / ntt (src/math/domain.rs:107-123, src/ntt.rs:24-53), and the WHOLE batch through inverse(forward(x)) == x."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

P = 2013265921


class Dev:
    def __init__(self, ta, nbytes):
        self.lib = ta._lib.lib
        p = ctypes.c_void_p()
        ta._lib.check(self.lib.toyni_malloc(ctypes.byref(p), max(nbytes, 16)), "malloc")
        self.ptr = p.value

    def up(self, arr):
        arr = np.ascontiguousarray(arr)
        assert self.lib.toyni_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes) == 0

    def down(self, count):
        out = np.empty(count, dtype=np.uint32)
        assert self.lib.toyni_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes) == 0
        return out

    def free(self):
        self.lib.toyni_free(self.ptr)


def batches_for(log_n, cap_log_elems):
    """Batches that select the different tile widths of an n-point plan: a lone transform and three (latency shapes / narrow tiles),
    then enough transforms for 2^7, 2^9 and 2^12 32-wide tiles (the thresholds of ntt_plan.hpp) of the plan's smallest and of its
    largest pass."""
    out = [1, 3]
    if log_n >= 11:
        ms = {log_n // 2, (log_n + 1) // 2} if log_n <= 20 else {log_n // 3, (log_n + 2) // 3}
        for m in ms:
            for lt in (7, 9, 12):
                b = 1 << max(0, lt + 5 + m - log_n)
                if log_n + (b - 1).bit_length() <= cap_log_elems:
                    out.append(b)
    else:
        out += [70, 1 << max(0, min(14, cap_log_elems) - log_n)]
    return sorted(set(out))


# profile -> (sizes, batches beyond [1, 3]?, LDE sizes, Ext?, log2 cap on the elements of a case)
PROFILES = {
    "default": dict(sizes=range(1, 23), tiers=True, lde=(12, 14, 16, 18, 20, 21, 22), ext=True, cap=26),
    # (2^21 in the `nt` profile: the non-temporal twin of the streaming 2048-point closing pass, which a default call takes from 512 MiB up)
    "nt": dict(sizes=range(8, 22), tiers=True, lde=(), ext=True, cap=24),
    "two_step": dict(sizes=range(11, 21), tiers=False, lde=(12, 14, 16, 18, 20), ext=True, cap=23),
    "two_step_nt": dict(sizes=range(11, 21), tiers=False, lde=(), ext=True, cap=23),
    "wide": dict(sizes=range(13, 20), tiers=False, lde=(), ext=True, cap=23),
    "wide_nt": dict(sizes=range(13, 20), tiers=False, lde=(), ext=True, cap=23),
    "lds": dict(sizes=range(11, 16), tiers=True, lde=(), ext=False, cap=22),
}
# environment of the child program for each profile (tests/test_gpu_dispatch_matrix.py)
PROFILE_ENV = {
    "nt": {"TOYNI_NT_MIN_BYTES": "0", "TOYNI_FOLD_NT_MIN_BYTES": "0"},
    "two_step": {"TOYNI_P3_TILES": "-1"},
    "two_step_nt": {"TOYNI_P3_TILES": "-1", "TOYNI_NT_MIN_BYTES": "0"},
    "wide": {"TOYNI_P3_TILES": "-1", "TOYNI_WIDE_TILES": "0"},
    "wide_nt": {"TOYNI_P3_TILES": "-1", "TOYNI_WIDE_TILES": "0", "TOYNI_NT_MIN_BYTES": "0"},
    "lds": {"TOYNI_LDS_MIN_ELEMS": "0", "TOYNI_LDS_MAX_LOG": "15", "TOYNI_LDS_ROWS": "4"},
}


def run_matrix(ta, oracle, profile="default", log=print):
    cfg = PROFILES[profile]
    full = profile == "default"
    rng = np.random.default_rng(0xD15BA7C4)
    ncases = 0

    def samples(batch, log_n):
        if not full:
            return [batch - 1]
        return sorted({0, batch - 1} if log_n >= 19 else {0, batch // 2, batch - 1})

    def ctx_of(log_n):
        return ta.ntt.get_or_create_ctx(1 << log_n)

    def check_base(log_n, batch, shift):
        n = 1 << log_n
        ctx = ctx_of(log_n)
        x = rng.integers(0, P, size=n * batch, dtype=np.uint32)
        a, b = Dev(ta, x.nbytes), Dev(ta, x.nbytes)
        try:
            a.up(x)
            ctx.run_device(a.ptr, b.ptr, batch, False, shift=shift)
            ctx.synchronize()
            y = b.down(x.size)
            for t in samples(batch, log_n):
                want = oracle.domain_fft(x[t * n:(t + 1) * n].astype(np.uint64), n, shift)
                assert (y[t * n:(t + 1) * n] == want).all(), f"ntt 2^{log_n} x{batch} shift {shift}: transform {t} differs from the oracle"
            ctx.run_device(b.ptr, b.ptr, batch, True, shift=shift)
            ctx.synchronize()
            assert (b.down(x.size) == x).all(), f"ntt 2^{log_n} x{batch} shift {shift}: round trip"
        finally:
            a.free(); b.free()

    def coords_of(t, first):
        return range(4) if t == first else [t % 4]     # every coordinate of one vector, one coordinate of the other samples

    def check_lde(log_n, batch, z, shift, ext):
        n, q = 1 << log_n, 4 if ext else 1
        n_in = n >> z
        ctx = ctx_of(log_n)
        c = rng.integers(0, P, size=n_in * batch * q, dtype=np.uint32)
        a, b = Dev(ta, c.nbytes), Dev(ta, 4 * n * batch * q)
        try:
            a.up(c)
            if ext:
                ctx.lde_ext_device(a.ptr, b.ptr, batch, z, shift)
            else:
                ctx.lde_device(a.ptr, b.ptr, batch, z, shift)
            ctx.synchronize()
            y = b.down(n * batch * q).reshape(batch, n, q)
            cc = c.reshape(batch, n_in, q)
            ts = [batch - 1] if (ext or not full) else sorted({0, batch - 1})
            for t in ts:
                for k in (coords_of(t, ts[0]) if ext else [0]):
                    want = oracle.domain_fft(cc[t, :, k].astype(np.uint64), n, shift)
                    assert (y[t, :, k] == want).all(), f"lde{'_ext' if ext else ''} 2^{log_n} x{batch} blow-up 2^{z}: vector {t} coordinate {k}"
        finally:
            a.free(); b.free()

    def check_ext(log_n, batch, shift):
        n = 1 << log_n
        ctx = ctx_of(log_n)
        x = rng.integers(0, P, size=4 * n * batch, dtype=np.uint32)
        a, b = Dev(ta, x.nbytes), Dev(ta, x.nbytes)
        try:
            a.up(x)
            ctx.run_device_ext_batch(a.ptr, b.ptr, batch, False, shift=shift)
            ctx.synchronize()
            y = b.down(x.size).reshape(batch, n, 4)
            xx = x.reshape(batch, n, 4)
            ts = samples(batch, log_n)
            for t in ts:
                for k in coords_of(t, ts[0]):
                    want = oracle.domain_fft(xx[t, :, k].astype(np.uint64), n, shift)
                    assert (y[t, :, k] == want).all(), f"ext 2^{log_n} x{batch} shift {shift}: vector {t} coordinate {k}"
            ctx.run_device_ext_batch(b.ptr, b.ptr, batch, True, shift=shift)
            ctx.synchronize()
            assert (b.down(x.size) == x).all(), f"ext 2^{log_n} x{batch} shift {shift}: round trip"
        finally:
            a.free(); b.free()

    for log_n in cfg["sizes"]:
        bs = batches_for(log_n, cfg["cap"]) if cfg["tiers"] else [1, 3]
        if log_n >= 21:
            bs = [1, bs[-1]]
        for batch in bs:
            check_base(log_n, batch, 7 if (log_n + batch) & 1 else 1)
            ncases += 1
            if cfg["ext"] and (batch <= 3 or batch % 4 == 0):
                # the same number of tiles as `batch` base transforms: batch / 4 vectors (a lone vector is four transforms' worth)
                check_ext(log_n, batch if batch <= 3 else batch // 4, 1 if (log_n + batch) & 1 else 7)
                ncases += 1
        if log_n in cfg["lde"]:
            m1 = int(ta._lib.lib.toyni_ntt_ctx_first_pass_points(ctx_of(log_n).handle)).bit_length() - 1
            for z in range(1, min(5, m1) + 1):
                # 1: latency shapes (two-step 8-wide under TOYNI_P3_TILES=-1); 4 / 16: the 16- and 32-wide tiles of a 1024-point first pass
                # 2^21 / 2^22: a lone vector (latency shapes) and four (the two-pass plans' streaming shapes, 16-wide 2048-point passes)
                for batch in ([1, 4, 16] if (full and log_n == 20) else [1, 16] if full and log_n < 20 else [1, 4] if full else [1]):
                    check_lde(log_n, batch, z, 7, False)
                    ncases += 1
                if cfg["ext"] and log_n <= 22:   # (2^21 / 2^22: the interleaved streaming 2048-point shapes, a lone vector)
                    for vecs in ([1, 4] if full and log_n <= 20 else [1]):
                        check_lde(log_n, vecs, z, 7, True)
                        ncases += 1
            if log_n <= 16:   # a blow-up beyond five bits (the LZ = 5 variant with its row guard) and one beyond the first pass (pad + transform)
                check_lde(log_n, 2, min(m1, 6), 7, False)
                check_lde(log_n, 2, m1 + 1, 3, False)
                if cfg["ext"]:
                    check_lde(log_n, 2, min(m1, 6), 7, True)
                    check_lde(log_n, 1, m1 + 1, 3, True)
                ncases += 4
        log(f"  2^{log_n}: ok ({ncases} cases so far)")
    # one structured fold per launcher branch that a footprint knob can move (the non-temporal twin of the 256-thread kernel)
    for m in (1 << 12, 1 << 16):
        ctx = ctx_of(16)
        e = rng.integers(0, P, size=m, dtype=np.uint32)
        a, o = Dev(ta, e.nbytes), Dev(ta, e.nbytes // 2)
        try:
            a.up(e)
            ta.fri_fold_device(ctx, a.ptr, o.ptr, m, 424242, 7)
            ctx.synchronize()
            want = oracle.fri_fold(e.astype(np.uint64), oracle.domain_elements(m, 7), 424242)
            assert (o.down(m // 2) == want).all(), f"fold m={m}"
        finally:
            a.free(); o.free()
        ncases += 1
    # the explicit-point fold above its 16-per-inversion gate (2^18 pairs), with one quad in the last chunk: the plain kernel by default,
    # its non-temporal twin under the footprint knob (TOYNI_FOLD_NT_MIN_BYTES=0 in the `nt` profile)
    m = (1 << 19) + 8
    e = rng.integers(0, P, size=m, dtype=np.uint32)
    xs = rng.integers(1, P, size=m // 2, dtype=np.uint32)
    xs[[3, 3 + 1024, m // 2 - 1]] = 0
    a, x, o = Dev(ta, e.nbytes), Dev(ta, xs.nbytes), Dev(ta, e.nbytes // 2)
    try:
        a.up(e); x.up(xs)
        ta.fri_fold_xs_device(a.ptr, x.ptr, o.ptr, m, 424242)
        ta._lib.check(ta._lib.lib.toyni_stream_synchronize(None, None), "sync")
        x1 = np.where(xs == 0, 1, xs).astype(np.uint64)
        want = oracle.fri_fold(e.astype(np.uint64), x1, 424242)
        for i in np.flatnonzero(xs == 0):   # x^-1 := 0: only the average survives
            want[i] = oracle.bb_mul(oracle.bb_add(int(e[i]), int(e[i + m // 2])), (P + 1) // 2)
        assert (o.down(m // 2) == want).all(), "explicit-point fold"
    finally:
        a.free(); x.free(); o.free()
    ncases += 1
    return ncases


def main(argv):
    profile = argv[1] if len(argv) > 1 else "default"
    import __graft_entry__ as entry
    entry.build_hip()
    import oracle
    import toyni_amd
    assert toyni_amd.gpu_available(), "no GPU visible"
    n = run_matrix(toyni_amd, oracle, profile, log=lambda s: print(s, flush=True))
    print(f"MATRIX OK profile={profile} cases={n}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
