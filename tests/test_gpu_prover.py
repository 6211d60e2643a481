"""GPU parity of the pointwise prover kernels and the fused FRI round (include/toyni_hip.h 3c) against the oracle's restatements
of src/fibonacci.rs:133-150,186-198,222-245, src/math/polynomial.rs:134-144 and src/merkle.rs:50-80 -- bit-exact."""
import numpy as np
import pytest

import oracle
from oracle import P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as entry
    entry.build_hip()
    import torch
    assert torch.cuda.is_available()
    torch.cuda.init()
    import toyni_amd
    return toyni_amd, torch, torch.device("cuda", 0)


def _dev(torch, dev, arr_u64):
    return torch.from_numpy(np.asarray(arr_u64, dtype=np.uint64).astype(np.uint32).view(np.int32)).to(dev)


def _host(t):
    return t.cpu().numpy().view(np.uint32).astype(np.uint64)


@pytest.mark.parametrize("log_N,log_blowup,shift", [(6, 2, 7), (8, 5, 7), (10, 3, 1234567), (16, 5, 7), (21, 5, 7), (3, 1, 7), (5, 0, 7)])
def test_quotient_and_deep(env, log_N, log_blowup, shift):
    ta, torch, dev = env
    N, n = 1 << log_N, (1 << log_N) >> log_blowup
    if n < 2:
        pytest.skip("trace of one row")
    ctx = ta.ntt.get_or_create_ctx(N)
    lde = oracle.splitmix(N, 9000 + log_N)
    c_want, q_want = oracle.fib_quotient(lde, n, shift)
    t = _dev(torch, dev, lde)
    c = torch.empty_like(t)
    q = torch.empty_like(t)
    ta.prover.fib_quotient_device(ctx, t.data_ptr(), c.data_ptr(), q.data_ptr(), log_blowup, shift)
    torch.cuda.synchronize()
    assert (_host(c) == c_want).all() and (_host(q) == q_want).all()
    q2 = torch.empty_like(t)
    ta.prover.fib_quotient_device(ctx, t.data_ptr(), 0, q2.data_ptr(), log_blowup, shift)       # c_evals optional
    z, ood = 987654321, [5, P - 1, 0, 123456789]
    d_want = oracle.fib_deep(lde, q_want, n, shift, z, *ood)
    d = torch.empty_like(t)
    ta.prover.fib_deep_device(ctx, t.data_ptr(), q2.data_ptr(), d.data_ptr(), log_blowup, shift, z, ood)
    torch.cuda.synchronize()
    assert (_host(d) == d_want).all()


def test_quotient_rejects_a_coset_on_which_z_h_vanishes(env):
    ta, torch, dev = env
    ctx = ta.ntt.get_or_create_ctx(64)
    t = torch.zeros(64, dtype=torch.int32, device=dev)
    with pytest.raises(ta._lib.ToyniError, match="Cannot invert zero"):                            # shift = 1: x^n - 1 = 0 on the trace domain
        ta.prover.fib_quotient_device(ctx, t.data_ptr(), 0, t.data_ptr(), 3, 1)


@pytest.mark.parametrize("ncoeffs", [0, 1, 15, 16, 17, 4095, 4096, 4097, 65536 + 140, 1 << 21])
def test_poly_eval(env, ncoeffs):
    ta, torch, dev = env
    ctx = ta.ntt.get_or_create_ctx(1 << 10)
    c = oracle.splitmix(max(ncoeffs, 1), 31 + ncoeffs)[:ncoeffs]
    t = _dev(torch, dev, c if ncoeffs else np.zeros(1, dtype=np.uint64))
    pts = [123456789, 0, 1, P - 1]
    out = torch.zeros(4, dtype=torch.int32, device=dev)
    ta.prover.poly_eval_device(ctx, t.data_ptr(), ncoeffs, pts, out.data_ptr())
    torch.cuda.synchronize()
    assert _host(out).tolist() == [oracle.poly_eval(c, x) for x in pts]
    out1 = torch.zeros(1, dtype=torch.int32, device=dev)
    ta.prover.poly_eval_device(ctx, t.data_ptr(), ncoeffs, pts[:1], out1.data_ptr())
    torch.cuda.synchronize()
    assert int(_host(out1)[0]) == oracle.poly_eval(c, pts[0])


@pytest.mark.parametrize("n,salted", [(1, True), (2, False), (3, True), (13, True), (1000, False), (1 << 12, True), (1 << 16, True)])
def test_merkle_openings(env, n, salted):
    ta, torch, dev = env
    lib = ta._lib.lib
    vals = oracle.splitmix(n, 77 + n)
    salts = np.random.default_rng(n).integers(0, 256, (n, 16), dtype=np.uint8) if salted else None
    levels_want = oracle.merkle_commit_values(vals, salts)
    v = _dev(torch, dev, vals)
    s = torch.from_numpy(salts).to(dev) if salted else None
    lv = torch.empty((lib.toyni_merkle_total_digests(n), 32), dtype=torch.uint8, device=dev)
    ta.merkle_commit_device(v.data_ptr(), s.data_ptr() if salted else 0, n, lv.data_ptr())
    idx = sorted(set([0, n - 1, n // 2, (n * 7) // 11] + list(np.random.default_rng(1).integers(0, n, 10))))
    it = torch.tensor(idx, dtype=torch.int32, device=dev)
    rec = ta.prover.merkle_open_record_bytes(n)
    out = torch.zeros(len(idx) * rec, dtype=torch.uint8, device=dev)
    ta.prover.merkle_open_device(lv.data_ptr(), n, v.data_ptr(), s.data_ptr() if salted else 0, it.data_ptr(), len(idx), out.data_ptr())
    torch.cuda.synchronize()
    ops = ta.prover.parse_openings(out.cpu().numpy(), n, idx, salted)
    for o in ops:
        path, pos = oracle.merkle_get_proof(levels_want, o["index"])
        assert o["path"] == path and o["position"] == pos and o["value"] == int(vals[o["index"]])
        assert o["salt"] == (salts[o["index"]].tobytes() if salted else b"")


def test_merkle_openings_of_several_trees_in_one_launch(env):
    """toyni_merkle_open_groups_device (round 3): the query phase opens ~19 trees per proof; one launch for all of them must write
    exactly the records that one toyni_merkle_open_device call per tree writes -- 40 trees here (two launches of <= 32), sizes 1 ...
    2^12, salted and not, an empty group in the middle."""
    import ctypes
    ta, torch, dev = env
    lib = ta._lib.lib

    class Group(ctypes.Structure):
        _fields_ = [("d_levels", ctypes.c_void_p), ("n", ctypes.c_size_t), ("d_values", ctypes.c_void_p), ("d_salts", ctypes.c_void_p),
                    ("d_indices", ctypes.c_void_p), ("nidx", ctypes.c_size_t), ("d_out", ctypes.c_void_p)]

    rng = np.random.default_rng(33)
    keep, groups, want = [], [], []
    for k in range(40):
        n = int(rng.integers(1, 1 << 12)) if k % 7 else 1 << (k % 13)
        salted = bool(k % 3)
        vals = oracle.splitmix(n, 900 + k)
        v = _dev(torch, dev, vals)
        s = torch.from_numpy(rng.integers(0, 256, (n, 16), dtype=np.uint8)).to(dev) if salted else None
        lv = torch.empty((lib.toyni_merkle_total_digests(n), 32), dtype=torch.uint8, device=dev)
        ta.merkle_commit_device(v.data_ptr(), s.data_ptr() if salted else 0, n, lv.data_ptr())
        nidx = 0 if k == 17 else int(rng.integers(1, 9))
        it = torch.from_numpy(rng.integers(0, n, max(nidx, 1)).astype(np.int32)).to(dev)
        rec = ta.prover.merkle_open_record_bytes(n)
        one = torch.zeros(max(nidx, 1) * rec, dtype=torch.uint8, device=dev)
        if nidx:
            ta.prover.merkle_open_device(lv.data_ptr(), n, v.data_ptr(), s.data_ptr() if salted else 0, it.data_ptr(), nidx, one.data_ptr())
        out = torch.full((max(nidx, 1) * rec,), 0, dtype=torch.uint8, device=dev)
        keep += [v, s, lv, it, out]
        want.append((one, out, nidx * rec))
        groups.append(Group(lv.data_ptr(), n, v.data_ptr(), s.data_ptr() if salted else None, it.data_ptr(), nidx, out.data_ptr()))
    arr = (Group * len(groups))(*groups)
    assert lib.toyni_merkle_open_groups_device(arr, len(groups), None) == 0
    torch.cuda.synchronize()
    for k, (one, out, nbytes) in enumerate(want):
        assert torch.equal(one[:nbytes], out[:nbytes]), f"group {k}"
    arr[3].d_out = None
    assert lib.toyni_merkle_open_groups_device(arr, len(groups), None) == 10002          # a null pointer in any group: nothing is launched


@pytest.mark.parametrize("log_m,salted", [(1, False), (2, True), (5, True), (11, True), (12, False), (16, True), (21, True)])
def test_fold_commit_round(env, log_m, salted):
    """One protocol round (src/fibonacci.rs:222-245): layer values as fri_fold, tree as build_merkle_tree of the folded layer."""
    ta, torch, dev = env
    lib = ta._lib.lib
    m = 1 << log_m
    ctx = ta.ntt.get_or_create_ctx(max(m, 2))
    e = oracle.splitmix(m, 500 + log_m)
    beta, x0 = 424242 + log_m, 49
    xs = oracle.domain_elements(m, x0)
    want = oracle.fri_fold(e, xs, beta)
    half = m // 2
    salts = np.random.default_rng(log_m).integers(0, 256, (half, 16), dtype=np.uint8) if salted else None
    levels_want = np.concatenate(oracle.merkle_commit_values(want, salts))
    te = _dev(torch, dev, e)
    out = torch.empty(half, dtype=torch.int32, device=dev)
    s = torch.from_numpy(salts).to(dev) if salted else None
    lv = torch.zeros((lib.toyni_merkle_total_digests(half), 32), dtype=torch.uint8, device=dev)
    ta.prover.fri_fold_commit_device(ctx, te.data_ptr(), out.data_ptr(), m, beta, x0, s.data_ptr() if salted else 0, lv.data_ptr())
    torch.cuda.synchronize()
    assert (_host(out) == want).all()
    assert (lv.cpu().numpy() == levels_want).all()
    # and identical to the two separate calls it replaces
    out2 = torch.empty_like(out)
    lv2 = torch.zeros_like(lv)
    ta.fri_fold_device(ctx, te.data_ptr(), out2.data_ptr(), m, beta, x0)
    ta.merkle_commit_device(out2.data_ptr(), s.data_ptr() if salted else 0, half, lv2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(lv, lv2)


@pytest.mark.parametrize("log_m0,final_size,salted", [(6, 4, True), (10, 1, False), (14, 16, True), (21, 16, True)])
def test_commit_phase_in_one_call(env, log_m0, final_size, salted):
    """toyni_fri_commit_phase_device = the fold loop of src/fibonacci.rs:222-245: every layer as the oracle's fri_fold on the squared
    domain, every tree as build_merkle_tree (the final layer unsalted, :236-240), betas from a transcript that absorbed the previous
    root (here: SHA-256 chaining as src/transcript.rs:29-40), and the callback sees exactly the committed roots in order."""
    import hashlib
    ta, torch, dev = env
    lib = ta._lib.lib
    m0 = 1 << log_m0
    ctx = ta.ntt.get_or_create_ctx(m0)
    e = oracle.splitmix(m0, 7000 + log_m0)
    x0 = 7
    sizes = []
    m = m0
    while m > final_size:
        m //= 2
        sizes.append(m)
    rng = np.random.default_rng(log_m0)
    salts = [rng.integers(0, 256, (h, 16), dtype=np.uint8) for h in sizes[:-1]] if salted else None
    state = {"t": b"toyni-test", "seen": []}

    def challenge(rnd, root, want_beta):
        if root is not None:
            state["seen"].append(root)
            state["t"] = state["t"] + root                       # absorb_commitment
        if not want_beta:
            return 0
        h = hashlib.sha256(state["t"]).digest()                  # squeeze_challenge: hash, feed back, reduce
        state["t"] = h
        return int.from_bytes(h, "little") % P

    te = _dev(torch, dev, e)
    layers = torch.empty(sum(sizes), dtype=torch.int32, device=dev)
    total = sum(lib.toyni_merkle_total_digests(h) for h in sizes)
    levels = torch.zeros((total, 32), dtype=torch.uint8, device=dev)
    d_salts = torch.from_numpy(np.concatenate(salts)).to(dev) if salted and salts else None
    roots = ta.prover.fri_commit_phase_device(ctx, te.data_ptr(), m0, x0, final_size, d_salts.data_ptr() if d_salts is not None else 0,
                                              challenge, layers.data_ptr(), levels.data_ptr())
    assert len(roots) == len(sizes) and state["seen"] == roots
    # replay on the CPU with the same transcript
    got_layers, got_levels = _host(layers), levels.cpu().numpy()
    t = b"toyni-test"
    cur, x, lo, dlo = e, x0, 0, 0
    for k, h in enumerate(sizes):
        if k:
            t = t + roots[k - 1]
        t = hashlib.sha256(t).digest()
        beta = int.from_bytes(t, "little") % P
        want = oracle.fri_fold(cur, oracle.domain_elements(2 * h, x), beta)
        assert (got_layers[lo:lo + h] == want).all(), k
        folded = got_layers[lo:lo + h]
        s = salts[k] if salted and k < len(sizes) - 1 else None
        tree = np.concatenate(oracle.merkle_commit_values(folded, s))
        nd = lib.toyni_merkle_total_digests(h)
        assert (got_levels[dlo:dlo + nd] == tree).all(), k
        assert got_levels[dlo + nd - 1].tobytes() == roots[k]
        cur, x, lo, dlo = folded, x * x % P, lo + h, dlo + nd


def test_commit_phase_reports_a_failing_transcript(env):
    ta, torch, dev = env
    ctx = ta.ntt.get_or_create_ctx(64)
    te = _dev(torch, dev, oracle.splitmix(64, 1))
    layers = torch.empty(63, dtype=torch.int32, device=dev)
    levels = torch.zeros((200, 32), dtype=torch.uint8, device=dev)

    def bad(rnd, root, want_beta):
        if rnd == 2:
            raise RuntimeError("transcript failure")
        return 5

    with pytest.raises(RuntimeError, match="transcript failure"):
        ta.prover.fri_commit_phase_device(ctx, te.data_ptr(), 64, 7, 1, 0, bad, layers.data_ptr(), levels.data_ptr())
    # a beta outside the field is refused, not reduced silently
    with pytest.raises(Exception):
        ta.prover.fri_commit_phase_device(ctx, te.data_ptr(), 64, 7, 1, 0, lambda r, root, w: P, layers.data_ptr(), levels.data_ptr())


def test_commit_phase_refuses_reentry_also_after_a_nested_phase_on_another_context(env):
    """A transcript callback runs with its context locked: an entry point on THAT context from inside it returns TOYNI_E_REENTRANT
    instead of deadlocking -- also after the callback ran a whole commit phase on ANOTHER context (whose own callbacks come and go in
    between; ADVICE r3: the guard used to be one thread-local slot, which the nested phase cleared)."""
    ta, torch, dev = env
    lib = ta._lib.lib
    a, b = ta.NttContext(64, device=dev.index), ta.NttContext(32, device=dev.index)
    ea, eb = _dev(torch, dev, oracle.splitmix(64, 2)), _dev(torch, dev, oracle.splitmix(32, 3))
    la, lb = torch.empty(63, dtype=torch.int32, device=dev), torch.empty(31, dtype=torch.int32, device=dev)
    va, vb = torch.zeros((200, 32), dtype=torch.uint8, device=dev), torch.zeros((100, 32), dtype=torch.uint8, device=dev)
    scratch = torch.zeros(64, dtype=torch.int32, device=dev)
    seen = {"inner_rounds": 0, "status": []}

    def try_a():
        return lib.toyni_ntt_device(a.handle, scratch.data_ptr(), scratch.data_ptr(), 1, 0, None)

    def inner(rnd, root, want_beta):
        seen["inner_rounds"] += 1
        seen["status"].append(("inside B's callback, on A", try_a()))     # A's callback is still further up this thread's stack
        return 9

    def outer(rnd, root, want_beta):
        if rnd == 1:
            seen["status"].append(("before the nested phase", try_a()))
            ta.prover.fri_commit_phase_device(b, eb.data_ptr(), 32, 7, 1, 0, inner, lb.data_ptr(), vb.data_ptr())
            seen["status"].append(("after the nested phase", try_a()))
            # B itself is free again once its phase has returned
            assert lib.toyni_ntt_device(b.handle, scratch.data_ptr(), scratch.data_ptr(), 1, 0, None) == 0
        return 5

    roots = ta.prover.fri_commit_phase_device(a, ea.data_ptr(), 64, 7, 1, 0, outer, la.data_ptr(), va.data_ptr())
    assert len(roots) == 6 and seen["inner_rounds"] == 6     # 5 folds + the closing absorb, on B
    reentrant = 10011
    assert ta._lib.error_string(reentrant).startswith("entry point called on a context from inside")
    assert seen["status"] and all(st == reentrant for _, st in seen["status"]), seen["status"]
    assert try_a() == 0                                       # and A is free once its own phase has returned
    torch.cuda.synchronize()
    a.destroy(); b.destroy()
