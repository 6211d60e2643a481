"""Pins oracle/toyni_oracle.c (the CPU restatement) against
 (a) every known-answer test the reference holds for this path, restated from
     /root/reference/src/{babybear,ntt}.rs and src/math/domain.rs test modules, and
 (b) the independent big-int golden vectors in tests/golden/vectors.json.
CPU only."""
import numpy as np
import pytest

import oracle
from oracle import P


# ---- src/babybear.rs:219-284 ----
def test_basic_arithmetic():
    assert oracle.bb_add(100, 200) == 300
    assert oracle.bb_sub(200, 100) == 100
    assert oracle.bb_mul(100, 200) == 20000


def test_modular_reduction():
    assert oracle.bb_new(P + 5) == 5


def test_inverse():
    assert oracle.bb_mul(7, oracle.bb_inverse(7)) == 1


def test_pow():
    assert oracle.bb_pow(3, 4) == 81


def test_root_of_unity():
    for log_n in range(1, 11):
        assert oracle.bb_pow(oracle.root_of_unity(log_n), 1 << log_n) == 1


def test_negation():
    assert oracle.bb_add(100, oracle.bb_neg(100)) == 0
    assert oracle.bb_neg(0) == 0


def test_division():
    q = oracle.bb_div(100, 7)
    assert oracle.bb_mul(q, 7) == 100


def test_field_golden(golden):
    g = golden["field"]
    assert oracle.bb_inverse(7) == g["inv_7"]
    assert oracle.bb_neg(100) == g["neg_100"]
    assert oracle.bb_div(100, 7) == g["div_100_7"]
    assert oracle.bb_inverse(2) == g["half_inv"]
    assert oracle.bb_mul(P - 1, P - 1) == g["mul_pm1_pm1"]
    assert [oracle.root_of_unity(k) for k in range(28)] == golden["roots_of_unity"]


def test_sub_wraps_and_add_reduces():
    assert oracle.bb_sub(0, 1) == P - 1
    assert oracle.bb_add(P - 1, P - 1) == P - 2
    rng = np.random.default_rng(1)
    for a, b in rng.integers(0, P, size=(200, 2)):
        a, b = int(a), int(b)
        assert oracle.bb_mul(a, b) == a * b % P
        assert oracle.bb_add(a, b) == (a + b) % P
        assert oracle.bb_sub(a, b) == (a - b) % P


# ---- src/ntt.rs:321-379 ----
def test_ntt_intt_roundtrip():
    n = 256
    omega = oracle.root_of_unity(8)
    values = oracle.pattern_7i3(n)
    out = oracle.intt(oracle.ntt(values, omega), omega)
    assert (out == values).all()


def test_polynomial_evaluation():
    n = 8
    omega = oracle.root_of_unity(3)
    domain = oracle.roots_of_unity_domain(n)
    coeffs = [1, 2, 3, 0, 0, 0, 0, 0]
    evals = oracle.ntt(coeffs, omega)
    assert evals[0] == 6
    x = int(domain[1])
    assert evals[1] == (1 + 2 * x + 3 * x * x) % P


def test_roots_of_unity():
    n = 16
    domain = oracle.roots_of_unity_domain(n)
    assert domain[0] == 1
    assert oracle.bb_pow(int(domain[1]), n) == 1
    assert len(set(domain.tolist())) == n


def test_ntt_golden(golden):
    for c in golden["ntt"]:
        fwd = oracle.ntt(c["input"], c["omega"])
        assert fwd.tolist() == c["forward"], c["name"]
        assert oracle.ntt(c["input"]).tolist() == c["forward"], c["name"]  # canonical root
        inv = oracle.intt(c["input"], c["omega"])
        assert inv.tolist() == c["inverse"], c["name"]


def test_ntt_rejects_non_pow2():
    with pytest.raises(AssertionError):
        oracle.ntt([1, 2, 3])


# ---- src/math/domain.rs:192-242 ----
def test_fft_ifft_roundtrip():
    coeffs = [(i * 3 + 1) % P for i in range(8)]
    assert oracle.domain_ifft(oracle.domain_fft(coeffs, 8)).tolist() == coeffs


def test_coset_fft_ifft_roundtrip():
    coeffs = [(i * 3 + 1) % P for i in range(8)]
    assert oracle.domain_ifft(oracle.domain_fft(coeffs, 8, 7), 7).tolist() == coeffs


def test_coset_golden(golden):
    for c in golden["coset"]:
        assert oracle.domain_elements(c["size"], c["shift"]).tolist() == c["points"], c["name"]
        evals = oracle.domain_fft(c["coeffs"], c["size"], c["shift"])
        assert evals.tolist() == c["evals"], c["name"]
        back = oracle.domain_ifft(evals, c["shift"]).tolist()
        assert back[: len(c["coeffs"])] == c["coeffs"] and not any(back[len(c["coeffs"]):])


# ---- src/math/fri.rs:27-48 ----
def test_fold_golden(golden):
    for c in golden["fold"]:
        assert oracle.fri_fold(c["evals"], c["xs"], c["beta"]).tolist() == c["folded"], c["name"]
        # only xs[0..half) is read (src/math/fri.rs:36)
        assert oracle.fri_fold(c["evals"], c["xs"][: c["n"] // 2], c["beta"]).tolist() == c["folded"]


def test_fold_layers_golden(golden):
    c = golden["fold_layers"]
    layers = oracle.fri_fold_layers(c["evals"], c["shift"], c["betas"])
    assert [l.tolist() for l in layers] == c["layers"]
    assert len(set(layers[-1].tolist())) == 1  # src/verifier.rs:69-75


def test_fold_matches_verifier_restatement():
    # src/verifier.rs:177-181: avg + diff * beta * x0.inverse()
    n = 64
    evals = oracle.splitmix(n, 5)
    xs = oracle.domain_elements(n, 7)
    beta = 987654321
    out = oracle.fri_fold(evals, xs, beta)
    half_inv = pow(2, P - 2, P)
    for i in range(n // 2):
        a, b, x = int(evals[i]), int(evals[i + n // 2]), int(xs[i])
        exp = ((a + b) * half_inv + (a - b) * half_inv * beta * pow(x, P - 2, P)) % P
        assert out[i] == exp


def test_fold_rejects_odd():
    with pytest.raises(AssertionError):
        oracle.fri_fold([1, 2, 3], [1, 2, 3], 5)


def test_fold_ext_embeds_base():
    # a base-field codeword embedded in Ext with a base beta folds to the embedded base fold
    n = 32
    evals = oracle.splitmix(n, 9)
    xs = oracle.domain_elements(n, 7)
    beta = 424242
    base = oracle.fri_fold(evals, xs, beta)
    e4 = np.zeros((n, 4), dtype=np.uint64)
    e4[:, 0] = evals
    out = oracle.fri_fold_ext(e4, xs, [beta, 0, 0, 0])
    assert (out[:, 0] == base).all() and not out[:, 1:].any()


def test_fold_ext_is_coordinatewise_linear_with_x4_eq_11():
    # beta = X: (d0 + d1 X + d2 X^2 + d3 X^3) * X = 11 d3 + d0 X + d1 X^2 + d2 X^3  (src/ext.rs:178-192)
    n = 8
    rng = np.random.default_rng(3)
    e4 = rng.integers(0, P, size=(n, 4)).astype(np.uint64)
    xs = oracle.domain_elements(n, 7)
    out = oracle.fri_fold_ext(e4, xs, [0, 1, 0, 0])
    hi = pow(2, P - 2, P)
    for i in range(n // 2):
        a, b = [int(v) for v in e4[i]], [int(v) for v in e4[i + n // 2]]
        xinv = pow(int(xs[i]), P - 2, P)
        avg = [(x + y) * hi % P for x, y in zip(a, b)]
        d = [(x - y) * hi % P for x, y in zip(a, b)]
        dx = [11 * d[3] % P, d[0], d[1], d[2]]
        assert out[i].tolist() == [(avg[k] + dx[k] * xinv) % P for k in range(4)]


def test_fold_ext_golden(golden):
    # independent schoolbook model of Ext = F_p[X]/(X^4 - 11) (tests/golden/gen_golden.py)
    assert golden["ext"]["x4"] == [11, 0, 0, 0]
    for c in golden["ext"]["fold"]:
        got = oracle.fri_fold_ext(np.array(c["evals"], dtype=np.uint64), c["xs"], c["beta"])
        assert got.tolist() == c["folded"], c["name"]


def test_merkle_golden(golden):
    for c in golden["merkle"]:
        salts = np.frombuffer(b"".join(bytes.fromhex(s) for s in c["salts_hex"]), dtype=np.uint8).reshape(c["n"], 16)
        assert oracle.merkle_commit_values(c["values"], None)[-1][0].tobytes().hex() == c["root_unsalted"]
        assert oracle.merkle_commit_values(c["values"], salts)[-1][0].tobytes().hex() == c["root_salted"]


# ---- pointwise prover steps (round 2): pinned on an independent model that follows the reference LITERALLY --------------
# src/fibonacci.rs:133-150,186-198 evaluate the trace polynomial by Horner at x, g x and g^2 x for every coset point; the
# oracle's restatement reads the LDE at positions i, i + B, i + 2B instead.  The model below does it the reference's way
# (python ints, Polynomial::evaluate per point), so the identity the restatement relies on is itself under test.
def _horner(coeffs, x):
    r = 0
    for c in reversed(coeffs):
        r = (r * x + c) % P
    return r


def test_fib_quotient_and_deep_match_the_reference_formulas():
    import random
    rnd = random.Random(5)
    for n, blow in ((8, 4), (16, 32), (4, 8)):
        N = n * blow
        shift = 7
        g = oracle.root_of_unity(n.bit_length() - 1)
        w = oracle.root_of_unity(N.bit_length() - 1)
        xs = [shift * pow(w, i, P) % P for i in range(N)]
        tpoly = [rnd.randrange(P) for _ in range(n + 3)]                       # a masked trace polynomial: degree >= n
        lde = [_horner(tpoly, x) for x in xs]
        b1, b2 = pow(g, n - 1, P), pow(g, n - 2, P)
        c_ref, q_ref = [], []
        for x in xs:
            fib = (_horner(tpoly, g * g * x % P) - (_horner(tpoly, g * x % P) + _horner(tpoly, x))) % P
            c = fib * ((x - b1) % P) % P * ((x - b2) % P) % P
            c_ref.append(c)
            q_ref.append(c * pow((pow(x, n, P) - 1) % P, P - 2, P) % P)
        c, q = oracle.fib_quotient(lde, n, shift)
        assert c.tolist() == c_ref and q.tolist() == q_ref, (n, blow)
        z = rnd.randrange(P)
        t_z, t_gz, t_ggz, q_z = (rnd.randrange(P) for _ in range(4))            # any values: the formula is checked, not the protocol
        d_ref = []
        for i, x in enumerate(xs):
            inv = pow((x - z) % P, P - 2, P)
            d_ref.append(((q_ref[i] - q_z) * inv + (_horner(tpoly, g * g * x % P) - t_ggz) * inv
                          + (_horner(tpoly, g * x % P) - t_gz) * inv + (lde[i] - t_z) * inv) % P)
        assert oracle.fib_deep(lde, q, n, shift, z, t_z, t_gz, t_ggz, q_z).tolist() == d_ref, (n, blow)


def test_poly_eval_is_horner():
    import random
    rnd = random.Random(6)
    assert oracle.poly_eval([], 5) == 0                                          # src/math/polynomial.rs:135-137
    for deg in (0, 1, 2, 17, 300):
        c = [rnd.randrange(P) for _ in range(deg + 1)]
        for x in (0, 1, P - 1, rnd.randrange(P)):
            assert oracle.poly_eval(c, x) == _horner(c, x)


def test_merkle_get_proof_verifies_like_the_reference():
    # src/merkle.rs:50-101: every proof of every leaf verifies against the root, odd level sizes included; a wrong leaf does not
    for n in (1, 2, 3, 5, 8, 13):
        vals = np.arange(10, 10 + n, dtype=np.uint64)
        salts = (np.arange(n * 16, dtype=np.uint32) * 37 % 251).astype(np.uint8).reshape(n, 16)
        for s in (None, salts):
            lv = oracle.merkle_commit_values(vals, s)
            root = lv[-1][0].tobytes()
            for i in range(n):
                path, pos = oracle.merkle_get_proof(lv, i)
                leaf = (s[i].tobytes() if s is not None else b"") + int(vals[i]).to_bytes(8, "little")
                for tamper in (False, True):
                    cur = oracle.hash_leaf(leaf if not tamper else leaf[:-1] + b"\x7f")
                    for sib, right in zip(path, pos):
                        cur = oracle.hash_node(sib, cur) if right else oracle.hash_node(cur, sib)
                    assert (cur == root) == (not tamper)
            assert oracle.merkle_get_proof(lv, n) is None
