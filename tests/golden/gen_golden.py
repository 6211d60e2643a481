#!/usr/bin/env python3
"""Generate tests/golden/vectors.json with an INDEPENDENT big-integer model.

Nothing here shares code with oracle/ or toyni_amd/: transforms are the O(n^2) definition
out[k] = sum_j in[j] * w^(j k) with Python ints, the fold expectation is the closed form
f_e(x^2) + beta * f_o(x^2).  The vectors pin the oracle (tests/test_oracle.py) and, through
it and directly, the HIP path (tests/test_gpu_parity.py).

Sources of the cases (reference tests, /root/reference):
  src/ntt.rs:338-357   n=8, coeffs [1,2,3,0,...] -> evals[0]=6, evals[1]=1+2w+3w^2
  src/ntt.rs:263-287   n=256, input (7 i + 3): GPU == CPU
  src/ntt.rs:322-336   roundtrip n=256
  src/ntt.rs:359-379   roots of unity n=16
  src/babybear.rs:219-284  field known answers
  src/math/domain.rs:220-242  coset FFT (shift 7, n=8, 1+2x+3x^2) == Horner at every coset point
  src/math/fri.rs:27-48 + src/verifier.rs:69-75  fold formula / constant final layer

Run:  python tests/golden/gen_golden.py      (rewrites vectors.json; deterministic)
"""
import json
import os

P = 2**31 - 2**27 + 1
GEN_2_27 = pow(31, 15, P)  # 440564289, order exactly 2^27
assert GEN_2_27 == 440564289
assert pow(GEN_2_27, 2**27, P) == 1 and pow(GEN_2_27, 2**26, P) == P - 1


def root(log_n):
    return pow(GEN_2_27, 1 << (27 - log_n), P)


def splitmix64(x):
    m = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return x ^ (x >> 31)


def rnd(n, seed):
    return [splitmix64((seed + i) & ((1 << 64) - 1)) % P for i in range(n)]


def dft(xs, w):
    n = len(xs)
    pw = [pow(w, e, P) for e in range(n)]
    return [sum(xs[j] * pw[(j * k) % n] for j in range(n)) % P for k in range(n)]


def horner(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


def main():
    out = {"p": P, "generator_2_27": GEN_2_27}

    out["field"] = {
        "add_100_200": (100 + 200) % P,
        "sub_200_100": (200 - 100) % P,
        "mul_100_200": (100 * 200) % P,
        "new_p_plus_5": (P + 5) % P,
        "inv_7": pow(7, P - 2, P),
        "pow_3_4": pow(3, 4, P),
        "neg_100": (P - 100) % P,
        "div_100_7": (100 * pow(7, P - 2, P)) % P,
        "half_inv": pow(2, P - 2, P),
        "mul_pm1_pm1": ((P - 1) * (P - 1)) % P,
        "barrett_mu": (1 << 64) // P,
    }
    out["roots_of_unity"] = [root(k) for k in range(28)]

    ntt_cases = []
    # src/ntt.rs:338-357
    ntt_cases.append({"name": "kat_n8_1_2_3", "n": 8, "input": [1, 2, 3, 0, 0, 0, 0, 0]})
    # src/ntt.rs:263-287 / :322-336
    ntt_cases.append({"name": "pattern_7i3_n256", "n": 256, "input": [(7 * i + 3) % P for i in range(256)]})
    ntt_cases.append({"name": "pattern_7i3_n16", "n": 16, "input": [(7 * i + 3) % P for i in range(16)]})
    for logn, seed in [(1, 11), (2, 12), (3, 13), (5, 15), (6, 16), (10, 20)]:
        ntt_cases.append({"name": f"splitmix_n{1 << logn}", "n": 1 << logn, "input": rnd(1 << logn, 0x70796E69 + (seed << 32))})
    n = 64
    ntt_cases.append({"name": "zeros_n64", "n": n, "input": [0] * n})
    ntt_cases.append({"name": "all_pm1_n64", "n": n, "input": [P - 1] * n})
    ntt_cases.append({"name": "delta0_n64", "n": n, "input": [1] + [0] * (n - 1)})
    ntt_cases.append({"name": "delta1_n64", "n": n, "input": [0, 1] + [0] * (n - 2)})
    ntt_cases.append({"name": "const_n64", "n": n, "input": [123456789] * n})
    for c in ntt_cases:
        w = root(c["n"].bit_length() - 1)
        c["omega"] = w
        c["forward"] = dft(c["input"], w)
        # inverse transform of `input` (as evaluations): n^-1 * sum in[j] w^(-jk)
        ninv = pow(c["n"], P - 2, P)
        c["inverse"] = [(v * ninv) % P for v in dft(c["input"], pow(w, P - 2, P))]
    out["ntt"] = ntt_cases

    # coset FFT (src/math/domain.rs:107-123, test :220-242)
    coset_cases = []
    for name, size, shift, coeffs in [
        ("domain_rs_test_n8_shift7", 8, 7, [1, 2, 3]),
        ("roundtrip_n8_shift7", 8, 7, [(3 * i + 1) % P for i in range(8)]),
        ("splitmix_n64_shift7", 64, 7, rnd(40, 0xC05E7)),
        ("splitmix_n256_shift_big", 256, 1234567891, rnd(256, 0xC05E8)),
    ]:
        w = root(size.bit_length() - 1)
        pts = [(shift * pow(w, i, P)) % P for i in range(size)]
        coset_cases.append({"name": name, "size": size, "shift": shift, "coeffs": coeffs,
                            "points": pts, "evals": [horner(coeffs, x) for x in pts]})
    out["coset"] = coset_cases

    # FRI fold closed form: f = f_e(x^2) + x f_o(x^2); fold_beta(f)(x^2) = f_e(x^2) + beta f_o(x^2)
    fold_cases = []
    for name, N, shift, deg, seed in [
        ("fold_n4_std", 4, 1, 3, 1), ("fold_n8_shift7", 8, 7, 5, 2),
        ("fold_n64_shift7", 64, 7, 40, 3), ("fold_n256_shift7", 256, 7, 256, 4),
        ("fold_n2", 2, 7, 2, 5),
    ]:
        coeffs = rnd(deg, 0xF01D0000 + seed)
        beta = splitmix64(0xBE7A + seed) % P
        w = root(N.bit_length() - 1)
        xs = [(shift * pow(w, i, P)) % P for i in range(N)]
        evals = [horner(coeffs, x) for x in xs]
        fe, fo = coeffs[0::2], coeffs[1::2]
        folded = [(horner(fe, x * x % P) + beta * horner(fo, x * x % P)) % P for x in xs[:N // 2]]
        fold_cases.append({"name": name, "n": N, "shift": shift, "beta": beta,
                           "xs": xs, "evals": evals, "folded": folded})
    out["fold"] = fold_cases

    # multi-layer: degree < bound codeword folds to a constant layer after log2(bound) folds
    # (src/fibonacci.rs:216-245, src/verifier.rs:69-75).  Closed form per layer as above.
    N, shift, bound = 256, 7, 16
    coeffs = rnd(bound, 0x1A7E8)
    betas = [splitmix64(0xBE7A00 + k) % P for k in range(4)]
    w = root(8)
    xs = [(shift * pow(w, i, P)) % P for i in range(N)]
    evals = [horner(coeffs, x) for x in xs]
    layers, c, pts = [], coeffs, xs
    for b in betas:
        fe, fo = c[0::2], c[1::2]
        fo = fo + [0] * (len(fe) - len(fo))
        c = [(a + b * o) % P for a, o in zip(fe, fo)]
        pts = [x * x % P for x in pts[:len(pts) // 2]]
        layers.append([horner(c, x) for x in pts])
    assert len(set(layers[-1])) == 1
    out["fold_layers"] = {"n": N, "shift": shift, "betas": betas, "evals": evals, "layers": layers}

    # Ext = F_p[X]/(X^4 - 11) fold (src/math/fri.rs:7-25, src/ext.rs:138-192), independent schoolbook model on Python ints
    def ext_mul(a, b):
        t = [0] * 7
        for i in range(4):
            for j in range(4):
                t[i + j] = (t[i + j] + a[i] * b[j]) % P
        return [(t[k] + 11 * t[k + 4]) % P if k + 4 < 7 else t[k] % P for k in range(4)]

    ext_cases = []
    for name, N, shift, seed in [("ext_fold_n8_shift7", 8, 7, 1), ("ext_fold_n64_shift7", 64, 7, 2), ("ext_fold_n2", 2, 7, 3)]:
        flat = rnd(4 * N, 0xE47F01D0 + seed)
        evals = [flat[4 * i:4 * i + 4] for i in range(N)]
        beta = rnd(4, 0xBE7AE47 + seed)
        w = root(N.bit_length() - 1)
        xs = [(shift * pow(w, i, P)) % P for i in range(N)]
        half_inv = pow(2, P - 2, P)
        folded = []
        for i in range(N // 2):
            a, b = evals[i], evals[i + N // 2]
            avg = [(x + y) * half_inv % P for x, y in zip(a, b)]
            diff = [(x - y) * half_inv % P for x, y in zip(a, b)]
            prod = ext_mul(ext_mul(diff, beta), [pow(xs[i], P - 2, P), 0, 0, 0])
            folded.append([(u + v) % P for u, v in zip(avg, prod)])
        ext_cases.append({"name": name, "n": N, "xs": xs, "beta": beta, "evals": evals, "folded": folded})
    # X^4 = 11 and a product known by hand: (1 + X)(1 + X^3) = 1 + X + X^3 + X^4 = 12 + X + X^3
    out["ext"] = {"x4": ext_mul([0, 0, 1, 0], [0, 0, 1, 0]), "hand_product": ext_mul([1, 1, 0, 0], [1, 0, 0, 1]), "fold": ext_cases}
    assert out["ext"]["x4"] == [11, 0, 0, 0] and out["ext"]["hand_product"] == [12, 1, 0, 1]

    # SHA-256 Merkle roots (src/merkle.rs, src/fibonacci.rs:340-361) from hashlib
    import hashlib
    def hl(b):
        return hashlib.sha256(b).digest()
    def root_of(leaves):
        cur = [hl(b"\x00" + l) for l in leaves]
        while len(cur) > 1:
            cur = [hl(b"\x01" + cur[i] + (cur[i + 1] if i + 1 < len(cur) else cur[i])) for i in range(0, len(cur), 2)]
        return cur[0].hex()
    mk = []
    for n_leaves in (1, 2, 3, 4, 7, 64):
        vals = rnd(n_leaves, 0x3E2C1E + n_leaves)
        salts = [bytes((17 * i + 13 * j + n_leaves) & 0xFF for j in range(16)) for i in range(n_leaves)]
        mk.append({"n": n_leaves, "values": vals, "salts_hex": [s_.hex() for s_ in salts],
                   "root_unsalted": root_of([v.to_bytes(8, "little") for v in vals]),
                   "root_salted": root_of([s_ + v.to_bytes(8, "little") for s_, v in zip(salts, vals)])})
    out["merkle"] = mk

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
