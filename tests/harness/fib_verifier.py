"""CPU restatement of the reference verifier `StarkVerifier::verify` (src/verifier.rs:14-232) -- TEST INFRASTRUCTURE.
Python integers + hashlib only (independent of the GPU path and of the harness' device code).  Returns True / False exactly
where the reference does; `why` (optional list) receives the failing check's name."""
import hashlib

P = 2013265921
NUM_QUERIES, BLOWUP, COSET_SHIFT = 44, 32, 7
MASK_DEGREE = 3 * NUM_QUERIES + 8


def root_of_unity(log_n):
    return pow(440564289, 1 << (27 - log_n), P)


def inv(a):
    assert a % P != 0, "Cannot invert zero"
    return pow(a, P - 2, P)


def h(b):
    return hashlib.sha256(b).digest()


class Transcript:   # src/transcript.rs
    def __init__(self):
        self.state = b"toyni-stark-v1"

    def absorb(self, d):
        self.state += d

    def absorb_field(self, v):
        self.absorb(int(v).to_bytes(8, "little"))

    def squeeze_challenge(self):
        d = h(self.state)
        self.state = d
        return int.from_bytes(d[:8], "little") % P

    def squeeze_indices(self, count, mx):
        out, seen = [], set()
        while len(out) < count:
            d = h(self.state)
            self.state = d
            i = int.from_bytes(d[:8], "little") % mx
            if i not in seen:
                seen.add(i)
                out.append(i)
        return out


def verify_merkle_proof(leaf, path, position, root):   # src/merkle.rs:87-101
    cur = h(b"\x00" + leaf)
    for sib, is_right in zip(path, position):
        cur = h(b"\x01" + sib + cur) if is_right else h(b"\x01" + cur + sib)
    return cur == root


def verify_opening(op, root):   # src/verifier.rs:234-237
    return verify_merkle_proof(op["salt"] + int(op["value"]).to_bytes(8, "little"), op["path"], op["position"], root)


def merkle_root_of(values):     # src/verifier.rs:240-243 (unsalted tree, odd levels duplicate the last node)
    cur = [h(b"\x00" + int(v).to_bytes(8, "little")) for v in values]
    while len(cur) > 1:
        cur = [h(b"\x01" + cur[i] + (cur[i + 1] if i + 1 < len(cur) else cur[i])) for i in range(0, len(cur), 2)]
    return cur[0]


def derive_z(tr, lde_size):     # src/verifier.rs:245-266 (set membership as z^N == 1 / (z/7)^N == 1)
    inv7 = inv(COSET_SHIFT)
    while True:
        z = tr.squeeze_challenge()
        if pow(z, lde_size, P) != 1 and pow(z * inv7 % P, lde_size, P) != 1:
            return z


def verify(proof, why=None):
    def fail(name):
        if why is not None:
            why.append(name)
        return False

    n, N = proof["trace_len"], proof["lde_size"]
    if N != n * BLOWUP:
        return fail("lde_size")
    log_n, log_N = n.bit_length() - 1, N.bit_length() - 1
    g = root_of_unity(log_n)
    w_N = root_of_unity(log_N)

    tr = Transcript()
    tr.absorb(proof["trace_commitment"])
    tr.absorb(proof["quotient_commitment"])
    z = derive_z(tr, N)
    for k in ("t_z", "t_gz", "t_ggz", "q_z"):
        tr.absorb_field(proof[k])

    # 2. OOD constraint check C(z) = Q(z) Z(z)  (src/verifier.rs:44-51)
    c_z = (proof["t_ggz"] - proof["t_gz"] - proof["t_z"]) % P * ((z - pow(g, n - 1, P)) % P) % P * ((z - pow(g, n - 2, P)) % P) % P
    if c_z != proof["q_z"] * ((pow(z, n, P) - 1) % P) % P:
        return fail("ood")

    # 3. FRI commitments, fixed fold schedule, constant committed final layer (src/verifier.rs:53-91)
    if not proof["fri_commitments"]:
        return fail("no_fri")
    bound = 1 << (n + MASK_DEGREE - 1).bit_length()
    final_size = N // bound
    expected_folds = (N // final_size).bit_length() - 1
    if len(proof["fri_commitments"]) != expected_folds + 1:
        return fail("fold_count")
    fl = proof["fri_final_layer"]
    if len(fl) != final_size:
        return fail("final_size")
    if any(v != fl[0] for v in fl):
        return fail("final_not_constant")
    if merkle_root_of(fl) != proof["fri_commitments"][-1]:
        return fail("final_commitment")
    tr.absorb(proof["fri_commitments"][0])
    betas = []
    for c in proof["fri_commitments"][1:]:
        betas.append(tr.squeeze_challenge())
        tr.absorb(c)

    # 4. queries (src/verifier.rs:93-229)
    half0 = N // 2
    qidx = tr.squeeze_indices(NUM_QUERIES, half0)
    if len(proof["query_proofs"]) != NUM_QUERIES:
        return fail("query_count")
    half_inv = inv(2)
    for qi, qp in zip(qidx, proof["query_proofs"]):
        if qp["index"] != qi:
            return fail("query_index")
        if len(qp["fri_openings"]) != expected_folds - 1:
            return fail("fri_opening_count")
        for k in ("trace_opening", "trace_opening_g", "trace_opening_gg"):
            if not verify_opening(qp[k], proof["trace_commitment"]):
                return fail("trace_merkle")
        if (qp["trace_opening"]["index"] != qi or qp["trace_opening_g"]["index"] != (qi + BLOWUP) % N
                or qp["trace_opening_gg"]["index"] != (qi + 2 * BLOWUP) % N):
            return fail("trace_index")
        if not verify_opening(qp["quotient_opening"], proof["quotient_commitment"]):
            return fail("quotient_merkle")
        if not verify_opening(qp["deep_opening"], proof["fri_commitments"][0]) or not verify_opening(qp["deep_opening_pair"], proof["fri_commitments"][0]):
            return fail("deep_merkle")
        x_i = COSET_SHIFT * pow(w_N, qi, P) % P
        ixz = inv((x_i - z) % P)
        expected_deep = ((qp["quotient_opening"]["value"] - proof["q_z"]) * ixz + (qp["trace_opening_gg"]["value"] - proof["t_ggz"]) * ixz
                         + (qp["trace_opening_g"]["value"] - proof["t_gz"]) * ixz + (qp["trace_opening"]["value"] - proof["t_z"]) * ixz) % P
        if qp["deep_opening"]["value"] != expected_deep:
            return fail("deep_value")
        a0, b0 = qp["deep_opening"]["value"], qp["deep_opening_pair"]["value"]
        prev = ((a0 + b0) * half_inv + (a0 - b0) * half_inv % P * betas[0] % P * inv(x_i)) % P
        pos = qi
        for layer, (op, op_pair) in enumerate(qp["fri_openings"]):
            fold_k = layer + 1
            half = (N >> fold_k) // 2
            lo = pos % half
            if not verify_opening(op, proof["fri_commitments"][fold_k]) or not verify_opening(op_pair, proof["fri_commitments"][fold_k]):
                return fail("fri_merkle")
            if (op["value"] if pos == lo else op_pair["value"]) != prev:
                return fail("fri_consistency")
            x = pow(COSET_SHIFT * pow(w_N, lo, P) % P, 1 << fold_k, P)
            a_l, b_l = op["value"], op_pair["value"]
            prev = ((a_l + b_l) * half_inv + (a_l - b_l) * half_inv % P * betas[fold_k] % P * inv(x)) % P
            pos = lo
        if fl[pos] != prev:  # src/verifier.rs:226
            return fail("final_value")
    return True
