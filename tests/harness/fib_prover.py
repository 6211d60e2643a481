"""Prover-shaped HARNESS for BASELINE configs[0]/[2] (Fibonacci AIR, blowup 32): the caller of the hot path, not a product.

It follows the protocol of the reference prover `StarkProver::generate_proof` (src/fibonacci.rs:99-310) step for step --
masking, trace/quotient/DEEP layers, salted Merkle commitments, Fiat-Shamir transcript (src/transcript.rs), the FRI fold
loop with a fixed round count, 44 queries -- but does the heavy steps the way SURVEY.md F3/F5 prescribes, because the
reference's own Lagrange interpolation (O(n^3)) and Horner LDE (O(N d)) are infeasible at trace_len = 2^16:

  interpolate            -> one INTT of size n            (toyni_amd device NTT)
  LDE on the coset       -> one coset FFT of size 32 n    (coset scaling fused on the device)
  the two `ifft` calls   -> coset INTTs of size 32 n      (src/fibonacci.rs:145,151)
  FRI fold loop          -> structured-point fold kernel  (src/fibonacci.rs:220-245)
  Merkle commitments     -> GPU SHA-256 trees, levels stay on the device; only roots and the 44 query paths cross PCIe
  pointwise constraint / quotient / DEEP arithmetic -> torch int64 elementwise ops on the device (plumbing: products of
                            two residues stay below 2^62).  These are SURVEY 8(f)3 "next" and have no kernels of their own.

Field arithmetic is exact, so every value equals what the reference's formulas define on the same randomness; the proof is
checked by tests/harness/fib_verifier.py, a CPU restatement of src/verifier.rs.
"""
import hashlib

import numpy as np
import torch

import toyni_amd
from toyni_amd._lib import lib as _lib

P = 2013265921
NUM_QUERIES = 44          # src/fibonacci.rs:11
BLOWUP = 32               # src/fibonacci.rs:14
COSET_SHIFT = 7           # src/fibonacci.rs:16
MASK_DEGREE = 3 * NUM_QUERIES + 8   # src/fibonacci.rs:19


# ---- field helpers on int64 tensors (device plumbing) ----
def mulmod(a, b):
    return (a * b) % P


def powmod_t(a, e: int):
    r = torch.ones_like(a)
    base = a.clone()
    while e:
        if e & 1:
            r = mulmod(r, base)
        base = mulmod(base, base)
        e >>= 1
    return r


def invmod_t(a):
    return powmod_t(a, P - 2)


def root_of_unity(log_n: int) -> int:      # src/babybear.rs:118-126
    return pow(440564289, 1 << (27 - log_n), P)


class Transcript:
    """src/transcript.rs:12-72"""

    def __init__(self):
        self.state = b"toyni-stark-v1"

    def absorb(self, data: bytes):
        self.state += data

    def absorb_field(self, v: int):
        self.absorb(int(v).to_bytes(8, "little"))

    def squeeze_challenge(self) -> int:
        h = hashlib.sha256(self.state).digest()
        self.state = h
        return int.from_bytes(h[:8], "little") % P      # from_bytes_mod_order, src/babybear.rs:65-71

    def squeeze_indices(self, count: int, mx: int):
        out, seen = [], set()
        while len(out) < count:
            h = hashlib.sha256(self.state).digest()
            self.state = h
            idx = int.from_bytes(h[:8], "little") % mx
            if idx not in seen:
                seen.add(idx)
                out.append(idx)
        return out


def derive_z(tr: Transcript, lde_size: int) -> int:
    """src/fibonacci.rs:379-399.  Membership in <w_N> / 7<w_N> is z^N == 1 / (z/7)^N == 1; the g*z, g^2*z tests are implied
    (the shifted domain is closed under multiplication by g = w_N)."""
    inv7 = pow(COSET_SHIFT, P - 2, P)
    while True:
        z = tr.squeeze_challenge()
        if pow(z, lde_size, P) != 1 and pow(z * inv7 % P, lde_size, P) != 1:
            return z


def poly_eval(coeffs: torch.Tensor, z: int) -> int:
    """sum c_i z^i on the device (int64 tensor of coefficients): powers by doubling, products < 2^62, partial sums < 2^53."""
    n = coeffs.numel()
    pw = torch.ones(1, dtype=torch.int64, device=coeffs.device)
    step = z % P
    while pw.numel() < n:
        pw = torch.cat([pw, mulmod(pw, step)])
        step = step * step % P
    return int((mulmod(coeffs, pw[:n]).sum() % P).item())


class DeviceTree:
    """Merkle tree built and kept on the device (toyni_merkle_commit_device); src/fibonacci.rs:340-361."""

    def __init__(self, values_i32: torch.Tensor, rng, salted: bool, stream: int):
        n = values_i32.numel()
        self.n = n
        self.values = values_i32
        self.salts = None
        d_salts = 0
        if salted:
            self.salts = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device=values_i32.device, generator=rng)
            d_salts = self.salts.data_ptr()
        total = _lib.toyni_merkle_total_digests(n)
        self.levels = torch.empty((total, 32), dtype=torch.uint8, device=values_i32.device)
        toyni_amd.merkle_commit_device(values_i32.data_ptr(), d_salts, n, self.levels.data_ptr(), stream=stream)
        self.offsets, off, m = [], 0, n
        while True:
            self.offsets.append((off, m))
            if m == 1:
                break
            off += m
            m = (m + 1) // 2

    def root(self) -> bytes:
        return bytes(self.levels[-1].cpu().numpy().tobytes())

    def open_many(self, indices):
        """open_merkle (src/fibonacci.rs:366-375) + MerkleTree::get_proof (src/merkle.rs:50-80) for a list of positions:
        ONE gather of all path digests / values / salts on the device, one copy to the host."""
        rows, positions = [], []
        for index in indices:
            pos, cur = [], index
            for off, m in self.offsets[:-1]:
                sib = cur + 1 if cur % 2 == 0 else cur - 1
                if sib >= m:
                    rows.append(off + cur)
                    pos.append(True)
                else:
                    rows.append(off + sib)
                    pos.append(cur % 2 == 1)
                cur //= 2
            positions.append(pos)
        dev = self.levels.device
        idx_t = torch.tensor(list(indices), dtype=torch.long, device=dev)
        depth = len(self.offsets) - 1
        paths = self.levels[torch.tensor(rows, dtype=torch.long, device=dev)].cpu().numpy().reshape(len(indices), depth, 32) if rows \
            else np.zeros((len(indices), 0, 32), np.uint8)
        vals = self.values[idx_t].cpu().numpy()
        salts = self.salts[idx_t].cpu().numpy() if self.salts is not None else None
        return [{"index": index, "value": int(vals[k]), "path": [p.tobytes() for p in paths[k]], "position": positions[k],
                 "salt": salts[k].tobytes() if salts is not None else b""} for k, index in enumerate(indices)]

    def open(self, index: int):
        return self.open_many([index])[0]


def fibonacci_trace(n: int) -> np.ndarray:
    """In-field Fibonacci column (the reference test's u64 wrapping_add stops being a field sequence after fib(93))."""
    out = np.empty(n, dtype=np.uint64)
    a, b = 1, 1
    for i in range(n):
        out[i] = a
        a, b = b, (a + b) % P
    return out


def generate_proof(trace_col: np.ndarray, seed: int = 0, stats: dict = None, timing: dict = None):
    import time
    dev = torch.device("cuda", 0)
    _t = [time.perf_counter()]

    def lap(name):
        if timing is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            timing[name] = timing.get(name, 0.0) + (now - _t[0]) * 1e3
            _t[0] = now

    stream = torch.cuda.current_stream().cuda_stream
    rng = torch.Generator(device=dev)       # salts and mask coefficients (the reference uses rand::thread_rng)
    rng.manual_seed(seed)
    n = int(trace_col.size)
    assert n & (n - 1) == 0
    log_n = n.bit_length() - 1
    N = n * BLOWUP
    log_N = log_n + 5
    g = root_of_unity(log_n)                       # domain.group_gen()
    ctx_n = toyni_amd.ntt.get_or_create_ctx(n)
    ctx_N = toyni_amd.ntt.get_or_create_ctx(N)

    def ntt_dev(ctx, t32, inverse, shift=1):
        ctx.run_device(t32.data_ptr(), t32.data_ptr(), 1, inverse, stream=stream, shift=shift)

    # ---- 1. trace polynomial + masking (src/fibonacci.rs:110-121): T_hat = T + (x^n - 1) R ----
    coeffs = torch.from_numpy(trace_col.astype(np.int32)).to(dev)
    ntt_dev(ctx_n, coeffs, True)                                   # interpolate: one INTT
    r = torch.randint(0, P, (MASK_DEGREE,), dtype=torch.int64, device=dev, generator=rng)
    poly = torch.zeros(N, dtype=torch.int64, device=dev)
    poly[:n] = coeffs.to(torch.int64)
    poly[:MASK_DEGREE] = (poly[:MASK_DEGREE] - r) % P
    poly[n:n + MASK_DEGREE] = (poly[n:n + MASK_DEGREE] + r) % P
    trace_poly = poly[: n + MASK_DEGREE].clone()                  # coefficients, for the OOD evaluations
    # LDE: one coset FFT of the masked polynomial (n + MASK_DEGREE coefficients); the zero padding up to N is
    # implied, not stored (toyni_lde_device: the first pass reads only the words that exist)
    log_c = (n + MASK_DEGREE - 1).bit_length()                     # compact length 2^log_c >= n + MASK_DEGREE
    assert log_c <= log_N
    compact = poly[: 1 << log_c].to(torch.int32).contiguous()
    trace_lde = torch.empty(N, dtype=torch.int32, device=dev)
    ctx_N.lde_device(compact.data_ptr(), trace_lde.data_ptr(), 1, log_N - log_c, COSET_SHIFT, stream=stream)
    trace_tree = DeviceTree(trace_lde, rng, True, stream)
    trace_commitment = trace_tree.root()
    lap("1_interpolate_mask_lde_commit")

    # x_i = 7 w_N^i
    xs32 = torch.empty(N, dtype=torch.int32, device=dev)
    ctx_N.domain_elements_device(xs32.data_ptr(), N, COSET_SHIFT, stream=stream)   # lde_domain.elements(), src/fibonacci.rs:133
    xs = xs32.to(torch.int64)

    # ---- 2. constraint & quotient (src/fibonacci.rs:133-153) ----
    T = trace_lde.to(torch.int64)
    T_g = torch.roll(T, -BLOWUP)                                   # T(g x_i) = trace_lde[(i + BLOWUP) % N]
    T_gg = torch.roll(T, -2 * BLOWUP)
    fib = (T_gg - T_g - T) % P
    b1 = (xs - pow(g, n - 1, P)) % P
    b2 = (xs - pow(g, n - 2, P)) % P
    c_evals = mulmod(mulmod(fib, b1), b2)
    c32 = c_evals.to(torch.int32)
    c_poly = c32.clone()
    ntt_dev(ctx_N, c_poly, True, shift=COSET_SHIFT)                # ifft #1 (c_poly; re-evaluating it on the coset is the identity)
    zh = (powmod_t(xs, n) - 1) % P                                 # Z_H(x_i) = x_i^n - 1
    q_evals = mulmod(c_evals, invmod_t(zh))
    q32 = q_evals.to(torch.int32)
    q_poly_t = q32.clone()
    ntt_dev(ctx_N, q_poly_t, True, shift=COSET_SHIFT)              # ifft #2
    q_poly = q_poly_t.to(torch.int64)
    quotient_tree = DeviceTree(q32, rng, True, stream)
    quotient_commitment = quotient_tree.root()
    lap("2_constraint_quotient_commit")

    # ---- 3./4. Fiat-Shamir, OOD evaluations (src/fibonacci.rs:155-183) ----
    tr = Transcript()
    tr.absorb(trace_commitment)
    tr.absorb(quotient_commitment)
    z = derive_z(tr, N)
    t_z, t_gz, t_ggz = poly_eval(trace_poly, z), poly_eval(trace_poly, g * z % P), poly_eval(trace_poly, g * g % P * z % P)
    q_z = poly_eval(q_poly, z)
    c_z = (t_ggz - t_gz - t_z) % P * ((z - pow(g, n - 1, P)) % P) % P * ((z - pow(g, n - 2, P)) % P) % P
    assert c_z == q_z * ((pow(z, n, P) - 1) % P) % P, "Constraint check at z failed"      # src/fibonacci.rs:173-177
    for v in (t_z, t_gz, t_ggz, q_z):
        tr.absorb_field(v)
    lap("3_transcript_ood")

    # ---- 5. DEEP layer (src/fibonacci.rs:186-198) ----
    inv_xz = invmod_t((xs - z) % P)
    d_evals = mulmod(((q_evals - q_z) + (T_gg - t_ggz) + (T_g - t_gz) + (T - t_z)) % P, inv_xz)

    lap("5_deep")

    # ---- 6. FRI: fold + commit (src/fibonacci.rs:200-247) ----
    bound = 1 << (n + MASK_DEGREE - 1).bit_length()               # next_power_of_two
    final_size = N // bound
    layers = [d_evals.to(torch.int32)]
    trees = [DeviceTree(layers[0], rng, True, stream)]
    commitments = [trees[0].root()]
    tr.absorb(commitments[0])
    x0 = COSET_SHIFT
    while layers[-1].numel() > final_size:
        beta = tr.squeeze_challenge()
        cur = layers[-1]
        m = cur.numel()
        folded = torch.empty(m // 2, dtype=torch.int32, device=dev)
        toyni_amd.fri_fold_device(ctx_N, cur.data_ptr(), folded.data_ptr(), m, beta, x0, stream=stream)
        x0 = x0 * x0 % P                                          # xs truncated and squared, src/fibonacci.rs:228-231
        layers.append(folded)
        trees.append(DeviceTree(folded, rng, folded.numel() != final_size, stream))
        commitments.append(trees[-1].root())
        tr.absorb(commitments[-1])
    final_layer = [int(v) for v in layers[-1].cpu().numpy()]
    lap("6_fri_fold_commit")

    # ---- 7. queries (src/fibonacci.rs:249-295) ----
    half0 = N // 2
    qidx = tr.squeeze_indices(NUM_QUERIES, half0)
    t_open = trace_tree.open_many([i for qi in qidx for i in (qi, (qi + BLOWUP) % N, (qi + 2 * BLOWUP) % N)])
    q_open = quotient_tree.open_many(qidx)
    d_open = trees[0].open_many([i for qi in qidx for i in (qi, qi + half0)])
    fri_open = []                                   # per intermediate layer: openings of (idx, idx + half) for every query
    cur_idx = list(qidx)
    for li in range(1, len(layers) - 1):
        half = layers[li].numel() // 2
        cur_idx = [i % half for i in cur_idx]
        fri_open.append(trees[li].open_many([j for i in cur_idx for j in (i, i + half)]))
    query_proofs = []
    for k, qi in enumerate(qidx):
        query_proofs.append({
            "index": qi,
            "trace_opening": t_open[3 * k], "trace_opening_g": t_open[3 * k + 1], "trace_opening_gg": t_open[3 * k + 2],
            "quotient_opening": q_open[k],
            "deep_opening": d_open[2 * k], "deep_opening_pair": d_open[2 * k + 1],
            "fri_openings": [(lo[2 * k], lo[2 * k + 1]) for lo in fri_open],
        })

    lap("7_queries")
    if stats is not None:
        stats.update({"n": n, "lde": N, "folds": len(layers) - 1, "final_layer_size": final_size})
    return {
        "trace_len": n, "lde_size": N, "trace_commitment": trace_commitment, "quotient_commitment": quotient_commitment,
        "t_z": t_z, "t_gz": t_gz, "t_ggz": t_ggz, "q_z": q_z, "fri_commitments": commitments, "fri_final_layer": final_layer,
        "query_proofs": query_proofs,
    }
