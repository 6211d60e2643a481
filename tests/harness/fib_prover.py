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
  constraint / quotient  -> toyni_fib_quotient_device          (src/fibonacci.rs:133-150, one kernel)
  OOD evaluations        -> toyni_poly_eval_device             (t_z, t_gz, t_ggz share one read of the coefficients)
  DEEP layer             -> toyni_fib_deep_device              (src/fibonacci.rs:186-198, one inversion per 8 points)
  FRI fold loop          -> toyni_fri_commit_phase_device      (per round: fold + leaf hashes in one sweep, node levels, root to the
                                                               transcript callback)
  query openings         -> toyni_merkle_open_device           (all ~1 700 paths gathered on the device, ONE copy to the host)
The only torch arithmetic left is the masking of 140 coefficients.

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


class DeviceTree:
    """Merkle tree built and kept on the device (src/fibonacci.rs:340-361).  `levels` may already hold the tree (a FRI round's
    fold + commit wrote it); otherwise it is built here."""

    def __init__(self, values_i32: torch.Tensor, salts, stream: int, levels=None):
        n = values_i32.numel()
        self.n = n
        self.values = values_i32
        self.salts = salts
        if levels is None:
            levels = torch.empty((_lib.toyni_merkle_total_digests(n), 32), dtype=torch.uint8, device=values_i32.device)
            toyni_amd.merkle_commit_device(values_i32.data_ptr(), salts.data_ptr() if salts is not None else 0, n, levels.data_ptr(), stream=stream)
        self.levels = levels

    def root(self) -> bytes:
        return bytes(self.levels[-1].cpu().numpy().tobytes())       # 32 bytes D2H (synchronises the stream)

    def open_into(self, d_indices: int, count: int, d_out: int, stream: int) -> None:
        """open_merkle (src/fibonacci.rs:366-375) for `count` positions: paths, salts and values gathered by one kernel."""
        toyni_amd.prover.merkle_open_device(self.levels.data_ptr(), self.n, self.values.data_ptr(),
                                            self.salts.data_ptr() if self.salts is not None else 0, d_indices, count, d_out, stream=stream)


def fibonacci_trace(n: int) -> np.ndarray:
    """In-field Fibonacci column (the reference test's u64 wrapping_add stops being a field sequence after fib(93))."""
    out = np.empty(n, dtype=np.uint64)
    a, b = 1, 1
    for i in range(n):
        out[i] = a
        a, b = b, (a + b) % P
    return out


def generate_proof(trace_col: np.ndarray, seed: int = 0, stats: dict = None, timing: dict = None, raw: bool = False):
    """raw=True returns the proof with its openings still in the serialized record form the device wrote (what a prover would
    put on the wire; `expand_proof` turns it into the field-by-field structure of StarkProof / QueryProof)."""
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
    pv = toyni_amd.prover
    bound = 1 << (n + MASK_DEGREE - 1).bit_length()               # fri_degree_bound = next_power_of_two, src/fibonacci.rs:218
    final_size = N // bound
    # every salt of the proof in one draw: 3 LDE-size trees + the salted FRI layers (16 bytes per leaf, src/fibonacci.rs:341-343)
    salted_leaves = 3 * N + sum(N >> k for k in range(1, 64) if (N >> k) > final_size)
    salt_pool = torch.randint(0, 256, (salted_leaves, 16), dtype=torch.uint8, device=dev, generator=rng)
    salt_off = [0]

    def take_salts(count):
        sl = salt_pool[salt_off[0]:salt_off[0] + count]
        salt_off[0] += count
        return sl

    def ntt_dev(ctx, t32, inverse, shift=1):
        ctx.run_device(t32.data_ptr(), t32.data_ptr(), 1, inverse, stream=stream, shift=shift)

    # ---- 1. trace polynomial + masking (src/fibonacci.rs:110-121): T_hat = T + (x^n - 1) R ----
    log_c = (n + MASK_DEGREE - 1).bit_length()                     # compact length 2^log_c >= n + MASK_DEGREE
    assert log_c <= log_N
    compact = torch.zeros(1 << log_c, dtype=torch.int32, device=dev)
    compact[:n] = torch.from_numpy(trace_col.astype(np.int32)).to(dev)
    ctx_n.run_device(compact.data_ptr(), compact.data_ptr(), 1, True, stream=stream)      # interpolate: one INTT
    r = torch.randint(0, P, (MASK_DEGREE,), dtype=torch.int64, device=dev, generator=rng)
    compact[:MASK_DEGREE] = ((compact[:MASK_DEGREE].to(torch.int64) - r) % P).to(torch.int32)           # - R
    compact[n:n + MASK_DEGREE] = ((compact[n:n + MASK_DEGREE].to(torch.int64) + r) % P).to(torch.int32)  # + x^n R
    ncoef_t = n + MASK_DEGREE                                      # trace_poly = compact[:ncoef_t], for the OOD evaluations
    # LDE: one coset FFT of the masked polynomial; the zero padding up to N is implied (toyni_lde_device)
    trace_lde = torch.empty(N, dtype=torch.int32, device=dev)
    ctx_N.lde_device(compact.data_ptr(), trace_lde.data_ptr(), 1, log_N - log_c, COSET_SHIFT, stream=stream)
    trace_tree = DeviceTree(trace_lde, take_salts(N), stream)
    lap("1_interpolate_mask_lde_commit")

    # ---- 2. constraint & quotient (src/fibonacci.rs:133-153): one kernel; then the reference's two ifft calls ----
    c32 = torch.empty(N, dtype=torch.int32, device=dev)
    q32 = torch.empty(N, dtype=torch.int32, device=dev)
    pv.fib_quotient_device(ctx_N, trace_lde.data_ptr(), c32.data_ptr(), q32.data_ptr(), 5, COSET_SHIFT, stream=stream)
    ntt_dev(ctx_N, c32, True, shift=COSET_SHIFT)                   # ifft #1 (c_poly; re-evaluating it on the coset is the identity)
    q_poly = q32.clone()
    ntt_dev(ctx_N, q_poly, True, shift=COSET_SHIFT)                # ifft #2
    quotient_tree = DeviceTree(q32, take_salts(N), stream)
    trace_commitment = trace_tree.root()
    quotient_commitment = quotient_tree.root()
    lap("2_constraint_quotient_commit")

    # ---- 3./4. Fiat-Shamir, OOD evaluations (src/fibonacci.rs:155-183) ----
    tr = Transcript()
    tr.absorb(trace_commitment)
    tr.absorb(quotient_commitment)
    z = derive_z(tr, N)
    ood_dev = torch.empty(4, dtype=torch.int32, device=dev)
    pv.poly_eval_device(ctx_N, compact.data_ptr(), ncoef_t, [z, g * z % P, g * g % P * z % P], ood_dev.data_ptr(), stream=stream)
    pv.poly_eval_device(ctx_N, q_poly.data_ptr(), N, [z], ood_dev.data_ptr() + 12, stream=stream)
    t_z, t_gz, t_ggz, q_z = (int(v) for v in ood_dev.cpu().numpy().view(np.uint32))
    c_z = (t_ggz - t_gz - t_z) % P * ((z - pow(g, n - 1, P)) % P) % P * ((z - pow(g, n - 2, P)) % P) % P
    assert c_z == q_z * ((pow(z, n, P) - 1) % P) % P, "Constraint check at z failed"      # src/fibonacci.rs:173-177
    for v in (t_z, t_gz, t_ggz, q_z):
        tr.absorb_field(v)
    lap("3_transcript_ood")

    # ---- 5. DEEP layer (src/fibonacci.rs:186-198): one kernel ----
    d32 = torch.empty(N, dtype=torch.int32, device=dev)
    pv.fib_deep_device(ctx_N, trace_lde.data_ptr(), q32.data_ptr(), d32.data_ptr(), 5, COSET_SHIFT, z, [t_z, t_gz, t_ggz, q_z], stream=stream)
    lap("5_deep")

    # ---- 6. FRI: the whole fold loop in ONE library call (src/fibonacci.rs:200-247); the transcript stays here, behind a callback:
    #         the library hands over round k's root, gets beta_{k+1} back, and only those 32 bytes cross PCIe per round ----
    layers = [d32]
    trees = [DeviceTree(d32, take_salts(N), stream)]
    commitments = [trees[0].root()]
    tr.absorb(commitments[0])
    sizes = []
    m = N
    while m > final_size:
        m //= 2
        sizes.append(m)
    layers_all = torch.empty(sum(sizes), dtype=torch.int32, device=dev)
    digests = [_lib.toyni_merkle_total_digests(h) for h in sizes]
    levels_all = torch.empty((sum(digests), 32), dtype=torch.uint8, device=dev)
    salted = [h for h in sizes if h != final_size]                 # the final layer is committed unsalted, :234-238
    salts_all = take_salts(sum(salted)) if salted else None

    def challenge(_round, root, want_beta):                        # depends on the previous round's root: rounds cannot be merged
        if root is not None:
            commitments.append(root)
            tr.absorb(root)
        return tr.squeeze_challenge() if want_beta else 0

    pv.fri_commit_phase_device(ctx_N, d32.data_ptr(), N, COSET_SHIFT, final_size, salts_all.data_ptr() if salts_all is not None else 0,
                               challenge, layers_all.data_ptr(), levels_all.data_ptr(), stream=stream)
    lo = dlo = slo = 0
    for h, nd in zip(sizes, digests):
        folded = layers_all[lo:lo + h]
        salts = salts_all[slo:slo + h] if h != final_size else None
        layers.append(folded)
        trees.append(DeviceTree(folded, salts, stream, levels=levels_all[dlo:dlo + nd]))
        lo, dlo = lo + h, dlo + nd
        if h != final_size:
            slo += h
    final_layer = [int(v) for v in layers[-1].cpu().numpy().view(np.uint32)]
    lap("6_fri_fold_commit")

    # ---- 7. queries (src/fibonacci.rs:249-295): every opening of the proof gathered on the device, ONE copy to the host ----
    half0 = N // 2
    qidx = tr.squeeze_indices(NUM_QUERIES, half0)
    q = np.asarray(qidx, dtype=np.int64)
    groups = [(trace_tree, np.stack([q, (q + BLOWUP) % N, (q + 2 * BLOWUP) % N], axis=1).reshape(-1)),     # T(x), T(g x), T(g^2 x)
              (quotient_tree, q),
              (trees[0], np.stack([q, q + half0], axis=1).reshape(-1))]                                   # DEEP layer: qi and its pair
    cur_idx = q.copy()
    for li in range(1, len(layers) - 1):
        half = layers[li].numel() // 2
        cur_idx = cur_idx % half
        groups.append((trees[li], np.stack([cur_idx, cur_idx + half], axis=1).reshape(-1)))
    all_idx = np.concatenate([ix for _, ix in groups]).astype(np.int32)
    d_idx = torch.from_numpy(all_idx).to(dev)
    sizes = [ix.size * pv.merkle_open_record_bytes(t.n) for t, ix in groups]
    out = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    ioff = boff = 0
    batch = []
    for (t, ix), sz in zip(groups, sizes):
        batch.append((t.levels.data_ptr(), t.n, t.values.data_ptr(), t.salts.data_ptr() if t.salts is not None else 0,
                      d_idx.data_ptr() + 4 * ioff, ix.size, out.data_ptr() + boff))
        ioff += ix.size
        boff += sz
    pv.merkle_open_groups_device(batch, stream=stream)            # every tree's openings: one launch (round 2: one per tree)
    records = out.cpu().numpy()
    lap("7_queries")
    if stats is not None:
        stats.update({"n": n, "lde": N, "folds": len(layers) - 1, "final_layer_size": final_size})
    proof = {
        "trace_len": n, "lde_size": N, "trace_commitment": trace_commitment, "quotient_commitment": quotient_commitment,
        "t_z": t_z, "t_gz": t_gz, "t_ggz": t_ggz, "q_z": q_z, "fri_commitments": commitments, "fri_final_layer": final_layer,
        "query_indices": qidx,
        "opening_records": records,                                   # serialized openings, in the group order above
        "opening_groups": [(t.n, t.salts is not None, ix.tolist()) for t, ix in groups],
    }
    return proof if raw else expand_proof(proof)


def expand_proof(proof):
    """Serialized opening records -> the QueryProof structure of src/fibonacci.rs:73-87 (what the verifier restatement reads)."""
    pv = toyni_amd.prover
    parsed, off = [], 0
    for tn, salted, idx in proof["opening_groups"]:
        sz = len(idx) * pv.merkle_open_record_bytes(tn)
        parsed.append(pv.parse_openings(proof["opening_records"][off:off + sz], tn, idx, salted))
        off += sz
    t_open, q_open, d_open, fri_open = parsed[0], parsed[1], parsed[2], parsed[3:]
    query_proofs = []
    for k, qi in enumerate(proof["query_indices"]):
        query_proofs.append({
            "index": qi,
            "trace_opening": t_open[3 * k], "trace_opening_g": t_open[3 * k + 1], "trace_opening_gg": t_open[3 * k + 2],
            "quotient_opening": q_open[k],
            "deep_opening": d_open[2 * k], "deep_opening_pair": d_open[2 * k + 1],
            "fri_openings": [(lo[2 * k], lo[2 * k + 1]) for lo in fri_open],
        })
    out = {k: v for k, v in proof.items() if k not in ("opening_records", "opening_groups", "query_indices")}
    out["query_proofs"] = query_proofs
    return out
