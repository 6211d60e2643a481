// Pointwise steps of the Fibonacci STARK prover on the LDE coset, for gfx950.
//
// What they restate (reference src/fibonacci.rs, `StarkProver::generate_proof`), element for element and exactly (field
// arithmetic, canonical outputs):
//   fib_quotient   :133-150  c(x) = (T(g^2 x) - (T(g x) + T(x))) (x - g^(n-1)) (x - g^(n-2)),  q(x) = c(x) / (x^n - 1)
//   fib_deep       :186-198  d(x) = (q(x) - q_z)/(x - z) + (T(g^2 x) - t_ggz)/(x - z) + (T(g x) - t_gz)/(x - z) + (T(x) - t_z)/(x - z)
//   poly_eval      src/math/polynomial.rs:134-144 (Horner), as per-thread Horner runs joined by powers of the point
//   merkle_open    src/merkle.rs:50-80 (get_proof) + open_merkle src/fibonacci.rs:366-375
//   fold + commit  src/fibonacci.rs:222-245: fri_fold of layer k fused with the leaf hashes of layer k + 1's tree
// on x_i = shift * w_N^i, where T(g x_i) = trace_lde[(i + B) mod N] (g = w_n = w_N^B, B = N / n the blow-up).
// The reference evaluates every one of these by Horner per point on one core (O(N d)); here they are HBM-bound sweeps.
// Bodies are plain C++ (tests/emu can step them); the kernels wrapping them live in toyni_hip.hip.
#pragma once
#include "merkle_kernels.hpp"
#include "ntt_kernels.hpp"

namespace toyni {

// x_i = shift * w_N^i from the context's forward two-level domain table; returned in MONTGOMERY form (x_i * R)
struct DomainArgs {
    SubDomain dom;        // w_N'^(i << s): the order-N subgroup inside the context's (possibly larger) forward domain table (ntt_kernels.hpp)
    uint32_t shiftR;      // Montgomery form of the coset shift
};
TOYNI_HD uint32_t domain_point_mont(const DomainArgs& d, uint64_t i) {
    return mont_mul(subdomain_mont(d.dom, (uint32_t)i), d.shiftR);   // shift * w^i * R
}

// a^-1 in Montgomery form (aR -> a^-1 R) by Fermat, as BabyBear::inverse (src/babybear.rs:111-114); 0 -> 0
TOYNI_HD uint32_t mont_inv(uint32_t aR) { return mont_inv_chain(aR); }   // bb_field.hpp: 41 products instead of the ladder's 60
TOYNI_HD uint32_t mont_pow(uint32_t aR, uint64_t e) {
    uint32_t r = BB_R1;
    while (e) {
        if (e & 1u) r = mont_mul(r, aR);
        aR = mont_mul(aR, aR);
        e >>= 1;
    }
    return r;
}

// ---- constraint and quotient (src/fibonacci.rs:133-150) ----
struct QuotientArgs {
    const uint32_t* trace;   // trace_lde, N words
    uint32_t* c_out;         // constraint evaluations (may be null)
    uint32_t* q_out;         // quotient evaluations
    DomainArgs dom;
    uint32_t log_N, log_blowup;
    uint32_t b1R, b2R;       // Montgomery forms of g^(n-1), g^(n-2) (boundary_constraint_1/2, :318-324)
    uint32_t shift_nR;       // Montgomery form of shift^n: x_i^n = shift^n * w_B^(i mod B), only B distinct values
    uint32_t wBR;            // Montgomery form of w_B = w_N^n
};
// 1 / Z_H(x_i) = 1 / (x_i^n - 1) for residue class t = i mod B, Montgomery form (one Fermat inversion per class, not per point)
TOYNI_HD uint32_t quotient_zh_inv(const QuotientArgs& a, uint32_t t) {
    const uint32_t xnR = mont_mul(a.shift_nR, mont_pow(a.wBR, t));   // (shift^n w_B^t) R
    return mont_inv(bb_sub(xnR, BB_R1));
}
TOYNI_HD void quotient_one(const QuotientArgs& a, uint64_t i, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t zh_invR, uint32_t& c, uint32_t& q) {
    const uint32_t xR = domain_point_mont(a.dom, i);
    const uint32_t fib = bb_sub(t2, bb_add(t1, t0));                      // fibonacci_constraint, :313-315
    const uint32_t u = mont_mul(fib, bb_sub(xR, a.b1R));                  // fib * (x - g^(n-1))      (plain)
    c = mont_mul(u, bb_sub(xR, a.b2R));                                   // ... * (x - g^(n-2))       (plain)
    q = mont_mul(c, zh_invR);                                             // c / Z_H(x)
}

// ---- DEEP layer (src/fibonacci.rs:186-198) ----
struct DeepArgs {
    const uint32_t* trace;
    const uint32_t* quot;
    uint32_t* out;
    DomainArgs dom;
    uint32_t log_N, log_blowup;
    uint32_t wNR;            // Montgomery form of w_N (step between consecutive points)
    uint32_t zR;             // Montgomery form of z
    uint32_t t_z, t_gz, t_ggz, q_z;  // plain
};
// K consecutive points share ONE Fermat inversion (Montgomery's trick); a zero x_i - z (excluded by derive_z,
// src/fibonacci.rs:379-399) is kept out of the product and contributes x^-1 = 0 for that point alone.
template <int K>
TOYNI_HD void deep_group(const DeepArgs& a, uint64_t i0, const uint32_t (&t0)[K], const uint32_t (&t1)[K], const uint32_t (&t2)[K],
                         const uint32_t (&qv)[K], uint32_t (&out)[K]) {
    uint32_t dR[K], pre[K];
    uint32_t xR = domain_point_mont(a.dom, i0);
    uint32_t acc = BB_R1;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        dR[j] = bb_sub(xR, a.zR);                 // (x_j - z) R
        if (dR[j] == 0u) dR[j] = BB_R1 | 0x80000000u;   // marker (never a canonical value): handled below
        pre[j] = acc;
        acc = mont_mul(acc, (dR[j] & 0x80000000u) ? BB_R1 : dR[j]);
        if (j + 1 < K) xR = mont_mul(xR, a.wNR);
    }
    uint32_t inv = mont_inv(acc);
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
        const bool zero = (dR[j] & 0x80000000u) != 0u;
        const uint32_t invR = zero ? 0u : mont_mul(inv, pre[j]);          // (x_j - z)^-1 R
        if (!zero) inv = mont_mul(inv, dR[j]);
        const uint32_t num = bb_add(bb_add(bb_sub(qv[j], a.q_z), bb_sub(t2[j], a.t_ggz)), bb_add(bb_sub(t1[j], a.t_gz), bb_sub(t0[j], a.t_z)));
        out[j] = mont_mul(num, invR);
    }
}

// ---- polynomial evaluation at up to POLY_MAX_POINTS points ----
constexpr int POLY_MAX_POINTS = 4;
constexpr uint32_t POLY_PER_THREAD = 16, POLY_THREADS = 256, POLY_CHUNK = POLY_PER_THREAD * POLY_THREADS;
struct PolyEvalArgs {
    const uint32_t* coeffs;
    uint64_t ncoeffs;
    uint32_t npoints;
    uint32_t zR[POLY_MAX_POINTS];        // points, Montgomery form
    uint32_t z16R[POLY_MAX_POINTS];      // z^16
    uint32_t zchunkR[POLY_MAX_POINTS];   // z^POLY_CHUNK
    uint32_t* partial;                   // [blocks][npoints]: sum over the block's chunk of c_i z^(i - chunk start)
    uint32_t* out;                       // [npoints]
    uint32_t nblocks;
};
// the 16 coefficients of one thread: sum_j c_j z^j by Horner (src/math/polynomial.rs:139-142 on a run of 16), then times z^(16 t)
TOYNI_HD uint32_t poly_thread_term(const PolyEvalArgs& a, uint32_t p, const uint32_t (&c)[POLY_PER_THREAD], uint32_t t) {
    uint32_t r = c[POLY_PER_THREAD - 1];
#pragma unroll
    for (int j = (int)POLY_PER_THREAD - 2; j >= 0; --j) r = bb_add(mont_mul(r, a.zR[p]), c[j]);
    return mont_mul(r, mont_pow(a.z16R[p], t));
}

// ---- Merkle openings (src/merkle.rs:50-80, src/fibonacci.rs:366-375) ----
// record of one opening: depth x 32 path bytes | 16 salt bytes (zero when unsalted) | value as 8 LE bytes | depth position bytes
// (1 = the sibling is the LEFT input of the node hash), padded to a multiple of 8
TOYNI_HD uint32_t merkle_depth(uint64_t n) {
    uint32_t d = 0;
    while (n > 1) { n = (n + 1) / 2; ++d; }
    return d;
}
TOYNI_HD uint64_t merkle_open_record_bytes(uint64_t n) {
    const uint32_t d = merkle_depth(n);
    return (uint64_t)d * 32u + 24u + ((d + 7u) & ~7u);
}
// sibling row (index into the flat level array) and position flag of level `level` for leaf `index`
TOYNI_HD uint64_t merkle_sibling_row(uint64_t n, uint64_t index, uint32_t level, bool& is_left) {
    uint64_t off = 0, m = n, cur = index;
    for (uint32_t l = 0; l < level; ++l) { off += m; m = (m + 1) / 2; cur >>= 1; }
    const uint64_t sib = (cur & 1u) ? cur - 1 : cur + 1;
    if (sib >= m) { is_left = true; return off + cur; }   // last node of an odd level: paired with itself (src/merkle.rs:67-70)
    is_left = (cur & 1u) != 0;
    return off + sib;
}


// ---- salts: a ChaCha20 keystream (RFC 8439 2.3: constants | key | 32-bit block counter | 96-bit nonce, 20 rounds) ----
// The reference salts every Merkle leaf with 16 bytes of `rand::thread_rng()` (src/fibonacci.rs:341-343), a ChaCha-based CSPRNG
// on the host; a device-resident prover wants them where the leaves are hashed instead of 130 MB over PCIe per proof.  One block
// = 64 bytes = the salts of four leaves.
TOYNI_HD uint32_t chacha_rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
TOYNI_HD void chacha20_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      counter, nonce[0], nonce[1], nonce[2]};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = s[i];
#define TOYNI_QR(a, b, c, d)                                      \
    x[a] += x[b]; x[d] = chacha_rotl(x[d] ^ x[a], 16);            \
    x[c] += x[d]; x[b] = chacha_rotl(x[b] ^ x[c], 12);            \
    x[a] += x[b]; x[d] = chacha_rotl(x[d] ^ x[a], 8);             \
    x[c] += x[d]; x[b] = chacha_rotl(x[b] ^ x[c], 7)
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        TOYNI_QR(0, 4, 8, 12); TOYNI_QR(1, 5, 9, 13); TOYNI_QR(2, 6, 10, 14); TOYNI_QR(3, 7, 11, 15);
        TOYNI_QR(0, 5, 10, 15); TOYNI_QR(1, 6, 11, 12); TOYNI_QR(2, 7, 8, 13); TOYNI_QR(3, 4, 9, 14);
    }
#undef TOYNI_QR
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];   // little-endian words = the keystream bytes on this (little-endian) target
}

}  // namespace toyni
