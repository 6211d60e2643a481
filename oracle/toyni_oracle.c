/*
 * toyni_oracle.c -- CPU restatement of the reference's BabyBear NTT / FRI-fold path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under toyni_amd/ may link, import or call this
 * file.  It is used by tests/, by __graft_entry__.smoke() as the checker, and by
 * bench.py's `cpu_baseline` leg ("kind": "port", 1 thread) -- never as the thing shipped.
 *
 * Parity status: PINNED.  The reference (Rust + CUDA) cannot be compiled in this image
 * (no cargo/rustc/nvcc -- SURVEY.md F6), so the pin is (a) every known-answer test the
 * reference's own test-suite holds for this path (src/ntt.rs:321-379, src/babybear.rs:219-284,
 * src/math/domain.rs:192-242) re-run against this file in tests/test_oracle.py, and (b) golden
 * vectors produced by an independent big-integer O(n^2) DFT model (tests/golden/gen_golden.py).
 *
 * Every function cites the reference file:line it restates.  The loop structure follows the
 * reference on purpose (same algorithm => meaningful CPU timing baseline): multiplication is
 * `unsigned __int128 %` exactly as src/babybear.rs:173-176.
 *
 * Element layout: one uint64_t per field element, canonical residue in [0,p)
 * (#[repr(C)] struct BabyBear { value: u64 } -- src/babybear.rs:10-14).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define BB_P 2013265921ULL /* src/babybear.rs:8  p = 2^31 - 2^27 + 1 */

/* src/babybear.rs:26-30  BabyBear::new */
uint64_t orc_bb_new(uint64_t v) { return v % BB_P; }

/* src/babybear.rs:80-89 reduce_wide + :129-139 Add */
uint64_t orc_bb_add(uint64_t a, uint64_t b) {
    uint64_t v = a + b;
    if (v >= BB_P) v -= BB_P;
    if (v >= BB_P) v -= BB_P;
    return v;
}

/* src/babybear.rs:148-160 Sub */
uint64_t orc_bb_sub(uint64_t a, uint64_t b) {
    return a >= b ? a - b : a + BB_P - b;
}

/* src/babybear.rs:169-178 Mul: (a as u128 * b as u128) % p */
uint64_t orc_bb_mul(uint64_t a, uint64_t b) {
    unsigned __int128 prod = (unsigned __int128)a * (unsigned __int128)b;
    return (uint64_t)(prod % (unsigned __int128)BB_P);
}

/* src/babybear.rs:195-207 Neg */
uint64_t orc_bb_neg(uint64_t a) { return a == 0 ? 0 : BB_P - a; }

/* src/babybear.rs:91-108 pow (square-and-multiply, LSB first) */
uint64_t orc_bb_pow(uint64_t base, uint64_t exp) {
    if (exp == 0) return 1;
    uint64_t result = 1;
    while (exp > 0) {
        if (exp & 1) result = orc_bb_mul(result, base);
        base = orc_bb_mul(base, base);
        exp >>= 1;
    }
    return result;
}

/* src/babybear.rs:111-114 inverse = a^(p-2); caller guarantees a != 0 (reference asserts) */
uint64_t orc_bb_inverse(uint64_t a) { return orc_bb_pow(a, BB_P - 2); }

/* src/babybear.rs:187-193 Div */
uint64_t orc_bb_div(uint64_t a, uint64_t b) { return orc_bb_mul(a, orc_bb_inverse(b)); }

/* src/babybear.rs:118-126 get_root_of_unity: 440564289^(2^(27-log_n)); log_n <= 27.
 * Returns 0 for log_n > 27 (the reference panics). */
uint64_t orc_bb_root_of_unity(uint32_t log_n) {
    if (log_n > 27) return 0;
    return orc_bb_pow(orc_bb_new(440564289ULL), 1ULL << (27 - log_n));
}

/* src/ntt.rs:14-21 bit_reverse */
static size_t orc_bit_reverse(size_t x, unsigned log_n) {
    size_t r = 0;
    for (unsigned i = 0; i < log_n; ++i) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

static unsigned orc_log2(size_t n) {
    unsigned l = 0;
    while (((size_t)1 << l) < n) ++l;
    return l;
}

/* src/ntt.rs:24-53 ntt: in-place radix-2 DIT Cooley-Tukey, natural order in and out.
 * Returns -1 if n is not a power of two (the reference asserts). */
int orc_ntt(uint64_t *values, size_t n, uint64_t omega) {
    if (n == 0 || (n & (n - 1))) return -1;
    unsigned log_n = orc_log2(n);

    for (size_t i = 0; i < n; ++i) {          /* :29-34 */
        size_t j = orc_bit_reverse(i, log_n);
        if (i < j) {
            uint64_t t = values[i];
            values[i] = values[j];
            values[j] = t;
        }
    }

    for (size_t len = 2; len <= n; len *= 2) { /* :36-52 */
        size_t step = n / len;
        uint64_t w_len = orc_bb_pow(omega, (uint64_t)step);
        for (size_t i = 0; i < n; i += len) {
            uint64_t w = 1;
            for (size_t j = 0; j < len / 2; ++j) {
                uint64_t u = values[i + j];
                uint64_t v = orc_bb_mul(values[i + j + len / 2], w);
                values[i + j] = orc_bb_add(u, v);
                values[i + j + len / 2] = orc_bb_sub(u, v);
                w = orc_bb_mul(w, w_len);
            }
        }
    }
    return 0;
}

/* src/ntt.rs:56-66 intt: ntt with omega^(n-1), then every element * n^-1 */
int orc_intt(uint64_t *values, size_t n, uint64_t omega) {
    uint64_t inv_omega = orc_bb_pow(omega, (uint64_t)n - 1);
    int rc = orc_ntt(values, n, inv_omega);
    if (rc) return rc;
    uint64_t inv_n = orc_bb_inverse(orc_bb_new((uint64_t)n));
    for (size_t i = 0; i < n; ++i) values[i] = orc_bb_mul(values[i], inv_n);
    return 0;
}

/* What src/ntt.rs:224-251 (ntt_cuda / intt_cuda) compute: the transform with the canonical
 * root get_root_of_unity(log2 n) -- they take no omega argument. */
int orc_ntt_canonical(uint64_t *values, size_t n) {
    if (n == 0 || (n & (n - 1)) || orc_log2(n) > 27) return -1;
    return orc_ntt(values, n, orc_bb_root_of_unity(orc_log2(n)));
}
int orc_intt_canonical(uint64_t *values, size_t n) {
    if (n == 0 || (n & (n - 1)) || orc_log2(n) > 27) return -1;
    return orc_intt(values, n, orc_bb_root_of_unity(orc_log2(n)));
}

/* src/ntt.rs:69-81 roots_of_unity_domain */
int orc_roots_of_unity_domain(uint64_t *out, size_t n) {
    if (n == 0 || (n & (n - 1)) || orc_log2(n) > 27) return -1;
    uint64_t omega = orc_bb_root_of_unity(orc_log2(n));
    uint64_t cur = 1;
    for (size_t i = 0; i < n; ++i) {
        out[i] = cur;
        cur = orc_bb_mul(cur, omega);
    }
    return 0;
}

/* src/math/domain.rs:61-69 BabyBearDomain::elements: shift * omega^i */
int orc_domain_elements(uint64_t *out, size_t n, uint64_t shift) {
    if (n == 0 || (n & (n - 1)) || orc_log2(n) > 27) return -1;
    uint64_t omega = orc_bb_root_of_unity(orc_log2(n));
    uint64_t cur = shift;
    for (size_t i = 0; i < n; ++i) {
        out[i] = cur;
        cur = orc_bb_mul(cur, omega);
    }
    return 0;
}

/* src/math/domain.rs:154-162 apply_coset_shift: values[i] *= shift^i (skipped when shift == 1) */
static void orc_apply_coset_shift(uint64_t *v, size_t n, uint64_t shift) {
    if (shift != 1) {
        uint64_t sp = 1;
        for (size_t i = 0; i < n; ++i) {
            v[i] = orc_bb_mul(v[i], sp);
            sp = orc_bb_mul(sp, shift);
        }
    }
}

/* src/math/domain.rs:165-174 undo_coset_shift: values[i] *= shift^-i */
static void orc_undo_coset_shift(uint64_t *v, size_t n, uint64_t shift) {
    if (shift != 1) {
        uint64_t sinv = orc_bb_inverse(shift);
        uint64_t sp = 1;
        for (size_t i = 0; i < n; ++i) {
            v[i] = orc_bb_mul(v[i], sp);
            sp = orc_bb_mul(sp, sinv);
        }
    }
}

/* src/math/domain.rs:107-123 BabyBearDomain::fft: zero-pad coeffs to `size`, coset pre-scale, NTT.
 * `out` has `size` elements; ncoeffs <= size. */
int orc_domain_fft(uint64_t *out, size_t size, const uint64_t *coeffs, size_t ncoeffs, uint64_t shift) {
    if (ncoeffs > size) return -1;
    memcpy(out, coeffs, ncoeffs * sizeof(uint64_t));
    memset(out + ncoeffs, 0, (size - ncoeffs) * sizeof(uint64_t));
    orc_apply_coset_shift(out, size, shift);
    return orc_ntt_canonical(out, size);
}

/* src/math/domain.rs:85-102 BabyBearDomain::ifft: INTT then coset post-scale. In place on `values`. */
int orc_domain_ifft(uint64_t *values, size_t size, uint64_t shift) {
    int rc = orc_intt_canonical(values, size);
    if (rc) return rc;
    orc_undo_coset_shift(values, size, shift);
    return 0;
}

/* src/math/fri.rs:27-48 fri_fold: out[i] = (a+b)/2 + (a-b)/2 * beta / xs[i],
 * a = evals[i], b = evals[i+half]; only xs[0..half) is read.  One Fermat inversion per element
 * exactly as the reference (x.inverse() at :42).  Returns -1 on odd length (reference asserts). */
int orc_fri_fold(uint64_t *out, const uint64_t *evals, size_t len, const uint64_t *xs, uint64_t beta) {
    if (len % 2) return -1;
    size_t half = len / 2;
    uint64_t half_inv = orc_bb_inverse(orc_bb_new(2));
    for (size_t i = 0; i < half; ++i) {
        uint64_t a = evals[i];
        uint64_t b = evals[i + half];
        uint64_t x = xs[i];
        uint64_t avg = orc_bb_mul(orc_bb_add(a, b), half_inv);
        uint64_t diff = orc_bb_mul(orc_bb_sub(a, b), half_inv);
        /* avg + diff * beta * x.inverse()  -- left-to-right as written at :42 */
        out[i] = orc_bb_add(avg, orc_bb_mul(orc_bb_mul(diff, beta), orc_bb_inverse(x)));
    }
    return 0;
}

/* The prover's use of fri_fold (src/fibonacci.rs:214,220-245): layer 0 points are the coset
 * elements shift*omega_N^i; after each fold xs is truncated to the folded length and squared.
 * This helper reproduces that loop for `nfolds` layers on a codeword of size N and writes every
 * folded layer back to back into `layers_out` (sizes N/2, N/4, ...).  betas[k] is the challenge
 * of fold k. */
int orc_fri_fold_layers(uint64_t *layers_out, const uint64_t *evals, size_t N, uint64_t shift,
                        const uint64_t *betas, unsigned nfolds) {
    if (N == 0 || (N & (N - 1))) return -1;
    uint64_t *xs = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t *cur = (uint64_t *)malloc(N * sizeof(uint64_t));
    if (!xs || !cur) { free(xs); free(cur); return -2; }
    orc_domain_elements(xs, N, shift);
    memcpy(cur, evals, N * sizeof(uint64_t));
    size_t len = N;
    uint64_t *dst = layers_out;
    for (unsigned k = 0; k < nfolds && len >= 2; ++k) {
        orc_fri_fold(dst, cur, len, xs, betas[k]);
        size_t half = len / 2;
        for (size_t i = 0; i < half; ++i) xs[i] = orc_bb_mul(xs[i], xs[i]); /* :228-231 */
        memcpy(cur, dst, half * sizeof(uint64_t));
        dst += half;
        len = half;
    }
    free(xs);
    free(cur);
    return 0;
}

/* ---- Ext = F_p[X]/(X^4 - 11), src/ext.rs -- only what fri_fold_ext needs (row a20, "next") ---- */

/* src/ext.rs:178-192 schoolbook multiply with X^4 = 11 */
static void orc_ext_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            t[i + j] = orc_bb_add(t[i + j], orc_bb_mul(a[i], b[j]));
    for (int k = 0; k < 4; ++k) {
        uint64_t v = t[k];
        if (k + 4 < 7) v = orc_bb_add(v, orc_bb_mul(t[k + 4], 11));
        r[k] = v;
    }
}

/* src/math/fri.rs:7-25 fri_fold_ext: Ext values and beta, base-field xs. evals/out are AoS 4xu64. */
int orc_fri_fold_ext(uint64_t *out, const uint64_t *evals, size_t len, const uint64_t *xs, const uint64_t beta[4]) {
    if (len % 2) return -1;
    size_t half = len / 2;
    uint64_t half_inv = orc_bb_inverse(orc_bb_new(2));
    for (size_t i = 0; i < half; ++i) {
        const uint64_t *a = evals + 4 * i;
        const uint64_t *b = evals + 4 * (i + half);
        uint64_t x_inv = orc_bb_inverse(xs[i]);
        uint64_t avg[4], diff[4], db[4], xe[4] = {x_inv, 0, 0, 0}, prod[4];
        for (int k = 0; k < 4; ++k) {
            avg[k] = orc_bb_mul(orc_bb_add(a[k], b[k]), half_inv);
            diff[k] = orc_bb_mul(orc_bb_sub(a[k], b[k]), half_inv);
        }
        orc_ext_mul(db, diff, beta);   /* diff * beta */
        orc_ext_mul(prod, db, xe);     /* * Ext::from_base(x_inv) */
        for (int k = 0; k < 4; ++k) out[4 * i + k] = orc_bb_add(avg[k], prod[k]);
    }
    return 0;
}

/* ---- synthetic inputs shared by tests and bench ---- */

/* I1: x[i] = (7*i + 3) mod p -- the reference's own test pattern, src/ntt.rs:272 */
void orc_fill_pattern_7i3(uint64_t *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = orc_bb_new((uint64_t)i * 7 + 3);
}

static uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* I2: x[i] = splitmix64(seed + i) mod p */
void orc_fill_splitmix(uint64_t *out, size_t n, uint64_t seed) {
    for (size_t i = 0; i < n; ++i) out[i] = orc_splitmix64(seed + (uint64_t)i) % BB_P;
}

/* ================================================================================================
 * Merkle commitment of a layer of field elements (SURVEY.md 8(f) rank 2 -- "next" row).
 * SHA-256 is FIPS 180-4 (the reference takes it from the `sha2 0.10.8` crate, Cargo.toml:13; not vendored
 * in /root/reference, so the published algorithm is restated here and pinned against Python's hashlib in
 * tests/test_oracle.py).  Tree shape, tags and leaf format follow the reference's own code.
 * ============================================================================================== */
static const uint32_t ORC_K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static uint32_t orc_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void orc_sha256_block(uint32_t st[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = orc_rotr(w[i - 15], 7) ^ orc_rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = orc_rotr(w[i - 2], 17) ^ orc_rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t t1 = h + (orc_rotr(e, 6) ^ orc_rotr(e, 11) ^ orc_rotr(e, 25)) + ((e & f) ^ (~e & g)) + ORC_K256[i] + w[i];
        uint32_t t2 = (orc_rotr(a, 2) ^ orc_rotr(a, 13) ^ orc_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

void orc_sha256(uint8_t out[32], const uint8_t *msg, size_t len) {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t off = 0;
    for (; off + 64 <= len; off += 64) orc_sha256_block(st, msg + off);
    uint8_t tail[128];
    size_t rem = len - off;
    memset(tail, 0, sizeof tail);
    memcpy(tail, msg + off, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    orc_sha256_block(st, tail);
    if (tl == 128) orc_sha256_block(st, tail + 64);
    for (int i = 0; i < 8; ++i) { out[4 * i] = st[i] >> 24; out[4 * i + 1] = st[i] >> 16; out[4 * i + 2] = st[i] >> 8; out[4 * i + 3] = st[i]; }
}

/* src/merkle.rs:105-114 hash_leaf: SHA256(0x00 || leaf) */
void orc_hash_leaf(uint8_t out[32], const uint8_t *data, size_t len) {
    uint8_t *buf = (uint8_t *)malloc(len + 1);
    buf[0] = 0x00;
    memcpy(buf + 1, data, len);
    orc_sha256(out, buf, len + 1);
    free(buf);
}

/* src/merkle.rs:116-123 hash_node: SHA256(0x01 || left || right) */
void orc_hash_node(uint8_t out[32], const uint8_t left[32], const uint8_t right[32]) {
    uint8_t buf[65];
    buf[0] = 0x01;
    memcpy(buf + 1, left, 32);
    memcpy(buf + 33, right, 32);
    orc_sha256(out, buf, 65);
}

/* number of digests in all levels of a tree over n leaves (src/merkle.rs:25-48: halve, rounding up, until 1) */
size_t orc_merkle_total_digests(size_t n) {
    size_t total = 0;
    if (n == 0) return 0;
    for (;;) { total += n; if (n == 1) break; n = (n + 1) / 2; }
    return total;
}

/* src/merkle.rs:25-48 build_tree over already-hashed leaves stored at levels[0 .. n); appends every upper level.
 * An odd level duplicates its last node (:38-42). */
static void orc_merkle_build_upper(uint8_t *levels, size_t n) {
    uint8_t *cur = levels;
    while (n > 1) {
        uint8_t *next = cur + 32 * n;
        size_t m = (n + 1) / 2;
        for (size_t i = 0; i < m; ++i) {
            const uint8_t *l = cur + 32 * (2 * i);
            const uint8_t *r = (2 * i + 1 < n) ? cur + 32 * (2 * i + 1) : l;
            orc_hash_node(next + 32 * i, l, r);
        }
        cur = next;
        n = m;
    }
}

/* MerkleTree::new(leaves) for byte-string leaves of a common length: all levels back to back, leaf hashes first */
void orc_merkle_levels(uint8_t *levels, const uint8_t *leaves, size_t leaf_len, size_t n) {
    for (size_t i = 0; i < n; ++i) orc_hash_leaf(levels + 32 * i, leaves + leaf_len * i, leaf_len);
    orc_merkle_build_upper(levels, n);
}

/* build_merkle_tree / build_unsalted_tree (src/fibonacci.rs:340-361): leaf = salt[16] || value.to_bytes() (8-byte LE,
 * src/babybear.rs:53-55), or just the 8 value bytes when salts == NULL. */
void orc_merkle_commit_values(uint8_t *levels, const uint64_t *values, const uint8_t *salts, size_t n) {
    uint8_t leaf[24];
    for (size_t i = 0; i < n; ++i) {
        size_t len = 0;
        if (salts) { memcpy(leaf, salts + 16 * i, 16); len = 16; }
        for (int b = 0; b < 8; ++b) leaf[len + b] = (uint8_t)(values[i] >> (8 * b));
        orc_hash_leaf(levels + 32 * i, leaf, len + 8);
    }
    orc_merkle_build_upper(levels, n);
}

/* ---- pointwise steps of the Fibonacci prover on the LDE coset ----------------------------
 * Restated on EVALUATIONS: with lde_domain = shift * <w_N>, g = w_n = w_N^B (B = N / n), the reference's
 * trace_poly.evaluate(g * x_i) (src/fibonacci.rs:137-141) is the LDE value at position (i + B) mod N -- the same
 * identity the reference itself uses for its query openings (:261-266).  Everything else is the reference's formula
 * with its own operation order (one division per term, Fermat inverses). */

/* src/math/polynomial.rs:134-144  Polynomial::evaluate (Horner) */
uint64_t orc_poly_eval(const uint64_t *coeffs, size_t n, uint64_t x) {
    if (n == 0) return 0;
    uint64_t result = coeffs[n - 1] % BB_P;
    for (size_t k = n - 1; k-- > 0;) result = orc_bb_add(orc_bb_mul(result, x), coeffs[k] % BB_P);
    return result;
}

/* src/fibonacci.rs:133-150: c_evals and q_evals over the coset; fibonacci_constraint :313-315,
 * boundary_constraint_1/2 :318-324, z_poly = x^n - 1 (vanishing_poly_coeffs).  c_out may be NULL. */
int orc_fib_quotient(uint64_t *c_out, uint64_t *q_out, const uint64_t *trace_lde, size_t N, size_t n, uint64_t shift) {
    if (N == 0 || (N & (N - 1)) || n < 2 || (n & (n - 1)) || n > N) return -1;
    const size_t B = N / n;
    uint64_t *xs = (uint64_t *)malloc(N * sizeof(uint64_t));
    if (!xs) return -2;
    orc_domain_elements(xs, N, shift);
    const uint64_t g = orc_bb_root_of_unity(orc_log2(n));
    const uint64_t b1 = orc_bb_pow(g, (uint64_t)(n - 1)), b2 = orc_bb_pow(g, (uint64_t)(n - 2));
    for (size_t i = 0; i < N; ++i) {
        const uint64_t x = xs[i];
        const uint64_t t0 = trace_lde[i], t1 = trace_lde[(i + B) % N], t2 = trace_lde[(i + 2 * B) % N];
        const uint64_t fib = orc_bb_sub(t2, orc_bb_add(t1, t0));
        const uint64_t c = orc_bb_mul(orc_bb_mul(fib, orc_bb_sub(x, b1)), orc_bb_sub(x, b2));
        const uint64_t zh = orc_bb_sub(orc_bb_pow(x, (uint64_t)n), 1);
        if (zh == 0) { free(xs); return -3; } /* "Cannot invert zero" */
        if (c_out) c_out[i] = c;
        q_out[i] = orc_bb_div(c, zh);
    }
    free(xs);
    return 0;
}

/* src/fibonacci.rs:186-198: the DEEP layer, four quotients summed in the reference's order */
int orc_fib_deep(uint64_t *out, const uint64_t *trace_lde, const uint64_t *q_evals, size_t N, size_t n, uint64_t shift,
                 uint64_t z, uint64_t t_z, uint64_t t_gz, uint64_t t_ggz, uint64_t q_z) {
    if (N == 0 || (N & (N - 1)) || n == 0 || n > N) return -1;
    const size_t B = N / n;
    uint64_t *xs = (uint64_t *)malloc(N * sizeof(uint64_t));
    if (!xs) return -2;
    orc_domain_elements(xs, N, shift);
    for (size_t i = 0; i < N; ++i) {
        const uint64_t d = orc_bb_sub(xs[i], z);
        if (d == 0) { free(xs); return -3; }
        const uint64_t t0 = trace_lde[i], t1 = trace_lde[(i + B) % N], t2 = trace_lde[(i + 2 * B) % N];
        uint64_t acc = orc_bb_div(orc_bb_sub(q_evals[i], q_z), d);
        acc = orc_bb_add(acc, orc_bb_div(orc_bb_sub(t2, t_ggz), d));
        acc = orc_bb_add(acc, orc_bb_div(orc_bb_sub(t1, t_gz), d));
        acc = orc_bb_add(acc, orc_bb_div(orc_bb_sub(t0, t_z), d));
        out[i] = acc;
    }
    free(xs);
    return 0;
}

/* src/merkle.rs:50-80 MerkleTree::get_proof on the flat level array of orc_merkle_levels: writes depth sibling digests
 * to `path` and depth flags to `position` (1 = "is_right" in the reference's naming: the sibling is hashed on the left,
 * :87-99); returns depth, or -1 when index >= n. */
int orc_merkle_get_proof(uint8_t *path, uint8_t *position, const uint8_t *levels, size_t n, size_t index) {
    if (index >= n) return -1;
    size_t off = 0, m = n, cur = index;
    int depth = 0;
    while (m > 1) {
        const size_t sib = (cur % 2 == 0) ? cur + 1 : cur - 1;
        if (sib >= m) {
            memcpy(path + 32 * (size_t)depth, levels + 32 * (off + cur), 32);
            position[depth] = 1;
        } else {
            memcpy(path + 32 * (size_t)depth, levels + 32 * (off + sib), 32);
            position[depth] = (uint8_t)(cur % 2 == 1);
        }
        off += m;
        m = (m + 1) / 2;
        cur /= 2;
        ++depth;
    }
    return depth;
}
