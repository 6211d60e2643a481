// SHA-256 Merkle commitment of a layer of field elements on gfx950.
//
// What it reproduces, byte for byte: build_merkle_tree / build_unsalted_tree (src/fibonacci.rs:340-361) over
// MerkleTree::build_tree (src/merkle.rs:25-48) with the domain-separated hashes of src/merkle.rs:105-123:
//   leaf  = SHA256(0x00 || salt[16] || value as 8 LE bytes)    (or 0x00 || value bytes when unsalted)  -> ONE block
//   node  = SHA256(0x01 || left[32] || right[32])              (65 bytes)                               -> TWO blocks
//   an odd level duplicates its last node.
// The reference builds this with a heap Vec<u8> per node on one core; here one thread hashes one leaf / one node,
// message words are assembled in registers (no byte buffers), digests are stored as the byte strings the reference holds.
// SHA-256 itself is FIPS 180-4 (the reference uses the sha2 0.10.8 crate).  Plain C++ so tests/emu can step it.
#pragma once
#include "bb_field.hpp"

namespace toyni {

TOYNI_HD uint32_t sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
// three-input XOR: one v_bitop3_b32 on gfx950 (the compiler emits two v_xor for a ^ b ^ c)
TOYNI_HD uint32_t sha_xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, 0x96);
#else
    return a ^ b ^ c;
#endif
}
// Ch(e, f, g) = (e & f) ^ (~e & g) and Maj(a, b, c) as ONE v_bitop3_b32 each (truth tables 0xCA / 0xE8 with the operands as
// 0xF0 / 0xCC / 0xAA): the compiler builds Maj from and / xor / bitop3 -- two instructions more per round, 255 of the 3 015 of a
// node hash, on a chain whose length IS the latency of a tree level.
TOYNI_HD uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_bitop3_b32((int)e, (int)f, (int)g, 0xCA);
#else
    return (e & f) ^ (~e & g);
#endif
}
TOYNI_HD uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, 0xE8);
#else
    return (a & b) ^ (a & c) ^ (b & c);
#endif
}
TOYNI_HD uint32_t sha_bswap(uint32_t x) { return (x >> 24) | ((x >> 8) & 0xFF00u) | ((x << 8) & 0xFF0000u) | (x << 24); }

struct Sha256State { uint32_t h[8]; };

TOYNI_HD Sha256State sha256_init() {
    return Sha256State{{0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u}};
}

// one compression; w[0..15] = the block as big-endian words (consumed: the schedule is expanded in place)
TOYNI_HD void sha256_compress(Sha256State& st, uint32_t (&w)[16]) {
    constexpr uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
        0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
        0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
        0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
        0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
        0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t a = st.h[0], b = st.h[1], c = st.h[2], d = st.h[3], e = st.h[4], f = st.h[5], g = st.h[6], h = st.h[7];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = sha_xor3(sha_rotr(w15, 7), sha_rotr(w15, 18), w15 >> 3);
            const uint32_t s1 = sha_xor3(sha_rotr(w2, 17), sha_rotr(w2, 19), w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const uint32_t t1 = h + sha_xor3(sha_rotr(e, 6), sha_rotr(e, 11), sha_rotr(e, 25)) + sha_ch(e, f, g) + K[i] + w[i & 15];
        const uint32_t t2 = sha_xor3(sha_rotr(a, 2), sha_rotr(a, 13), sha_rotr(a, 22)) + sha_maj(a, b, c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st.h[0] += a; st.h[1] += b; st.h[2] += c; st.h[3] += d; st.h[4] += e; st.h[5] += f; st.h[6] += g; st.h[7] += h;
}

// A digest in memory is the reference's 32-byte string; as 8 little-endian u32 loads that is bswap of the state words.
struct Digest { uint32_t m[8]; };  // memory words (little-endian loads of the byte string)
TOYNI_HD Digest digest_of(const Sha256State& st) {
    Digest d;
#pragma unroll
    for (int j = 0; j < 8; ++j) d.m[j] = sha_bswap(st.h[j]);
    return d;
}

// leaf of a canonical residue `value` (its 8-byte LE encoding is value, 0,0,0,0): salted (salt = 4 memory words) or not
TOYNI_HD Digest merkle_leaf(uint32_t value, const uint32_t* salt_words /* 4 LE words or nullptr */) {
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = 0u;
    const uint32_t vb = sha_bswap(value);  // v0 v1 v2 v3 as a big-endian word
    if (salt_words) {
        // bytes: 00 | s0..s15 | v0..v3 00 00 00 00 | 80 ...   (25 bytes -> length 200 bits)
        uint32_t s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = sha_bswap(salt_words[j]);  // big-endian view of salt bytes 4j..4j+3
        w[0] = s[0] >> 8;                       // 00 s0 s1 s2
        w[1] = (s[0] << 24) | (s[1] >> 8);      // s3 s4 s5 s6
        w[2] = (s[1] << 24) | (s[2] >> 8);
        w[3] = (s[2] << 24) | (s[3] >> 8);
        w[4] = (s[3] << 24) | (vb >> 8);        // s15 v0 v1 v2
        w[5] = vb << 24;                        // v3 00 00 00
        w[6] = 0x00800000u;                     // 00 80 00 00  (byte 24 = last zero value byte, byte 25 = 0x80)
        w[15] = 200u;
    } else {
        // bytes: 00 | v0..v3 00 00 00 00 | 80 ...            (9 bytes -> 72 bits)
        w[0] = vb >> 8;                         // 00 v0 v1 v2
        w[1] = vb << 24;                        // v3 00 00 00
        w[2] = 0x00800000u;                     // 00 80 00 00
        w[15] = 72u;
    }
    Sha256State st = sha256_init();
    sha256_compress(st, w);
    return digest_of(st);
}

// node = SHA256(01 || left || right): 65 bytes, two blocks
TOYNI_HD Digest merkle_node(const Digest& left, const Digest& right) {
    uint32_t d[16];  // big-endian words of left || right
#pragma unroll
    for (int j = 0; j < 8; ++j) { d[j] = sha_bswap(left.m[j]); d[8 + j] = sha_bswap(right.m[j]); }
    uint32_t w[16];
    w[0] = 0x01000000u | (d[0] >> 8);
#pragma unroll
    for (int j = 1; j < 16; ++j) w[j] = (d[j - 1] << 24) | (d[j] >> 8);
    Sha256State st = sha256_init();
    sha256_compress(st, w);
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = 0u;
    w[0] = (d[15] << 24) | 0x00800000u;         // last byte of right, then 0x80
    w[15] = 520u;                               // 65 bytes
    sha256_compress(st, w);
    return digest_of(st);
}


// ---- a node hash on TWO waves (round 3) ------------------------------------------------------------------------------------
// A tree level with few nodes is pure latency: one thread's node hash is ~3 000 dependent-ish instructions, and a lone wave issues one
// instruction per 5 cycles whatever the other 255 CUs do -- 6.5 us per level, 200-odd levels per proof (a 2^16-row Fibonacci proof
// spends 83 % of its time in these kernels).  What CAN be taken off the critical wave is everything that does not depend on the
// running state:
//   * block 1's message schedule W[16..63] (10 of the 26 instructions of a round): a HELPER wave on another SIMD computes it from
//     the same 16 message words and hands K[i] + W[i] over through LDS, a third of a block ahead of the rounds that consume it;
//   * block 2's schedule: its message is one data byte (the last byte of the right child) followed by constants, so K[i] + W[i] is a
//     table of 64 x 256 words (SHA_B2, built at compile time), gathered by the main wave while block 1 runs.
// The main wave is left with the 128 round functions: ~2 100 instructions instead of ~3 000.  Same digests, bit for bit (tests/emu
// steps the phase bodies below against merkle_node; the GPU tests compare whole trees with hashlib).
// Phases (a workgroup-wide barrier between them; both roles execute every barrier):
//   A  main: rounds 0..15 of block 1        helper: K + W[16..39] -> LDS
//   B  main: rounds 16..39 (W from LDS)     helper: K + W[40..63] -> LDS
//   C  main: rounds 40..63, block 2 from the table, digest
constexpr uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// K[i] + W[i] of a node hash's SECOND block for each value of its one data byte: block = byte 80 00 .. 00 | length 520 bits
struct ShaB2Table { uint32_t kw[64][256]; };
constexpr uint32_t cx_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
constexpr ShaB2Table make_sha_b2_table() {
    ShaB2Table t{};
    for (uint32_t byte = 0; byte < 256; ++byte) {
        uint32_t w[64] = {};
        w[0] = (byte << 24) | 0x00800000u;
        w[15] = 520u;
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = cx_rotr(w[i - 15], 7) ^ cx_rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = cx_rotr(w[i - 2], 17) ^ cx_rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        for (int i = 0; i < 64; ++i) t.kw[i][byte] = SHA_K[i] + w[i];
    }
    return t;
}

struct ShaRegs { uint32_t a, b, c, d, e, f, g, h; };
TOYNI_HD void sha_round(ShaRegs& s, uint32_t kw) {
    const uint32_t t1 = s.h + sha_xor3(sha_rotr(s.e, 6), sha_rotr(s.e, 11), sha_rotr(s.e, 25)) + sha_ch(s.e, s.f, s.g) + kw;
    const uint32_t t2 = sha_xor3(sha_rotr(s.a, 2), sha_rotr(s.a, 13), sha_rotr(s.a, 22)) + sha_maj(s.a, s.b, s.c);
    s.h = s.g; s.g = s.f; s.f = s.e; s.e = s.d + t1; s.d = s.c; s.c = s.b; s.b = s.a; s.a = t1 + t2;
}
// the 16 message words of a node's first block: 01 || left || right[0..30]; `last` = the byte left over for block 2
TOYNI_HD void node_block1(const Digest& left, const Digest& right, uint32_t (&w)[16], uint32_t& last) {
    uint32_t d[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) { d[j] = sha_bswap(left.m[j]); d[8 + j] = sha_bswap(right.m[j]); }
    w[0] = 0x01000000u | (d[0] >> 8);
#pragma unroll
    for (int j = 1; j < 16; ++j) w[j] = (d[j - 1] << 24) | (d[j] >> 8);
    last = d[15] & 0xFFu;
}
// LDS hand-over area of one main / helper pair: word (i - 16) * 64 + lane holds K[i] + W[i] of block 1, i = 16 .. 63
constexpr uint32_t COOP_SCHED_WORDS = 48u * 64u;

// helper, phases A (FIRST = 16, COUNT = 24) and B (FIRST = 40): W ring in registers across the two calls
template <int FIRST, int COUNT>
TOYNI_HD void coop_helper_schedule(uint32_t (&w)[16], uint32_t* sched, uint32_t lane) {
#pragma unroll
    for (int i = FIRST; i < FIRST + COUNT; ++i) {
        const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
        const uint32_t s0 = sha_xor3(sha_rotr(w15, 7), sha_rotr(w15, 18), w15 >> 3);
        const uint32_t s1 = sha_xor3(sha_rotr(w2, 17), sha_rotr(w2, 19), w2 >> 10);
        w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        sched[(uint32_t)(i - 16) * 64u + lane] = SHA_K[i] + w[i & 15];
    }
}
// main, phase A: rounds 0 .. 15 straight from the message words
TOYNI_HD void coop_main_first16(ShaRegs& s, const uint32_t (&w)[16]) {
    const Sha256State iv = sha256_init();
    s = ShaRegs{iv.h[0], iv.h[1], iv.h[2], iv.h[3], iv.h[4], iv.h[5], iv.h[6], iv.h[7]};
#pragma unroll
    for (int i = 0; i < 16; ++i) sha_round(s, SHA_K[i] + w[i]);
}
// main, phases B (FIRST = 16) and C (FIRST = 40): 24 rounds each, K + W from the pair's LDS area
template <int FIRST>
TOYNI_HD void coop_main_rounds(ShaRegs& s, const uint32_t* sched, uint32_t lane) {
    uint32_t kw[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) kw[i] = sched[(uint32_t)(FIRST - 16 + i) * 64u + lane];   // all issued up front: one LDS latency, not 24
#pragma unroll
    for (int i = 0; i < 24; ++i) sha_round(s, kw[i]);
}
// main, end of phase C: close block 1, run block 2 from the gathered table words, return the digest
TOYNI_HD Digest coop_main_finish(const ShaRegs& s1, const uint32_t (&kw2)[64]) {
    Sha256State st = sha256_init();
    st.h[0] += s1.a; st.h[1] += s1.b; st.h[2] += s1.c; st.h[3] += s1.d; st.h[4] += s1.e; st.h[5] += s1.f; st.h[6] += s1.g; st.h[7] += s1.h;
    ShaRegs s{st.h[0], st.h[1], st.h[2], st.h[3], st.h[4], st.h[5], st.h[6], st.h[7]};
#pragma unroll
    for (int i = 0; i < 64; ++i) sha_round(s, kw2[i]);
    st.h[0] += s.a; st.h[1] += s.b; st.h[2] += s.c; st.h[3] += s.d; st.h[4] += s.e; st.h[5] += s.f; st.h[6] += s.g; st.h[7] += s.h;
    return digest_of(st);
}

}  // namespace toyni
