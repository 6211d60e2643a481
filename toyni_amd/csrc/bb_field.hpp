// BabyBear arithmetic for the gfx950 kernels: p = 2^31 - 2^27 + 1.
//
// Data elements are canonical residues in packed u32 (the reference's element is a u64 holding the
// same residue: src/babybear.rs:10-14; the u64 <-> u32 edge is toyni_hip.hip's narrow/widen).
// Twiddles are stored in Montgomery form (w * 2^32 mod p) so that one product
//     mont_mul(x, wR) = x * w mod p
// costs two 32x32->64 multiply-adds (v_mad_u64_u32) and one 32-bit multiply, with a canonical
// result after one conditional subtract.  Every function returns the exact canonical residue the
// reference's `u128 % p` multiply (src/babybear.rs:169-178) and Barrett multiply
// (cuda/ntt_kernel.cu:49-67) return; bb_mul_barrett64 below is the literal 64-bit Barrett of the
// north-star text and is kept as the cross-check variant (tests/emu compares all of them).
//
// The file is plain C++ so that tests/emu can compile the kernel bodies with g++ and step them on
// the CPU (test infrastructure; the shipped library contains device code only).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TOYNI_HD __host__ __device__ __forceinline__
#define TOYNI_DEV __device__ __forceinline__
#else
#define TOYNI_HD inline
#define TOYNI_DEV inline
#endif

// a value that is the same in every lane: on the device force it into an SGPR
#if defined(__HIP_DEVICE_COMPILE__)
#define TOYNI_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define TOYNI_UNIFORM(x) (x)
#endif

// scheduling fences (device only): TOYNI_SCHED_FENCE stops the instruction scheduler from moving anything across;
// TOYNI_PIN(v) makes v opaque at this point so that a running product is not unrolled into many live values
#if defined(__HIP_DEVICE_COMPILE__)
#define TOYNI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define TOYNI_PIN(v) asm volatile("" : "+v"(v))
// s_waitcnt vmcnt(0) alone (expcnt / lgkmcnt fields at their maxima), as a builtin so that the compiler's own
// wait-count insertion knows every earlier VMEM operation has retired
#define TOYNI_WAIT_VMEM0() __builtin_amdgcn_s_waitcnt(0x0F70)
// s_waitcnt vmcnt(N): everything but the N youngest VMEM operations has retired (N <= 63, compile-time)
#define TOYNI_WAIT_VMEM_ALLOW(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 0xF) | (((N) >> 4) << 14))
// Workgroup barrier that orders LDS traffic only.  Written as inline asm with a "memory" clobber so that the COMPILER keeps
// every LDS access on its side of the barrier (the s_barrier builtin is IntrNoMem: nothing at IR level would order the next
// tile's LDS stores against it); no vmcnt drain, unlike __syncthreads().  TOYNI_LDS_BARRIER first waits for this wave's own
// LDS operations (lgkmcnt), i.e. "my writes have landed"; TOYNI_BARRIER is the bare rendezvous ("everyone has read").
#define TOYNI_BARRIER() asm volatile("s_barrier" ::: "memory")
#define TOYNI_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define TOYNI_SCHED_FENCE() ((void)0)
#define TOYNI_PIN(v) ((void)0)
#define TOYNI_WAIT_VMEM0() ((void)0)
#define TOYNI_WAIT_VMEM_ALLOW(N) ((void)0)
#define TOYNI_BARRIER() ((void)0)
#define TOYNI_LDS_BARRIER() ((void)0)
#endif

namespace toyni {

constexpr uint32_t BB_P = 2013265921u;          // src/babybear.rs:8
constexpr uint64_t BB_BARRETT_MU = 9162596893ull; // floor(2^64 / p), cuda/ntt_kernel.cu:32

// ---- compile-time helpers ----
constexpr uint32_t cx_mulmod(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % BB_P); }
constexpr uint32_t cx_powmod(uint32_t a, uint64_t e) {
    uint32_t r = 1;
    while (e) { if (e & 1) r = cx_mulmod(r, a); a = cx_mulmod(a, a); e >>= 1; }
    return r;
}
// -p^-1 mod 2^32 by Newton iteration
constexpr uint32_t cx_neg_pinv() {
    uint32_t inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2u - BB_P * inv;
    return 0u - inv;
}
constexpr uint32_t BB_NPINV = cx_neg_pinv();                         // -p^-1 mod 2^32
constexpr uint32_t BB_R1 = (uint32_t)((1ull << 32) % BB_P);          // R mod p   (Montgomery one)
constexpr uint32_t BB_R2 = cx_mulmod(BB_R1, BB_R1);                  // R^2 mod p
constexpr uint32_t BB_HALF = (BB_P + 1) / 2;                         // 2^-1 = 1006632961
constexpr uint32_t BB_GEN_2_27 = 440564289u;                         // src/babybear.rs:122
static_assert((uint32_t)(BB_P * (0u - BB_NPINV)) == 1u, "pinv");
static_assert(BB_HALF == 1006632961u, "2^-1");

// ---- canonical add / sub: one min() replaces compare+select ----
// a, b in [0,p).  src/babybear.rs:129-139
TOYNI_HD uint32_t bb_add(uint32_t a, uint32_t b) {
    uint32_t s = a + b;               // < 2p < 2^32
    uint32_t t = s - BB_P;            // wraps high when s < p
    return s < t ? s : t;
}
// src/babybear.rs:148-160
TOYNI_HD uint32_t bb_sub(uint32_t a, uint32_t b) {
    uint32_t d = a - b;               // wraps high when a < b
    uint32_t t = d + BB_P;
    return d < t ? d : t;
}
// a - b + p in (0, 2p): congruent to a - b, valid as the left operand of mont_mul_lazy
TOYNI_HD uint32_t bb_sub_lazy(uint32_t a, uint32_t b) { return a + (BB_P - b); }
// fold [0,2p) -> [0,p)
TOYNI_HD uint32_t bb_reduce_2p(uint32_t x) {
    uint32_t t = x - BB_P;
    return x < t ? x : t;
}
// x / 2 mod p for canonical x
TOYNI_HD uint32_t bb_halve(uint32_t x) { return (x >> 1) + ((x & 1u) ? BB_HALF : 0u); }

// ---- Montgomery product, R = 2^32 ----
// a: any u32, bR: Montgomery form of b with bR < p.  Returns a*b mod p up to one extra p: [0, 2p).
// Also valid for a < p and bR < 2p (sum stays below 2^64 and the quotient below 2p).
// So a RUNNING PRODUCT of twiddles (tw <- tw * G, used only as the right-hand factor of products with canonical data and as
// the left-hand factor of its own next step) can stay in [0, 2p): the chain skips the conditional subtract (2 of 5 instructions).
TOYNI_HD uint32_t mont_mul_lazy(uint32_t a, uint32_t bR) {
    uint64_t prod = (uint64_t)a * bR;
    uint32_t m = (uint32_t)prod * BB_NPINV;
    uint64_t t = prod + (uint64_t)m * BB_P;   // low 32 bits are zero by construction
    return (uint32_t)(t >> 32);
}
TOYNI_HD uint32_t mont_mul(uint32_t a, uint32_t bR) { return bb_reduce_2p(mont_mul_lazy(a, bR)); }

// (a - b) * w mod p as ONE Montgomery reduction of the two-term dot product a*wR + b*(p - wR):
// a, b canonical, wR and nwR = p - wR canonical Montgomery forms (wR != 0).  The 64-bit sum stays below
// 2 p^2, plus m*p below 2^32 p: < 2^64, quotient < 2p.  Saves the separate subtract of the butterfly
// (integer multiply-adds issue at the add rate on gfx950, so a v_mad_u64_u32 is as cheap as a v_sub).
TOYNI_HD uint32_t mont_dot_sub(uint32_t a, uint32_t b, uint32_t wR, uint32_t nwR) {
    uint64_t t = (uint64_t)a * wR;
    t += (uint64_t)b * nwR;
    const uint32_t m = (uint32_t)t * BB_NPINV;
    t += (uint64_t)m * BB_P;
    return bb_reduce_2p((uint32_t)(t >> 32));
}
// a*bR + c*dR as one Montgomery reduction (a, c canonical; bR, dR canonical Montgomery forms): same bound as above
TOYNI_HD uint32_t mont_dot2(uint32_t a, uint32_t bR, uint32_t c, uint32_t dR) {
    uint64_t t = (uint64_t)a * bR;
    t += (uint64_t)c * dR;
    const uint32_t m = (uint32_t)t * BB_NPINV;
    t += (uint64_t)m * BB_P;
    return bb_reduce_2p((uint32_t)(t >> 32));
}
// aR^(p-2) in the Montgomery domain (a R -> a^-1 R; 0 -> 0), BabyBear::inverse (src/babybear.rs:111-114) by an addition chain:
// p - 2 = 0b111_0_(27 ones) = 15 * 2^27 - 1 is built as 0b1110 followed by nine times "shift by three, append 0b111":
// 30 squarings + 11 products = 41 Montgomery products where the square-and-multiply ladder spends 30 + 30.
TOYNI_HD uint32_t mont_inv_chain(uint32_t aR) {
    const uint32_t x3 = mont_mul(mont_mul(aR, aR), aR);   // exponent 0b11
    const uint32_t x7 = mont_mul(mont_mul(x3, x3), aR);   // 0b111
    uint32_t r = mont_mul(x7, x7);                         // 0b1110
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        r = mont_mul(r, r);
        r = mont_mul(r, r);
        r = mont_mul(r, r);
        r = mont_mul(r, x7);
    }
    return r;
}
TOYNI_HD uint32_t to_mont(uint32_t a) { return mont_mul(a, BB_R2); }
TOYNI_HD uint32_t from_mont(uint32_t aR) { return mont_mul(aR, 1u); }

// ---- plain products (host-side table building, cross-checks) ----
// The north-star's "64-bit Barrett": q = hi64(prod * MU), r = prod - q p, one conditional subtract
// (cuda/ntt_kernel.cu:49-67).
TOYNI_HD uint32_t bb_mul_barrett64(uint32_t a, uint32_t b) {
    uint64_t prod = (uint64_t)a * b;
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t q = __umul64hi(prod, BB_BARRETT_MU);
#else
    uint64_t q = (uint64_t)(((unsigned __int128)prod * BB_BARRETT_MU) >> 64);
#endif
    uint64_t r = prod - q * BB_P;
    return (uint32_t)(r >= BB_P ? r - BB_P : r);
}
inline uint32_t bb_mul_host(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % BB_P); }
inline uint32_t bb_pow_host(uint32_t a, uint64_t e) {
    uint32_t r = 1;
    while (e) { if (e & 1) r = bb_mul_host(r, a); a = bb_mul_host(a, a); e >>= 1; }
    return r;
}
inline uint32_t bb_inv_host(uint32_t a) { return bb_pow_host(a, BB_P - 2); }          // src/babybear.rs:111-114
inline uint32_t bb_root_of_unity_host(uint32_t log_n) {                                 // src/babybear.rs:118-126
    return bb_pow_host(BB_GEN_2_27, 1ull << (27 - log_n));
}
inline uint32_t to_mont_host(uint32_t a) { return (uint32_t)((((uint64_t)a) << 32) % BB_P); }

TOYNI_HD uint32_t bitrev32(uint32_t x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0u;
#else
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1u); x >>= 1; }
    return r;
#endif
}

}  // namespace toyni
