// Latency probe (diagnostic tool, NOT part of libtoyni_hip.so): how long does ONE SHA-256 compression take on a lone wave --
//   (a) the per-lane VALU form the Merkle kernels use (sha256_compress, merkle_kernels.hpp), and
//   (b) the same arithmetic on wave-UNIFORM values, which the compiler turns into scalar-ALU instructions (s_add_u32, s_xor_b32,
//       s_lshr_b64 for the rotations)?
// A tree level with few nodes is a chain of dependent compressions (204 levels x ~5.8 us per 2^16-row proof); if the
// scalar unit retires a dependent chain faster than one VALU instruction per ~5 cycles, the top levels of every tree could run there.
// Build: hipcc --offload-arch=gfx950 -O3 -I toyni_amd/csrc -o build/shabench tools/shabench.hip     Run: ./build/shabench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "merkle_kernels.hpp"

using namespace toyni;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t s_rotr(uint32_t x, int n) {
    const uint64_t xx = ((uint64_t)x << 32) | x;   // uniform: one s_lshr_b64 on a register pair
    return (uint32_t)(xx >> n);
}
__device__ __forceinline__ void sha_compress_uniform(uint32_t (&h)[8], uint32_t (&w)[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = s_rotr(w15, 7) ^ s_rotr(w15, 18) ^ (w15 >> 3);
            const uint32_t s1 = s_rotr(w2, 17) ^ s_rotr(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const uint32_t t1 = hh + (s_rotr(e, 6) ^ s_rotr(e, 11) ^ s_rotr(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + w[i & 15];
        const uint32_t t2 = (s_rotr(a, 2) ^ s_rotr(a, 13) ^ s_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// reps dependent compressions (the digest of one feeds the message of the next), one result per wave
__global__ void __launch_bounds__(64) chain_valu(int reps, uint32_t seed, uint32_t* out, uint64_t* cycles) {
    Sha256State st = sha256_init();
    uint32_t w[16];
    for (int j = 0; j < 16; ++j) w[j] = seed + j + threadIdx.x;
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        sha256_compress(st, w);
        for (int j = 0; j < 8; ++j) { w[j] = st.h[j]; w[8 + j] = st.h[j] ^ 0x5C5C5C5Cu; }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = st.h[0] ^ st.h[7];
    if (threadIdx.x == 0) cycles[blockIdx.x] = c1 - c0;
}
__global__ void __launch_bounds__(64) chain_salu(int reps, uint32_t seed, uint32_t* out, uint64_t* cycles) {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint32_t w[16];
    for (int j = 0; j < 16; ++j) w[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(seed + j + blockIdx.x));
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        sha_compress_uniform(h, w);
        for (int j = 0; j < 8; ++j) { w[j] = h[j]; w[8 + j] = h[j] ^ 0x5C5C5C5Cu; }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = h[0] ^ h[7]; cycles[blockIdx.x] = c1 - c0; }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 64;
    uint32_t* out;
    uint64_t* cyc;
    CK(hipMalloc(&out, 4 * 64 * 4096));
    CK(hipMalloc(&cyc, 8 * 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int blocks : {1, 256, 1024, 4096}) {
        for (int form = 0; form < 2; ++form) {
            for (int warm = 0; warm < 2; ++warm) {
                CK(hipEventRecord(e0));
                if (form == 0) chain_valu<<<blocks, 64>>>(reps, 12345u, out, cyc);
                else chain_salu<<<blocks, 64>>>(reps, 12345u, out, cyc);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t c[1];
            CK(hipMemcpy(c, cyc, 8, hipMemcpyDeviceToHost));
            printf("%s  waves=%-5d  %7.3f us per compression (wall %8.1f us for %d dependent)   s_memtime ticks per compression: %.0f\n",
                   form ? "SALU (uniform)" : "VALU (per lane)", blocks, ms * 1e3 / reps, ms * 1e3, reps, (double)c[0] / reps);
        }
    }
    return 0;
}
