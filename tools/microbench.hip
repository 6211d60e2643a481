// VALU issue-rate probe for gfx950 (diagnostic tool, NOT part of libtoyni_hip.so).
//
// Question it answers ("settle the VALU ceiling"): MI355X_MICROARCH.md says a wave64 v_fma_f32 issues in
// 2 cycles on CDNA4's SIMD-32 once a SIMD holds more than one wave; round 1 measured ~39 T lane-ops/s for the integer
// instructions the NTT is made of, i.e. 4 cycles per wave-instruction.  This tool measures each instruction ALONE
// (inline asm, 8 independent chains per lane), at 1 / 2 / 4 / 8 waves per SIMD, and reports
//   * cycles per wave-instruction per SIMD from s_memtime (clock-independent), and
//   * chip-wide T lane-ops/s from HIP events, plus the in-kernel clock (s_memtime / s_memrealtime).
//
// Build: hipcc --offload-arch=gfx950 -O3 -o build/microbench tools/microbench.hip
// Run  : ./build/microbench [iters]          (prints a table; profiles/r02_microbench.txt is its output)
#include <hip/hip_runtime.h>

#include "../toyni_amd/csrc/bb_field.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int CHAINS = 8;
constexpr int UNROLL = 4;

struct Stamp { uint64_t cycles, real; };

extern __shared__ uint32_t dyn_lds[];

// One kernel per instruction.  T = register type of the chain, BODY = one asm statement acting on x[j] (and w).
#define DEF_KERNEL(NAME, T, INIT, ...)                                                                              \
    __global__ void __launch_bounds__(1024) NAME(int iters, uint32_t seed, uint32_t* sink, Stamp* stamps) {          \
        T x[CHAINS];                                                                                                 \
        const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;                                                    \
        _Pragma("unroll") for (int j = 0; j < CHAINS; ++j) x[j] = INIT;                                              \
        T w = (T)(seed | 1u);                                                                                        \
        uint32_t w32 = seed * 2654435761u | 1u;                                                                      \
        (void)w32;                                                                                                   \
        if (seed == 0xFFFFFFFFu) dyn_lds[threadIdx.x] = t; /* keeps the dynamic LDS allocation referenced */         \
        __syncthreads();                                                                                             \
        const uint64_t c0 = __builtin_amdgcn_s_memtime();                                                            \
        const uint64_t r0 = __builtin_amdgcn_s_memrealtime();                                                        \
        for (int it = 0; it < iters; ++it) {                                                                         \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                                     \
                _Pragma("unroll") for (int j = 0; j < CHAINS; ++j) { __VA_ARGS__; }                                      \
            }                                                                                                        \
        }                                                                                                            \
        const uint64_t c1 = __builtin_amdgcn_s_memtime();                                                            \
        const uint64_t r1 = __builtin_amdgcn_s_memrealtime();                                                        \
        uint32_t acc = 0;                                                                                            \
        _Pragma("unroll") for (int j = 0; j < CHAINS; ++j) {                                                         \
            const T v = x[j];                                                                                        \
            uint32_t bits[sizeof(T) / 4];                                                                            \
            __builtin_memcpy(bits, &v, sizeof(T));                                                                   \
            for (unsigned b = 0; b < sizeof(T) / 4; ++b) acc ^= bits[b];                                             \
        }                                                                                                            \
        if (acc == 0x12345678u) sink[0] = acc;                                                                       \
        if ((threadIdx.x & 63u) == 0) stamps[t >> 6] = Stamp{c1 - c0, r1 - r0};                                      \
    }

// ---- single instructions ----
DEF_KERNEL(k_fma_f32, float, (float)(t + j) * 1e-9f, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_add_f32, float, (float)(t + j) * 1e-9f, asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_mul_f32, float, (float)(t + j) * 1e-9f, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_pk_fma_f32, double, (double)(t + j), asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_fma_f64, double, (double)(t + j) * 1e-9, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_add_u32, uint32_t, t * 8u + j, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_sub_u32, uint32_t, t * 8u + j, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_min_u32, uint32_t, t * 8u + j, asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_xor_b32, uint32_t, t * 8u + j, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_add3_u32, uint32_t, t * 8u + j, asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_lshl_add_u32, uint32_t, t * 8u + j, asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_cndmask, uint32_t, t * 8u + j, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_mul_lo_u32, uint32_t, t * 8u + j, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_mul_hi_u32, uint32_t, t * 8u + j, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_mul_u32_u24, uint32_t, t * 8u + j, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_mul_hi_u32_u24, uint32_t, t * 8u + j, asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_mad_u32_u24, uint32_t, t * 8u + j, asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_mad_u64_u32, uint64_t, (uint64_t)t * 8u + j,
           asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[j]) : "v"(w32), "v"((uint32_t)w) : "vcc"))
DEF_KERNEL(k_lshl_add_u64, uint64_t, (uint64_t)t * 8u + j,
           asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[j]) : "v"((uint64_t)w)))
DEF_KERNEL(k_alignbit, uint32_t, t * 8u + j, asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_bitop3, uint32_t, t * 8u + j, asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_pk_add_u16, uint32_t, t * 8u + j, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_pk_mul_lo_u16, uint32_t, t * 8u + j, asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_pk_mad_u16, uint32_t, t * 8u + j, asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_cvt_f32_u32, uint32_t, t * 8u + j, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_mov_b32, uint32_t, t * 8u + j, asm volatile("v_mov_b32 %0, %1" : "+v"(x[j]) : "v"(w)))
// ---- pairs: do float and integer instructions share the issue slots? (2 instructions per BODY) ----
DEF_KERNEL(k_mix_fma_add, uint32_t, t * 8u + j,
           asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(w)))
DEF_KERNEL(k_mix_mad64_add, uint64_t, (uint64_t)t * 8u + j,
           asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2" : "+v"(x[j]), "+v"(w32) : "v"((uint32_t)w) : "vcc"))
// ---- the NTT's own sequences ----
// Montgomery product, R = 2^32 (bb_field.hpp mont_mul, compiled from the library's own source): mad_u64, mul_lo, mad_u64,
// sub, min = 5 instructions
DEF_KERNEL(k_mont_mul, uint32_t, (t * 8u + j) % toyni::BB_P, x[j] = toyni::mont_mul(x[j], w); asm volatile("" : "+v"(x[j])))

struct Probe {
    const char* name;
    void (*kernel)(int, uint32_t, uint32_t*, Stamp*);
    int insts_per_body;   // wave-instructions per BODY
    int lanes_ops;        // lane-operations counted per instruction (2 for packed)
};

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, %d MHz nominal; %d iterations x %d unroll x %d chains per lane\n", prop.gcnArchName, cus, prop.clockRate / 1000,
           iters, UNROLL, CHAINS);
    const Probe probes[] = {
        {"v_fma_f32", k_fma_f32, 1, 1},         {"v_add_f32", k_add_f32, 1, 1},           {"v_mul_f32", k_mul_f32, 1, 1},
        {"v_pk_fma_f32", k_pk_fma_f32, 1, 2},   {"v_fma_f64", k_fma_f64, 1, 1},           {"v_add_u32", k_add_u32, 1, 1},
        {"v_sub_u32", k_sub_u32, 1, 1},         {"v_min_u32", k_min_u32, 1, 1},           {"v_xor_b32", k_xor_b32, 1, 1},
        {"v_add3_u32", k_add3_u32, 1, 1},       {"v_lshl_add_u32", k_lshl_add_u32, 1, 1}, {"v_cndmask_b32", k_cndmask, 1, 1},
        {"v_mov_b32", k_mov_b32, 1, 1},         {"v_cvt_f32_u32", k_cvt_f32_u32, 1, 1},   {"v_mul_lo_u32", k_mul_lo_u32, 1, 1},
        {"v_mul_hi_u32", k_mul_hi_u32, 1, 1},   {"v_mul_u32_u24", k_mul_u32_u24, 1, 1},   {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1, 1},
        {"v_mad_u32_u24", k_mad_u32_u24, 1, 1}, {"v_mad_u64_u32", k_mad_u64_u32, 1, 1},   {"v_pk_add_u16", k_pk_add_u16, 1, 2},
        {"v_lshl_add_u64", k_lshl_add_u64, 1, 1}, {"v_alignbit_b32", k_alignbit, 1, 1},     {"v_bitop3_b32", k_bitop3, 1, 1},
        {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1, 2}, {"v_pk_mad_u16", k_pk_mad_u16, 1, 2},
        {"fma_f32+add_u32", k_mix_fma_add, 2, 1}, {"mad_u64+add_u32", k_mix_mad64_add, 2, 1}, {"mont_mul(5 instr)", k_mont_mul, 5, 1},
    };
    uint32_t* d_sink;
    Stamp* d_stamps;
    const int max_waves = cus * 32;
    CK(hipMalloc((void**)&d_sink, 16));
    CK(hipMalloc((void**)&d_stamps, sizeof(Stamp) * max_waves));
    std::vector<Stamp> h(max_waves);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-20s", "instruction");
    for (int k : {1, 2, 4, 8}) printf(" | k=%d: cyc/inst/SIMD  Tlane-op/s  GHz", k);
    printf("\n");
    for (const Probe& p : probes) {
        printf("%-20s", p.name);
        for (int k : {1, 2, 4, 8}) {
            // k waves per SIMD = 4k waves per CU: one workgroup of 256*min(k,4) threads, k/4 (>= 1) workgroups per CU pinned by LDS
            const int threads = 256 * std::min(k, 4);
            const int wg_per_cu = std::max(1, k / 4);
            const size_t lds = wg_per_cu == 1 ? 96 * 1024 : 64 * 1024;  // 1 x 96 KiB or 2 x 64 KiB fit in 160 KiB, one more does not
            CK(hipFuncSetAttribute((const void*)p.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = cus * wg_per_cu;
            hipLaunchKernelGGL(p.kernel, dim3(grid), dim3(threads), lds, 0, iters / 8 + 1, 12345u, d_sink, d_stamps);  // warm
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(p.kernel, dim3(grid), dim3(threads), lds, 0, iters, 12345u, d_sink, d_stamps);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const int waves = grid * threads / 64;
            CK(hipMemcpy(h.data(), d_stamps, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
            std::vector<double> cyc(waves), ghz(waves);
            for (int i = 0; i < waves; ++i) { cyc[i] = (double)h[i].cycles; ghz[i] = h[i].real ? (double)h[i].cycles / (double)h[i].real * 0.1 : 0.0; }
            std::nth_element(cyc.begin(), cyc.begin() + waves / 2, cyc.end());
            std::nth_element(ghz.begin(), ghz.begin() + waves / 2, ghz.end());
            const double insts_per_wave = (double)iters * UNROLL * CHAINS * p.insts_per_body;
            const double cyc_per_inst_simd = cyc[waves / 2] / (insts_per_wave * k);  // k waves share one SIMD
            const double tops = insts_per_wave * waves * 64.0 * p.lanes_ops / (ms * 1e-3) / 1e12;
            printf(" | %19.2f %11.2f %5.2f", cyc_per_inst_simd, tops, ghz[waves / 2]);
        }
        printf("\n");
        fflush(stdout);
    }
    printf("# cyc/inst/SIMD = median over waves of (s_memtime delta) / (wave-instructions of one wave x k waves per SIMD)\n");
    printf("# 64 lanes / (cyc/inst/SIMD) x 4 SIMDs x %d CUs x clock = chip-wide lane-op rate; 4.0 cycles <-> 39.3 T at 2.4 GHz, 2.0 <-> 78.6 T\n", cus);
    return 0;
}
