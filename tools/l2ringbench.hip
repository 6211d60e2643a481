// Would an XCD-local intermediate pay?  (diagnostic, bandwidth only -- no ordering between workgroups, the VALUES are meaningless)
// 256 workgroups, b -> XCD b & 7 (tools/xccprobe.hip), slot b >> 3.  Per chunk every workgroup streams its slice of X into a ring that
// belongs to ITS XCD (ring slots of `ring_kib` KiB per XCD), then reads the slice another workgroup OF THE SAME XCD wrote (L1-bypassing
// loads, sc0) and streams it to Y.  ring_kib small enough -> the middle traffic can live in the XCD's 4 MiB L2; ring = all chunks ->
// it goes through memory like today's two sweeps.  Compare "TB/s moved" (16 B per word) with tools/cachebench.hip part B.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/l2ringbench tools/l2ringbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline uint32_t load_l1_bypass(const uint32_t* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int BYPASS>
__global__ void __launch_bounds__(1024) ring(const uint32_t* __restrict__ x, uint32_t* __restrict__ rings, uint32_t* __restrict__ y,
                                             uint32_t nchunks, uint32_t slice_words, uint32_t slots) {
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;          // 32 workgroups per XCD
    const uint32_t chunk_words = slice_words * 32u;                         // one chunk per XCD per iteration
    const uint32_t other = (slot + 13u) & 31u;
    uint32_t* my_ring = rings + (size_t)xcd * slots * chunk_words;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const size_t g = ((size_t)c * 8u + xcd) * chunk_words;             // this XCD's chunk of X / Y
        uint32_t* mid = my_ring + (size_t)(c % slots) * chunk_words;
        for (uint32_t i = threadIdx.x; i < slice_words; i += 1024u)
            mid[slot * slice_words + i] = __builtin_nontemporal_load(x + g + slot * slice_words + i) + 1u;
        for (uint32_t i = threadIdx.x; i < slice_words; i += 1024u) {
            const uint32_t v = BYPASS ? load_l1_bypass(mid + other * slice_words + i) : mid[other * slice_words + i];
            __builtin_nontemporal_store(v, y + g + slot * slice_words + i);
        }
    }
}

int main() {
    const size_t n = (size_t)1 << 30;   // 4 GiB of u32 for X and Y each
    uint32_t *x, *y, *r;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&r, n * 4));
    CK(hipMemset(x, 1, n * 4)); CK(hipMemset(y, 2, n * 4)); CK(hipMemset(r, 3, n * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (uint32_t slice_kib : {32u, 128u}) {          // per workgroup per chunk; a chunk per XCD = 32 slices = 1 / 4 MiB
        const uint32_t slice_words = slice_kib * 256u, chunk_words = slice_words * 32u;
        const uint32_t nchunks = (uint32_t)(n / ((size_t)chunk_words * 8u));
        for (uint32_t slots : {1u, 2u, 4u, 16u, nchunks}) {
            if (slots > nchunks) continue;
            for (int bypass = 0; bypass < 2; ++bypass) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(a));
                    if (bypass) hipLaunchKernelGGL((ring<1>), dim3(256), dim3(1024), 0, 0, x, r, y, nchunks, slice_words, slots);
                    else hipLaunchKernelGGL((ring<0>), dim3(256), dim3(1024), 0, 0, x, r, y, nchunks, slice_words, slots);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    if (ms < best) best = ms;
                }
                printf("slice %3u KiB, ring per XCD %8.1f MiB, %s : %7.3f ms  %5.2f TB/s moved  (%.2f TB/s of X+Y)\n", slice_kib,
                       (double)slots * chunk_words * 4 / 1048576.0, bypass ? "sc0 loads  " : "plain loads", best, 4.0 * n * 4 / best / 1e9, 2.0 * n * 4 / best / 1e9);
            }
        }
    }
    return 0;
}
