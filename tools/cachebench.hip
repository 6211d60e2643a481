// Infinity-Cache probe (diagnostic, not part of the library): is a two-sweep pipeline faster when the intermediate between the
// sweeps stays within the 256 MiB memory-side cache?  Build: hipcc --offload-arch=gfx950 -O3 -o build/cachebench tools/cachebench.hip
//   A  copy inside a working set of W bytes, repeated: the rate a cache-resident stream reaches
//   B  X -> ring -> Y in one persistent kernel: per chunk every workgroup copies its slice of X into the ring slot, then a
//      (shifted) slice of the ring slot into Y.  No ordering between workgroups: the bytes moved are what matters, not the values.
//      ring = all chunks: the intermediate is written once and read once from HBM (today's two sweeps); ring = a few slots: it
//      lives in the cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void copy4(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

template <int NT>
__global__ void __launch_bounds__(256) ring_pipe(const uint32_t* __restrict__ x, uint32_t* __restrict__ ring, uint32_t* __restrict__ y,
                                                 uint32_t nchunks, uint32_t chunk_words, uint32_t slots) {
    const uint32_t per_wg = chunk_words / gridDim.x;           // words of a chunk each workgroup handles
    const uint32_t mine = blockIdx.x * per_wg;
    const uint32_t other = ((blockIdx.x + gridDim.x / 2 + 1) % gridDim.x) * per_wg;   // someone else's slice
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint32_t* src = x + (size_t)c * chunk_words + mine;
        uint32_t* mid = ring + (size_t)(c % slots) * chunk_words;
        uint32_t* dst = y + (size_t)c * chunk_words + mine;
        for (uint32_t i = threadIdx.x; i < per_wg; i += 256) {
            const uint32_t v = NT ? __builtin_nontemporal_load(src + i) : src[i];
            mid[mine + i] = v + 1u;
        }
        for (uint32_t i = threadIdx.x; i < per_wg; i += 256) {
            const uint32_t v = mid[other + i];
            if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
        }
    }
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t n = (size_t)1 << 30;  // 4 GiB of u32 each for X, Y and the full-size intermediate
    uint32_t *x, *y, *t;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&t, n * 4));
    CK(hipMemset(x, 1, n * 4)); CK(hipMemset(y, 2, n * 4)); CK(hipMemset(t, 3, n * 4));
    printf("A: copy inside a working set (read half, write half), repeated\n");
    for (size_t mib : {8, 16, 32, 64, 128, 192, 256, 384, 512, 1024, 4096}) {
        const size_t words = (mib << 20) / 4 / 2;
        const int reps = (int)((size_t)16384 / mib) + 2;
        float ms = timeit([&] { hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, x, x + words, words); }, reps);
        printf("  working set %5zu MiB : %8.4f ms  %6.2f TB/s (r+w)\n", mib, ms, 2.0 * words * 4 / ms / 1e9);
    }
    printf("B: X -> ring -> Y, 4 GiB through a ring of `slots` chunks (16 B moved per word)\n");
    for (uint32_t chunk_mib : {4u, 16u}) {
        const uint32_t chunk_words = chunk_mib << 18;
        const uint32_t nchunks = (uint32_t)(n / chunk_words);
        for (uint32_t slots : {2u, 4u, 8u, 16u, 32u, nchunks}) {
            if (slots > nchunks) continue;
            for (int nt = 0; nt < 2; ++nt) {
                float ms = timeit([&] {
                    if (nt) hipLaunchKernelGGL((ring_pipe<1>), dim3(2048), dim3(256), 0, 0, x, t, y, nchunks, chunk_words, slots);
                    else hipLaunchKernelGGL((ring_pipe<0>), dim3(2048), dim3(256), 0, 0, x, t, y, nchunks, chunk_words, slots);
                }, 3);
                printf("  chunk %2u MiB ring %5u MiB %s : %8.3f ms  %6.2f TB/s moved  (%.2f TB/s of X+Y)\n", chunk_mib, slots * chunk_mib,
                       nt ? "nt" : "  ", ms, 4.0 * n * 4 / ms / 1e9, 2.0 * n * 4 / ms / 1e9);
            }
        }
    }
    return 0;
}
