// Memory-pattern probe (diagnostic, not part of the library): what HBM rate do the NTT passes' access patterns
// reach with no arithmetic at all?  Build: hipcc --offload-arch=gfx950 -O3 -o build/membench tools/membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// contiguous 16 B / lane copy
__global__ void copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = in[i];
}
// contiguous 4 B / lane copy
__global__ void copy4(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}
// the same with non-temporal hints (streaming data: read once, written once)
template <int LD_NT, int ST_NT>
__global__ void copy4_nt(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t v = LD_NT ? __builtin_nontemporal_load(in + i) : in[i];
        if (ST_NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
// column tile: tile = C columns x M rows of a [M][S] matrix (row stride ld words); 32 elements per thread, like the pass.
// mode 0: read strided, write same place (in-place layout, other buffer); mode 1: read strided only (sum -> rare store); mode 2: write strided only
template <int C, int MODE>
__global__ void __launch_bounds__(1024) coltile(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t ntiles, uint32_t log_s, uint32_t ld, uint32_t M) {
    const uint32_t tid = threadIdx.x;
    const uint32_t c = tid % C, lo = tid / C;
    const uint32_t rows_per_iter = blockDim.x / C;  // 32
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t tiles_per = (1u << log_s) / C;
        const size_t base = (size_t)(t / tiles_per) * M * ld + (size_t)(t % tiles_per) * C;
        uint32_t x[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const size_t off = base + (size_t)(lo + i * rows_per_iter) * ld + c;
            if (MODE != 2) x[i] = in[off]; else x[i] = tid + i;
        }
        if (MODE == 1) {
            uint32_t s = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) s ^= x[i];
            if (s == 0xDEADBEEF) out[base + tid] = s;
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) out[base + (size_t)(lo + i * rows_per_iter) * ld + c] = x[i] + 1;
        }
    }
}

// narrow column tiles (C = 4 / 8 / 16 words = 16 / 32 / 64-byte row segments) in the XCD-aware order of Pass3::tile_order: the
// G = 32 / C tiles that share 128-byte lines go to workgroups p, p + 8, ... (one XCD under round-robin dispatch), so the L2 sees
// every line once.  E elements per thread, T = C * M / E threads.  PAIRED = 0: tiles in plain order.
template <int C, int E, int PAIRED>
__global__ void __launch_bounds__(1024) coltile_narrow(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t ntiles, uint32_t log_s, uint32_t M) {
    const uint32_t tid = threadIdx.x;
    const uint32_t c = tid % C, lo = tid / C;
    const uint32_t rows_per_iter = blockDim.x / C;
    constexpr uint32_t G = 32 / C;
    const uint32_t S = 1u << log_s;
    for (uint32_t v = blockIdx.x; v < ntiles; v += gridDim.x) {
        uint32_t t = v;
        if (PAIRED) { const uint32_t s8 = v & (8u * G - 1u); t = (v & ~(8u * G - 1u)) | ((s8 & 7u) * G + (s8 >> 3)); }
        const uint32_t tiles_per = S / C;
        const size_t base = (size_t)(t / tiles_per) * M * S + (size_t)(t % tiles_per) * C;
        uint32_t x[E];
#pragma unroll
        for (int i = 0; i < E; ++i) x[i] = in[base + (size_t)(lo + i * rows_per_iter) * S + c];
#pragma unroll
        for (int i = 0; i < E; ++i) out[base + (size_t)(lo + i * rows_per_iter) * S + c] = x[i] + 1;
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
    const size_t n = (size_t)1 << 28;  // 1 GiB of u32
    uint32_t *in, *out;
    CK(hipMalloc(&in, n * 4 + (64 << 20))); CK(hipMalloc(&out, n * 4 + (64 << 20)));
    CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 2, n * 4));
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(copy16, dim3(2048), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, n / 4); }, 10);
    printf("copy16 contiguous           : %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0 * n * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, in, out, n); }, 10);
    printf("copy4 contiguous            : %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0 * n * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((copy4_nt<1, 0>), dim3(2048), dim3(256), 0, 0, in, out, n); }, 10);
    printf("copy4 nt-load               : %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0 * n * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((copy4_nt<0, 1>), dim3(2048), dim3(256), 0, 0, in, out, n); }, 10);
    printf("copy4 nt-store              : %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0 * n * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((copy4_nt<1, 1>), dim3(2048), dim3(256), 0, 0, in, out, n); }, 10);
    printf("copy4 nt-load + nt-store    : %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0 * n * 4 / ms / 1e9);
    for (int g : {1024, 4096, 8192}) {
        ms = timeit([&] { hipLaunchKernelGGL(copy4, dim3(g), dim3(256), 0, 0, in, out, n); }, 10);
        printf("copy4 contiguous grid %5d : %.3f ms  %.2f TB/s (r+w)\n", g, ms, 2.0 * n * 4 / ms / 1e9);
    }
    const uint32_t M = 1024, log_s = 10;
    for (int grid : {256, 512}) {
        const uint32_t nt32 = (uint32_t)(n / (M * 32)), nt16 = (uint32_t)(n / (M * 16));
        ms = timeit([&] { hipLaunchKernelGGL((coltile<32, 0>), dim3(grid), dim3(1024), 0, 0, in, out, nt32, log_s, 1024u, M); }, 10);
        printf("coltile C=32 r+w  grid %4d  : %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((coltile<32, 1>), dim3(grid), dim3(1024), 0, 0, in, out, nt32, log_s, 1024u, M); }, 10);
        printf("coltile C=32 read grid %4d  : %.3f ms  %.2f TB/s\n", grid, ms, 1.0 * n * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((coltile<32, 2>), dim3(grid), dim3(1024), 0, 0, in, out, nt32, log_s, 1024u, M); }, 10);
        printf("coltile C=32 write grid %4d : %.3f ms  %.2f TB/s\n", grid, ms, 1.0 * n * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((coltile<16, 0>), dim3(grid * 2), dim3(512), 0, 0, in, out, nt16, log_s, 1024u, M); }, 10);
        printf("coltile C=16 r+w  grid %4d  : %.3f ms  %.2f TB/s\n", grid * 2, ms, 2.0 * n * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((coltile<64, 0>), dim3(grid), dim3(1024), 0, 0, in, out, (uint32_t)(n / (512 * 64)), log_s, 1024u, 512u); }, 10);
        printf("coltile C=64 M=512 r+w %4d  : %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
    }
    // 4096-point columns of a 4096 x 4096 matrix (n = 2^24, two sweeps): 8- and 4-wide tiles, plain and XCD-paired order
    {
        const uint32_t M4 = 4096, ls = 12;
        for (int grid : {256, 512}) {
            ms = timeit([&] { hipLaunchKernelGGL((coltile_narrow<8, 32, 0>), dim3(grid), dim3(1024), 0, 0, in, out, (uint32_t)(n / (M4 * 8)), ls, M4); }, 10);
            printf("narrow C=8 M=4096 plain  grid %4d : %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
            ms = timeit([&] { hipLaunchKernelGGL((coltile_narrow<8, 32, 1>), dim3(grid), dim3(1024), 0, 0, in, out, (uint32_t)(n / (M4 * 8)), ls, M4); }, 10);
            printf("narrow C=8 M=4096 paired grid %4d : %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
            ms = timeit([&] { hipLaunchKernelGGL((coltile_narrow<4, 16, 1>), dim3(grid), dim3(1024), 0, 0, in, out, (uint32_t)(n / (M4 * 4)), ls, M4); }, 10);
            printf("narrow C=4 M=4096 paired grid %4d : %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
            ms = timeit([&] { hipLaunchKernelGGL((coltile_narrow<16, 32, 1>), dim3(grid), dim3(1024), 0, 0, in, out, (uint32_t)(n / (2048 * 16)), ls, 2048u); }, 10);
            printf("narrow C=16 M=2048 paired grid %4d: %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
            ms = timeit([&] { hipLaunchKernelGGL((coltile_narrow<16, 32, 0>), dim3(grid), dim3(1024), 0, 0, in, out, (uint32_t)(n / (2048 * 16)), ls, 2048u); }, 10);
            printf("narrow C=16 M=2048 plain  grid %4d: %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
        }
    }
    // padded leading dimension: rows 4 KB + 128 B apart (breaks the power-of-two stride)
    {
        const uint32_t ld = 1024 + 32;
        const uint32_t nt32 = (uint32_t)((n / ld / M) * 32);  // whole matrices only
        ms = timeit([&] { hipLaunchKernelGGL((coltile<32, 0>), dim3(256), dim3(1024), 0, 0, in, out, nt32, log_s, ld, M); }, 10);
        printf("coltile C=32 r+w ld=1056    : %.3f ms  %.2f TB/s\n", ms, 2.0 * (double)nt32 * M * 32 * 4 / ms / 1e9);
    }
    return 0;
}
