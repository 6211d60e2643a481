// Access-pattern probe (diagnostic, not part of the library): the tile patterns of the NTT passes with 4-byte and with 16-byte
// lanes.  tools/copyceiling.hip showed that a contiguous copy moves 6.2 TB/s with 16-byte lanes (64 KiB chunks per workgroup) but
// only 5.3-5.45 TB/s with 4-byte lanes; the pass kernels use 4-byte lanes on 128-byte row segments.  Question: what do the SAME
// tile shapes move when a lane takes 4 consecutive words (8 lanes per 128-byte segment, a wave instruction = 8 row segments)?
//   col : tile = C columns x M rows of an [M][S] matrix per transform, read from X, written to the same place of W
//   row : tile = C rows of M contiguous words (C * M contiguous), written transposed: C-word segments at stride S (the closing pass)
// No arithmetic (values + 1), persistent workgroups, the next tile's loads issued before this tile's stores (like the pass kernels).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/tilebench tools/tilebench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int W> struct Lane;
template <> struct Lane<1> {
    typedef unsigned T;
    static __device__ __forceinline__ T ld(const unsigned* p) { return __builtin_nontemporal_load(p); }
    static __device__ __forceinline__ void st(unsigned* p, T v) { __builtin_nontemporal_store(v + 1u, p); }
};
template <> struct Lane<4> {
    typedef v4u T;
    static __device__ __forceinline__ T ld(const unsigned* p) { return __builtin_nontemporal_load((const v4u*)p); }
    static __device__ __forceinline__ void st(unsigned* p, T v) { __builtin_nontemporal_store(v + 1u, (v4u*)p); }
};

// Column tiles.  LW / SW = words per lane of the loads / stores (1 or 4).  A thread moves 32 words per tile either way.
// Words (r, c) of the tile, r < M, c < C: address = tile base + r * ld + c.
// Lane map for width W: lanes per row segment = C / W; thread t -> c = (t % (C / W)) * W, r0 = t / (C / W); rows r0 + k * (T * W / C).
template <int C, int LW, int SW, int T>
__global__ void __launch_bounds__(T) col_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out, unsigned ntiles, unsigned M, unsigned ld,
                                                unsigned tiles_per_matrix) {
    constexpr int NL = 32 / LW, NS = 32 / SW;              // accesses per thread per tile
    const unsigned t = threadIdx.x;
    const unsigned lc = (t % (C / LW)) * LW, lr = t / (C / LW), lstep = T * LW / C;
    const unsigned sc = (t % (C / SW)) * SW, sr = t / (C / SW), sstep = T * SW / C;
    typename Lane<LW>::T x[NL];
    unsigned tile = blockIdx.x;
    auto base_of = [&](unsigned tl) { return (size_t)(tl / tiles_per_matrix) * M * ld + (size_t)(tl % tiles_per_matrix) * C; };
    if (tile < ntiles) {
        const unsigned* p = in + base_of(tile) + (size_t)lr * ld + lc;
#pragma unroll
        for (int k = 0; k < NL; ++k) x[k] = Lane<LW>::ld(p + (size_t)k * lstep * ld);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        // the tile "goes through LDS" in the real kernels; here the registers are re-used as they are when LW == SW, and shuffled
        // through a register copy otherwise (the probe measures memory, not the transposition)
        typename Lane<SW>::T y[NS];
        if constexpr (LW == SW) {
#pragma unroll
            for (int k = 0; k < NS; ++k) y[k] = x[k];
        } else if constexpr (LW == 4) {
#pragma unroll
            for (int k = 0; k < NS; ++k) y[k] = x[k / 4][k % 4];
        } else {
#pragma unroll
            for (int k = 0; k < NS; ++k) { y[k].x = x[4 * k]; y[k].y = x[4 * k + 1]; y[k].z = x[4 * k + 2]; y[k].w = x[4 * k + 3]; }
        }
        const unsigned next = tile + gridDim.x;
        if (next < ntiles) {
            const unsigned* p = in + base_of(next) + (size_t)lr * ld + lc;
#pragma unroll
            for (int k = 0; k < NL; ++k) x[k] = Lane<LW>::ld(p + (size_t)k * lstep * ld);
        }
        unsigned* q = out + base_of(tile) + (size_t)sr * ld + sc;
#pragma unroll
        for (int k = 0; k < NS; ++k) Lane<SW>::st(q + (size_t)k * sstep * ld, y[k]);
    }
}

// Row tiles (closing pass): reads C * M contiguous words (C rows k1 of M words), stores word (k1, k2) at out[k2 * S + k1] (C-word
// segments at stride S).  Loads: lane width LW on the contiguous block.  Stores: lane width SW along k1.
template <int C, int LW, int SW, int T>
__global__ void __launch_bounds__(T) row_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out, unsigned ntiles, unsigned M, unsigned S,
                                                unsigned tiles_per_matrix) {
    constexpr int NL = 32 / LW, NS = 32 / SW;
    const unsigned t = threadIdx.x;
    const unsigned sc = (t % (C / SW)) * SW, sr = t / (C / SW), sstep = T * SW / C;
    typename Lane<LW>::T x[NL];
    unsigned tile = blockIdx.x;
    auto in_of = [&](unsigned tl) { return (size_t)tl * C * M; };
    auto out_of = [&](unsigned tl) { return (size_t)(tl / tiles_per_matrix) * M * S + (size_t)(tl % tiles_per_matrix) * C; };
    if (tile < ntiles) {
        const unsigned* p = in + in_of(tile) + (size_t)t * LW;
#pragma unroll
        for (int k = 0; k < NL; ++k) x[k] = Lane<LW>::ld(p + (size_t)k * T * LW);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        typename Lane<SW>::T y[NS];
        if constexpr (LW == SW) {
#pragma unroll
            for (int k = 0; k < NS; ++k) y[k] = x[k];
        } else if constexpr (LW == 4) {
#pragma unroll
            for (int k = 0; k < NS; ++k) y[k] = x[k / 4][k % 4];
        } else {
#pragma unroll
            for (int k = 0; k < NS; ++k) { y[k].x = x[4 * k]; y[k].y = x[4 * k + 1]; y[k].z = x[4 * k + 2]; y[k].w = x[4 * k + 3]; }
        }
        const unsigned next = tile + gridDim.x;
        if (next < ntiles) {
            const unsigned* p = in + in_of(next) + (size_t)t * LW;
#pragma unroll
            for (int k = 0; k < NL; ++k) x[k] = Lane<LW>::ld(p + (size_t)k * T * LW);
        }
        unsigned* q = out + out_of(tile) + (size_t)sr * S + sc;
#pragma unroll
        for (int k = 0; k < NS; ++k) Lane<SW>::st(q + (size_t)k * sstep * S, y[k]);
    }
}

template <class F> static float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms / reps);
    }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return *std::min_element(t.begin(), t.end());
}

int main() {
    const size_t n = (size_t)1 << 30;  // 4 GiB of u32: the headline batch (1024 x 2^20, 64 x 2^24)
    unsigned *in, *out;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 2, n * 4)); CK(hipDeviceSynchronize());
    const double bytes = 2.0 * n * 4;
#define RUN(name, kern, T, ntiles, ...)                                                                                              \
    for (int grid : {256, 512}) {                                                                                                    \
        if (grid * T > 256 * 2048) continue;                                                                                         \
        const float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(T), 0, 0, in, out, (unsigned)(ntiles), __VA_ARGS__); }, 4); \
        printf("%-58s grid %4d x %4d : %.3f ms  %.2f TB/s\n", name, grid, T, ms, bytes / ms / 1e9);                                  \
        fflush(stdout);                                                                                                              \
    }
    // ---- n = 2^20: [1024][1024] per transform, 32-wide tiles of 1024 rows (the shapes of Pass<0,5,5,5> / Pass<1,5,5,5>)
    {
        const unsigned M = 1024, S = 1024, tiles = (unsigned)(n / (32 * 1024)), tpm = S / 32;
        RUN("2^20 col 32 x 1024, loads 4 B, stores 4 B", (col_kernel<32, 1, 1, 1024>), 1024, tiles, M, S, tpm);
        RUN("2^20 col 32 x 1024, loads 16 B, stores 4 B", (col_kernel<32, 4, 1, 1024>), 1024, tiles, M, S, tpm);
        RUN("2^20 col 32 x 1024, loads 4 B, stores 16 B", (col_kernel<32, 1, 4, 1024>), 1024, tiles, M, S, tpm);
        RUN("2^20 col 32 x 1024, loads 16 B, stores 16 B", (col_kernel<32, 4, 4, 1024>), 1024, tiles, M, S, tpm);
        RUN("2^20 row 32 x 1024, loads 4 B, stores 4 B", (row_kernel<32, 1, 1, 1024>), 1024, tiles, M, S, tpm);
        RUN("2^20 row 32 x 1024, loads 16 B, stores 4 B", (row_kernel<32, 4, 1, 1024>), 1024, tiles, M, S, tpm);
        RUN("2^20 row 32 x 1024, loads 16 B, stores 16 B", (row_kernel<32, 4, 4, 1024>), 1024, tiles, M, S, tpm);
        // 64-wide tiles of 512 rows: what a (9, 11) or tile-major variant would see
        RUN("2^20-like col 64 x 512 (ld 1024), loads 4 B, stores 4 B", (col_kernel<64, 1, 1, 1024>), 1024, tiles, 512u, S, S / 64);
        RUN("2^20-like col 64 x 512 (ld 1024), loads 16 B, stores 16 B", (col_kernel<64, 4, 4, 1024>), 1024, tiles, 512u, S, S / 64);
    }
    // ---- n = 2^24, three passes of 256 points: first pass [256][65536], 64-wide tiles (Pass<0,4,4,6>): 16 K words per tile, 512 threads
    {
        const unsigned M = 256, S = 65536, tiles = (unsigned)(n / (64 * 256)), tpm = S / 64;
        RUN("2^24 col 64 x 256 (ld 65536), loads 4 B, stores 4 B", (col_kernel<64, 1, 1, 512>), 512, tiles, M, S, tpm);
        RUN("2^24 col 64 x 256 (ld 65536), loads 16 B, stores 4 B", (col_kernel<64, 4, 1, 512>), 512, tiles, M, S, tpm);
        RUN("2^24 col 64 x 256 (ld 65536), loads 16 B, stores 16 B", (col_kernel<64, 4, 4, 512>), 512, tiles, M, S, tpm);
        // middle pass [256][256] per block of 65536: ld 256
        RUN("2^24 mid col 64 x 256 (ld 256), loads 4 B, stores 4 B", (col_kernel<64, 1, 1, 512>), 512, tiles, M, 256u, 256u / 64);
        RUN("2^24 mid col 64 x 256 (ld 256), loads 16 B, stores 16 B", (col_kernel<64, 4, 4, 512>), 512, tiles, M, 256u, 256u / 64);
        // closing pass: 64 rows of 256 contiguous words, stored as 64-word segments at stride 65536
        RUN("2^24 row 64 x 256 -> stride 65536, loads 4 B, stores 4 B", (row_kernel<64, 1, 1, 512>), 512, tiles, M, S, tpm);
        RUN("2^24 row 64 x 256 -> stride 65536, loads 16 B, stores 4 B", (row_kernel<64, 4, 1, 512>), 512, tiles, M, S, tpm);
        RUN("2^24 row 64 x 256 -> stride 65536, loads 16 B, stores 16 B", (row_kernel<64, 4, 4, 512>), 512, tiles, M, S, tpm);
        // 128-wide tiles
        RUN("2^24 col 128 x 256 (ld 65536), loads 16 B, stores 16 B", (col_kernel<128, 4, 4, 1024>), 1024, tiles / 2, M, S, S / 128);
        RUN("2^24 col 128 x 256 (ld 65536), loads 4 B, stores 4 B", (col_kernel<128, 1, 1, 1024>), 1024, tiles / 2, M, S, S / 128);
    }
    // ---- two-sweep n = 2^24 (4096 x 4096): 8-wide tiles of 4096 rows with 16-byte lanes (2 lanes per 32-byte segment)
    {
        const unsigned M = 4096, S = 4096, tiles = (unsigned)(n / (8 * 4096)), tpm = S / 8;
        RUN("2^24 two-sweep col 8 x 4096, loads 4 B, stores 4 B", (col_kernel<8, 1, 1, 1024>), 1024, tiles, M, S, tpm);
        RUN("2^24 two-sweep col 8 x 4096, loads 16 B, stores 16 B", (col_kernel<8, 4, 4, 1024>), 1024, tiles, M, S, tpm);
        RUN("2^24 two-sweep col 16 x 2048, loads 16 B, stores 16 B", (col_kernel<16, 4, 4, 1024>), 1024, tiles, 2048u, S, S / 16);
    }
    CK(hipFree(in)); CK(hipFree(out));
    return 0;
}
