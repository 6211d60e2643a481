// Structure probe (diagnostic, not part of the library) for VERDICT r3 #6: would the single-pass transforms (n <= 1024, KIND_ROW_N)
// gain from 16-byte lanes?  A tile of those shapes is C rows of M contiguous words = ONE contiguous block on both sides, but
// step 1 needs every thread's elements at stride E2, so 16-byte lanes mean staging the tile through LDS on the way in and out.
// Two kernels over the same 4 GiB -> 4 GiB, persistent workgroups, next tile's loads issued ahead of this tile's stores, and a
// dial of dummy Montgomery products per element standing in for the butterflies (W = 0: memory only):
//   direct : today's structure -- each thread loads its E1 words at stride E2 with 4-byte lanes (runs of E2 words per row and
//            instruction), parks them in LDS, reads E2 words back under the other lane map, stores 4-byte lanes in runs of E1 words
//   staged : the tile comes in as contiguous 16-byte lanes -> LDS (b128) -> barrier -> the same two LDS round trips ->
//            natural order in LDS -> barrier -> contiguous 16-byte lanes out: two more barriers and one more LDS round trip per side
// Build: hipcc --offload-arch=gfx950 -O3 -o build/rowstage_bench tools/rowstage_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr unsigned P = 2013265921u, NPINV = 2013265919u;   // -p^-1 mod 2^32 of BabyBear

__device__ __forceinline__ unsigned mont(unsigned a, unsigned b) {
    unsigned long long t = (unsigned long long)a * b;
    const unsigned m = (unsigned)t * NPINV;
    t += (unsigned long long)m * P;
    const unsigned r = (unsigned)(t >> 32), s = r - P;
    return r < s ? r : s;
}
template <int W> __device__ __forceinline__ unsigned work(unsigned x, unsigned tw) {
#pragma unroll
    for (int k = 0; k < W; ++k) x = mont(x, tw);
    return x;
}

// LE1 + LE2 = log2 M, C rows per tile, T = C * E2 threads, E1 words per thread (the library's KIND_ROW_N shapes)
template <int LE1, int LE2, int LC, int W, bool STAGED>
__global__ void __launch_bounds__((1 << LC) << LE2) rowtile_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out, unsigned ntiles, unsigned tw) {
    constexpr unsigned E1 = 1u << LE1, E2 = 1u << LE2, M = E1 * E2, C = 1u << LC, T = C * E2, TILE = C * M;
    constexpr unsigned PITCH = E1 * (E2 + 1) + 1;           // the library's padded row layout (conflict-free both ways)
    constexpr unsigned Q = TILE / 4 / T;                     // 16-byte accesses per thread per tile (staged)
    __shared__ __attribute__((aligned(16))) unsigned lds[C * PITCH + (STAGED ? TILE : 0)];
    unsigned* stage = lds + C * PITCH;                       // staged: the linear copy of the tile (in, then out)
    const unsigned t = threadIdx.x;
    const unsigned lo = t & (E2 - 1), c = t >> LE2;          // step 1: lanes over the contiguous row
    const unsigned hi = t & (E1 - 1), c2 = (t >> LE1) % C;   // step 2: lanes over hi (the library's coords2 for ROW_N); G2 = E1 / E2 groups
    unsigned x[E1];
    v4u pre[Q ? Q : 1];
    unsigned tile = blockIdx.x;
    auto issue = [&](unsigned tl) {
        const unsigned* base = in + (size_t)tl * TILE;
        if constexpr (STAGED) {
#pragma unroll
            for (unsigned q = 0; q < Q; ++q) pre[q] = __builtin_nontemporal_load((const v4u*)base + t + q * T);
        } else {
#pragma unroll
            for (unsigned i = 0; i < E1; ++i) x[i] = __builtin_nontemporal_load(base + c * M + lo + i * E2);
        }
    };
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        if constexpr (STAGED) {
#pragma unroll
            for (unsigned q = 0; q < Q; ++q) *((v4u*)stage + t + q * T) = pre[q];
            __syncthreads();
#pragma unroll
            for (unsigned i = 0; i < E1; ++i) x[i] = stage[c * M + lo + i * E2];
        }
#pragma unroll
        for (unsigned i = 0; i < E1; ++i) lds[c * PITCH + i * (E2 + 1) + lo] = work<W>(x[i], tw);      // "step 1", parked
        const unsigned next = tile + gridDim.x;
        if (next < ntiles) issue(next);                      // the next tile's loads: ahead of this tile's stores
        __syncthreads();
        unsigned* obase = out + (size_t)tile * TILE;
#pragma unroll
        for (unsigned g = 0; g < E1 / E2; ++g) {             // "step 2": groups of E2 words under the other lane map
            const unsigned gamma = t + g * T, h = gamma & (E1 - 1), cc = gamma >> LE1;
            unsigned y[E2];
#pragma unroll
            for (unsigned j = 0; j < E2; ++j) y[j] = work<W>(lds[cc * PITCH + h * (E2 + 1) + j], tw);
#pragma unroll
            for (unsigned j = 0; j < E2; ++j) {
                const unsigned k = j * E1 + h;               // natural sub-index (the bit reversals do not change the pattern)
                if constexpr (STAGED) stage[cc * M + k] = y[j];
                else __builtin_nontemporal_store(y[j], obase + cc * M + k);
            }
        }
        (void)hi; (void)c2;
        __syncthreads();                                      // the tile's LDS reads are done (and, staged, the natural copy is complete)
        if constexpr (STAGED) {
#pragma unroll
            for (unsigned q = 0; q < Q; ++q) __builtin_nontemporal_store(*((const v4u*)stage + t + q * T), (v4u*)obase + t + q * T);
            __syncthreads();                                  // before the next tile's staging overwrites it
        }
    }
}

template <int LE1, int LE2, int LC, int W, bool STAGED>
static double run(const unsigned* in, unsigned* out, size_t words, int wg_per_cu) {
    constexpr unsigned TILE = (1u << LC) << (LE1 + LE2), T = (1u << LC) << LE2;
    const unsigned ntiles = (unsigned)(words / TILE);
    unsigned grid = 256u * (unsigned)wg_per_cu;
    if (grid > ntiles) grid = ntiles;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    rowtile_kernel<LE1, LE2, LC, W, STAGED><<<grid, T>>>(in, out, ntiles, 123456789u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) rowtile_kernel<LE1, LE2, LC, W, STAGED><<<grid, T>>>(in, out, ntiles, 123456789u);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return 2.0 * words * 4 / (ms / 5 * 1e-3) / 1e12;        // TB/s read + written
}

template <int LE1, int LE2, int LC>
static void shape(const unsigned* in, unsigned* out, size_t words) {
    constexpr unsigned T = (1u << LC) << LE2, PITCH = (1u << LE1) * ((1u << LE2) + 1) + 1, TILE = (1u << LC) << (LE1 + LE2);
    const int wg_direct = (int)(160u * 1024u / ((1u << LC) * PITCH * 4u)), wg_staged = (int)(160u * 1024u / (((1u << LC) * PITCH + TILE) * 4u));
    auto cap = [](int v, unsigned threads) { const int byw = (int)(2048u / threads); return v < 1 ? 1 : (v > byw ? byw : (v > 8 ? 8 : v)); };
    printf("n = 2^%d, %u rows per tile, %u threads (workgroups per CU: %d direct / %d staged)\n", LE1 + LE2, 1u << LC, T, cap(wg_direct, T), cap(wg_staged, T));
    printf("   dummy products per element:        0        3        7\n");
    printf("   direct, 4-byte lanes      :  %6.2f   %6.2f   %6.2f  TB/s\n", run<LE1, LE2, LC, 0, false>(in, out, words, cap(wg_direct, T)),
           run<LE1, LE2, LC, 3, false>(in, out, words, cap(wg_direct, T)), run<LE1, LE2, LC, 7, false>(in, out, words, cap(wg_direct, T)));
    printf("   staged, 16-byte lanes     :  %6.2f   %6.2f   %6.2f  TB/s\n", run<LE1, LE2, LC, 0, true>(in, out, words, cap(wg_staged, T)),
           run<LE1, LE2, LC, 3, true>(in, out, words, cap(wg_staged, T)), run<LE1, LE2, LC, 7, true>(in, out, words, cap(wg_staged, T)));
    fflush(stdout);
}

int main() {
    const size_t words = (size_t)1 << 30;                     // 4 GiB in, 4 GiB out
    unsigned *in = nullptr, *out = nullptr;
    CK(hipMalloc((void**)&in, words * 4));
    CK(hipMalloc((void**)&out, words * 4));
    CK(hipMemset(in, 1, words * 4));
    CK(hipMemset(out, 0, words * 4));
    printf("# two dummy steps of W products each per element; 7 + 7 products = 70 VALU instructions per element (a 1024-point single pass: ~45)\n");
    shape<4, 3, 5>(in, out, words);   // n = 128: the library's (4,3,5)
    shape<4, 4, 4>(in, out, words);   // n = 256: (4,4,4)
    shape<5, 4, 4>(in, out, words);   // n = 512: (5,4,4)
    shape<5, 5, 3>(in, out, words);   // n = 1024: (5,5,3)
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
