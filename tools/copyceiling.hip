// Stream-ceiling probe (diagnostic, not part of the library): what does the memory fabric of THIS box move when nothing but
// loads and stores is issued?  VERDICT r2 item 1: MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy, round 2's own probe
// (tools/membench.hip) topped out at 4.9-5.4 TB/s while the fold kernel moved 5.7.  This sweeps every knob a plain stream has:
//   * hipMemcpyDtoDAsync (the runtime's blit kernel) as the outside reference;
//   * 16-byte copies: grid = 256 x {1,2,4,8,16} workgroups, 256/512/1024 threads, 1/2/4/8 independent loads in flight per lane,
//     plain and non-temporal, chunk-interleaved (grid-stride) and partitioned (one contiguous region per workgroup) orders;
//   * read-only, write-only, 1:1 and 2:1 read:write mixes, in-place;
//   * footprints from cache-resident to 4 GiB + 4 GiB.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/copyceiling tools/copyceiling.hip ; rates are (bytes read + bytes written) / time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ v4u ld16(const v4u* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <int NT> __device__ __forceinline__ void st16(v4u* p, v4u v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// MODE 0: copy (1:1).  MODE 1: read only.  MODE 2: write only.  MODE 3: two reads, one write (the fold's mix; `in` is read at
// i and i + n16, n16 outputs).  MODE 4: in place (out == in region, read-modify-write).
// ORDER 0: chunk c of U * blockDim 16-byte words goes to workgroup c mod grid (grid-stride).  ORDER 1: every workgroup owns one
// contiguous region of n16 / grid words.
template <int U, int NT, int MODE, int ORDER>
__global__ void stream_kernel(const v4u* __restrict__ in, v4u* __restrict__ out, size_t n16) {
    const size_t chunk = (size_t)U * blockDim.x;
    const size_t nchunks = n16 / chunk;
    size_t c, cend, cstep;
    if (ORDER == 0) { c = blockIdx.x; cend = nchunks; cstep = gridDim.x; }
    else { const size_t per = nchunks / gridDim.x; c = per * blockIdx.x; cend = c + per; cstep = 1; }
    v4u acc = {0, 0, 0, 0};
    for (; c < cend; c += cstep) {
        const size_t base = c * chunk + threadIdx.x;
        v4u x[U], y[U];
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld16<NT>(in + base + (size_t)u * blockDim.x);
            if (MODE == 3) {
#pragma unroll
                for (int u = 0; u < U; ++u) y[u] = ld16<NT>(in + n16 + base + (size_t)u * blockDim.x);
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= x[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v4u v;
                if (MODE == 2) { v = acc; v.x = (unsigned)base + u; }
                else if (MODE == 3) v = x[u] + y[u];
                else v = x[u] + 1u;
                st16<NT>(out + base + (size_t)u * blockDim.x, v);
            }
        }
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0xDEADBEEFu) out[threadIdx.x] = acc;
}

// 4-byte lanes (what the NTT passes issue): same chunking with one dword per lane per access
template <int U, int NT>
__global__ void stream4_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out, size_t n) {
    const size_t chunk = (size_t)U * blockDim.x;
    const size_t nchunks = n / chunk;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        unsigned x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = NT ? __builtin_nontemporal_load(in + base + (size_t)u * blockDim.x) : in[base + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(x[u] + 1u, out + base + (size_t)u * blockDim.x); else out[base + (size_t)u * blockDim.x] = x[u] + 1u;
        }
    }
}

template <class F> static float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 3; ++r) {  // best of three groups: the first groups of a run see the chip ramping its clocks
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms / reps);
    }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return *std::min_element(t.begin(), t.end());
}

static const v4u* g_in; static v4u* g_out;

template <int U, int NT, int MODE, int ORDER>
static double run(int grid, int block, size_t bytes_in, int reps = 4) {
    // bytes_in = bytes of the region every access stream covers (the input of a copy)
    const size_t n16 = bytes_in / 16;
    const v4u* in = g_in; v4u* out = (MODE == 4) ? (v4u*)g_in : g_out;
    const float ms = timeit([&] { hipLaunchKernelGGL((stream_kernel<U, NT, MODE, ORDER>), dim3(grid), dim3(block), 0, 0, in, out, n16); }, reps);
    const double moved = (MODE == 0 || MODE == 4) ? 2.0 * bytes_in : (MODE == 3 ? 3.0 * bytes_in : 1.0 * bytes_in);
    return moved / ms / 1e9;
}

template <int NT, int MODE, int ORDER>
static void sweep(const char* name, size_t bytes_in) {
    printf("%s, %zu MiB per stream: TB/s by (grid x block) and 16-byte loads in flight per lane U = 1 / 2 / 4 / 8\n", name, bytes_in >> 20);
    double best = 0; int bg = 0, bb = 0, bu = 0;
    for (int mult : {1, 2, 4, 8, 16}) for (int block : {256, 512, 1024}) {
        const int grid = 256 * mult;
        if ((size_t)grid * block > (size_t)256 * 2048 * 2) continue;  // more threads than the chip holds at once: skip the largest
        const double r1 = run<1, NT, MODE, ORDER>(grid, block, bytes_in), r2 = run<2, NT, MODE, ORDER>(grid, block, bytes_in),
                     r4 = run<4, NT, MODE, ORDER>(grid, block, bytes_in), r8 = run<8, NT, MODE, ORDER>(grid, block, bytes_in);
        printf("  grid %5d x %4d : %.2f  %.2f  %.2f  %.2f\n", grid, block, r1, r2, r4, r8);
        const double rs[4] = {r1, r2, r4, r8};
        for (int k = 0; k < 4; ++k) if (rs[k] > best) { best = rs[k]; bg = grid; bb = block; bu = 1 << k; }
    }
    printf("  best: %.2f TB/s at grid %d x %d, U = %d\n", best, bg, bb, bu);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const size_t GiB = (size_t)1 << 30;
    const size_t big = quick ? GiB : 4 * GiB;
    void *a, *b;
    CK(hipMalloc(&a, 2 * big)); CK(hipMalloc(&b, big));   // a: 2 x so that the 2:1 mix has its second input stream
    CK(hipMemset(a, 1, 2 * big)); CK(hipMemset(b, 2, big)); CK(hipDeviceSynchronize());
    g_in = (const v4u*)a; g_out = (v4u*)b;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs, memory clock %d kHz, bus %d bits\n", prop.name, prop.multiProcessorCount, prop.memoryClockRate, prop.memoryBusWidth);

    for (size_t bytes : {big, GiB, GiB / 4}) {
        const float ms = timeit([&] { CK(hipMemcpyDtoDAsync(b, a, bytes, 0)); }, 4);
        printf("hipMemcpyDtoDAsync %zu MiB -> %zu MiB : %.3f ms  %.2f TB/s (r+w)\n", bytes >> 20, bytes >> 20, ms, 2.0 * bytes / ms / 1e9);
    }
    sweep<0, 0, 0>("copy 1:1 plain, grid-stride chunks", big);
    sweep<1, 0, 0>("copy 1:1 non-temporal, grid-stride chunks", big);
    sweep<1, 0, 1>("copy 1:1 non-temporal, one contiguous region per workgroup", big);
    sweep<1, 1, 0>("read only, non-temporal", big);
    sweep<0, 1, 0>("read only, plain", big);
    sweep<1, 2, 0>("write only, non-temporal", big);
    sweep<0, 2, 0>("write only, plain", big);
    sweep<1, 3, 0>("2 reads : 1 write, non-temporal", big);
    sweep<1, 4, 0>("in place (read-modify-write of one buffer), non-temporal", big);
    sweep<0, 4, 0>("in place, plain", big);
    // footprint dependence of the best plain form (Infinity Cache: 256 MiB)
    for (size_t mib : {32, 64, 128, 256, 512, 1024, 2048}) {
        if (mib * (1u << 20) > big) break;
        const double p = run<4, 0, 0, 0>(2048, 256, mib << 20, 16), q = run<4, 1, 0, 0>(2048, 256, mib << 20, 16);
        printf("copy 1:1 footprint %4zu MiB -> %4zu MiB (grid 2048 x 256, U = 4): plain %.2f  nt %.2f TB/s\n", mib, mib, p, q);
    }
    // 4-byte lanes
    printf("4-byte lanes, %zu MiB (grid x block; U = 1 / 4 / 8 / 16 / 32 dwords in flight per lane), plain | nt\n", big >> 20);
    for (int mult : {1, 2, 4, 8}) for (int block : {256, 1024}) {
        const int grid = 256 * mult;
        if ((size_t)grid * block > (size_t)256 * 2048) continue;
        const size_t n = big / 4;
        auto t4 = [&](auto kern) { const float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, (const unsigned*)a, (unsigned*)b, n); }, 4); return 2.0 * big / ms / 1e9; };
        printf("  grid %5d x %4d : %.2f %.2f %.2f %.2f %.2f | %.2f %.2f %.2f %.2f %.2f\n", grid, block,
               t4(stream4_kernel<1, 0>), t4(stream4_kernel<4, 0>), t4(stream4_kernel<8, 0>), t4(stream4_kernel<16, 0>), t4(stream4_kernel<32, 0>),
               t4(stream4_kernel<1, 1>), t4(stream4_kernel<4, 1>), t4(stream4_kernel<8, 1>), t4(stream4_kernel<16, 1>), t4(stream4_kernel<32, 1>));
        fflush(stdout);
    }
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}
