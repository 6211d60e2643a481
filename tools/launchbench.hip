// Launch-floor probe (diagnostic, not part of the library): what does a kernel boundary cost on this box, and what would a
// grid-wide barrier inside one kernel cost instead?  Build: hipcc --offload-arch=gfx950 -O3 -o build/launchbench tools/launchbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void empty_kernel(uint32_t* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

// touch `words` words per launch (read + write), like a pass over one small transform
__global__ void touch_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t words) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) out[i] = in[i] + 1u;
}

// `rounds` grid barriers in one launch: counter-based, agent-scope release / acquire (what a fused column -> row kernel would need)
__global__ void barrier_kernel(uint32_t* counter, uint32_t rounds, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t words) {
    for (uint32_t r = 0; r < rounds; ++r) {
        if (words) for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) out[i] = in[i] + r;
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);
            const uint32_t target = (r + 1) * gridDim.x;
            while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1000.f;   // microseconds
}

int main() {
    uint32_t *in, *out, *counter;
    const uint32_t words = 1u << 20;
    CK(hipMalloc(&in, words * 4)); CK(hipMalloc(&out, words * 4)); CK(hipMalloc(&counter, 4));
    CK(hipMemset(in, 1, words * 4));
    for (int wgs : {1, 256, 512, 1024}) {
        for (int thr : {256, 1024}) {
            float us = timeit([&] { hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(thr), 0, 0, (uint32_t*)nullptr); }, 2000);
            printf("empty kernel   %5d x %4d, back to back : %6.2f us per launch\n", wgs, thr, us);
        }
    }
    for (int wgs : {256, 512, 1024}) {
        float us = timeit([&] { hipLaunchKernelGGL(touch_kernel, dim3(wgs), dim3(1024), 0, 0, in, out, words); }, 2000);
        printf("touch 4 MiB r+w %4d x 1024, back to back : %6.2f us per launch\n", wgs, us);
    }
    {   // the same chain of dependent launches replayed from a captured graph
        hipStream_t st; CK(hipStreamCreate(&st));
        hipGraph_t g; hipGraphExec_t ge;
        const int chain = 200;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(1024), 0, st, i & 1 ? out : in, i & 1 ? in : out, words);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, st));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("touch 4 MiB r+w  256 x 1024, graph of %d   : %6.2f us per launch\n", chain, ms * 1000.f / (10 * chain));
    }
    for (int wgs : {128, 256}) {   // co-resident by construction: one 1024-thread workgroup per CU at most
        for (uint32_t w : {0u, words}) {
            const uint32_t rounds = 64;
            float us = timeit([&] {
                hipMemsetAsync(counter, 0, 4, 0);
                hipLaunchKernelGGL(barrier_kernel, dim3(wgs), dim3(1024), 0, 0, counter, rounds, in, out, w);
            }, 50);
            float base = timeit([&] {
                hipMemsetAsync(counter, 0, 4, 0);
                hipLaunchKernelGGL(barrier_kernel, dim3(wgs), dim3(1024), 0, 0, counter, 1u, in, out, w);
            }, 50);
            printf("grid barrier   %4d x 1024 %s : %6.2f us per round (64 rounds %.1f us, 1 round %.1f us)\n", wgs, w ? "+ 4 MiB r+w" : "           ",
                   (us - base) / (rounds - 1), us, base);
        }
    }
    return 0;
}
