// Which XCD runs which workgroup?  (diagnostic)  Pass3::tile_order and the narrow-tile probe of tools/membench.hip assume that workgroups
// p, p + 8, p + 16, ... of a launch land on one XCD (round-robin dispatch over the 8 XCDs).  This reads XCC_ID in every workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/xccprobe tools/xccprobe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void probe(uint32_t* out) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = xcc & 0xFu;
}

int main() {
    for (int grid : {256, 512, 1024, 4096}) {
        for (int threads : {256, 1024}) {
            uint32_t* d;
            CK(hipMalloc(&d, grid * 4));
            hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), 0, 0, d);
            std::vector<uint32_t> h(grid);
            CK(hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost));
            int match = 0, hist[16] = {0};
            for (int b = 0; b < grid; ++b) { match += (int)(h[b] == (uint32_t)(b & 7)); ++hist[h[b]]; }
            printf("grid %5d x %4d: blockIdx & 7 == XCC_ID for %d of %d workgroups; per XCD:", grid, threads, match, grid);
            for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
            printf("\n");
            CK(hipFree(d));
        }
    }
    return 0;
}
