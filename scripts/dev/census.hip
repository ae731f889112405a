// Where do the workgroups of a 1,024-block launch with two resident blocks per CU land, and in which wave slots?
// (placement is a speed matter only; used to design the de-phased start of the fused ResBlock kernels)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256, 2) void census(unsigned *out, unsigned long long *t, int spin) {
    extern __shared__ float smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) {
        out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = hw;
        out[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = xcc;
        t[blockIdx.x * 4 + threadIdx.x / 64] = t0;
    }
    smem[threadIdx.x] = hw;
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if (smem[(threadIdx.x + 1) % 256] == 12345.f) out[0] = 1;
}
int main() {
    const int G = 1024;
    unsigned *d; unsigned long long *dt;
    hipMalloc(&d, G * 4 * 2 * 4); hipMalloc(&dt, G * 4 * 8);
    hipFuncSetAttribute((const void *)census, hipFuncAttributeMaxDynamicSharedMemorySize, 75000);
    hipLaunchKernelGGL(census, dim3(G), dim3(256), 75000, 0, d, dt, 100000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 8); std::vector<unsigned long long> ht(G * 4);
    hipMemcpy(h.data(), d, G * 32, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, G * 32, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull; for (auto v : ht) tmin = v < tmin ? v : tmin;
    for (int b = 0; b < G; b += (b < 40 ? 1 : 37)) {
        printf("blk %4d t=%7llu:", b, ht[b * 4] - tmin);
        for (int w = 0; w < 4; ++w) { unsigned hw = h[(b * 4 + w) * 2];
            printf("  [xcc %u se %u sh %u cu %2u simd %u wave %u]", h[(b * 4 + w) * 2 + 1] & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15); }
        printf("\n");
    }
    // how many of the first 512 blocks have wave slot != 0 for wave 0?
    int n1 = 0, n2 = 0; for (int b = 0; b < 512; ++b) { n1 += (h[b * 8] & 15) != 0; } for (int b = 256; b < 512; ++b) n2 += (h[b * 8] & 15) != 0;
    printf("first 512 blocks with wave slot != 0: %d (of them in 256..511: %d)\n", n1, n2);
    return 0;
}
