// Prints what v_permlane16_swap / v_permlane32_swap do to lane ids (development probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
    auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"swap16 first ", "swap16 second", "swap32 first ", "swap32 second"};
    for (int q = 0; q < 4; ++q) {
        printf("%s:", names[q]);
        for (int g = 0; g < 4; ++g) printf("  rows%d: %u..%u", g, h[64 * q + 16 * g], h[64 * q + 16 * g + 15]);
        printf("\n");
    }
    return 0;
}
