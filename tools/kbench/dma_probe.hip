// Probe: does an LDS-DMA piece reach LDS offsets >= 64 KiB through M0?  (development tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(64) void probe(const unsigned *src, unsigned *out, unsigned ldsoff) {
    __shared__ __attribute__((aligned(16))) char lds[160 * 1024];
    const unsigned lane16 = threadIdx.x * 16;
    for (int i = threadIdx.x; i < 160 * 256; i += 64) ((unsigned *)lds)[i] = 0xdead0000u;
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    const unsigned dst = lds0 + ldsoff;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lane16), "s"(src) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // report where the 256 dwords landed
    for (int i = threadIdx.x; i < 160 * 256; i += 64) {
        unsigned v = ((unsigned *)lds)[i];
        if (v != 0xdead0000u) out[v & 255] = i * 4;  // byte offset where source dword (v & 255) was found
    }
}
int main() {
    unsigned h[256], *d, *o;
    for (int i = 0; i < 256; ++i) h[i] = 0x12340000u | i;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    unsigned offs[] = {0, 32768, 65536, 100 * 1024, 159 * 1024};
    for (unsigned off : offs) {
        hipMemset(o, 0xff, 1024);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, off);
        hipDeviceSynchronize();
        unsigned r[256];
        hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        printf("dst offset %6u: dword0 at %u, dword1 at %u, dword4 at %u, dword255 at %u\n", off, r[0], r[1], r[4], r[255]);
    }
    return 0;
}
