// Probe of ds_read_b64_tr_b16 on gfx950: every lane addresses its own 8-byte chunk of LDS (chunk q of the lane holds the 16-bit values
// 4q..4q+3); print which (source lane, element) each destination lane receives.   hipcc --offload-arch=gfx950 -O3 trread.hip -o trread
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4];
    const int l = threadIdx.x;
    for (int e = 0; e < 4; ++e) lds[l * 4 + e] = (unsigned short)(l * 4 + e);
    __syncthreads();
    unsigned lo, hi;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds + l * 8;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(unsigned long long*)&lo) : "v"(addr) : "memory");
    unsigned long long v; asm volatile("" : "=v"(v) : "0"(*(unsigned long long*)&lo));
    (void)hi;
    unsigned long long r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)(r >> (16 * e));
}
int main() {
    unsigned short* d; hipMalloc(&d, 512);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf("  (src lane %2d, elem %d)", h[l * 4 + e] / 4, h[l * 4 + e] % 4);
        printf("\n");
    }
    return 0;
}
