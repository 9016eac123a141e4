// Microbenchmark: how fast can every workgroup of a launch pull its 64 KiB "window" of an f32 stream at kernel start?
// Mirrors phase 0 of attn_block (C = 256: 64 rows x 1 KiB; a window = 8 segments of 8 KiB, 64 KiB apart) and times it with
// s_memtime inside the kernel and with events outside.  Variants: row pattern (window / contiguous), loads per thread in
// flight, workgroups per CU, nontemporal / LDS-DMA loads.   hipcc --offload-arch=gfx950 -O3 burst.hip -o burst && ./burst
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U>
__global__ __launch_bounds__(256, 2) void burst(const float* __restrict__ x, float* out, unsigned long long* cyc, int W, int nWc, int nW, int lds_pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bw = blockIdx.x;
    const int b = bw / nW, wi = bw - b * nW, wr = wi / nWc, wc = wi - wr * nWc;
    const unsigned long long t0 = __builtin_readcyclecounter();
    f32x4 acc = {0, 0, 0, 0};
    constexpr int C = 256;
#pragma unroll 1
    for (int r0 = 0; r0 < 64; r0 += 4 * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = r0 + u * 4 + wave;                 // row of the window
            size_t tok;
            if (MODE == 0 || MODE == 2) tok = (size_t)(b * W + (wr * 8 + (t >> 3))) * W + wc * 8 + (t & 7);   // window pattern
            else tok = (size_t)bw * 64 + t;                                                                     // contiguous 64 KiB
            const float* p = x + tok * C + lane * 4;
            if (MODE == 2) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
            else v[u] = *reinterpret_cast<const f32x4*>(p);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[bw] = acc[0];
    if (tid == 0) cyc[bw] = t1 - t0;
}

int main() {
    const int B = 16, W = 64, C = 256, nWc = W / 8, nW = nWc * nWc;
    const size_t n = (size_t)B * W * W * C;
    float* x; float* out; unsigned long long* cyc;
    hipMalloc(&x, n * 4); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8 * 4096);
    hipMemset(x, 0, n * 4);
    std::vector<unsigned long long> h(4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int lds) {
        const int grid = B * nW;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, x, out, cyc, W, nWc, nW, 0);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, x, out, cyc, W, nWc, nW, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, 8 * grid, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
        printf("%-44s %7.1f us/launch  %6.2f TB/s  mean workgroup burst %7.0f cycles (64 KiB) = %.1f B/clk/WG\n", name, ms * 100, n * 4.0 / (ms / 10 * 1e-3) / 1e12,
               s / grid, 65536.0 / (s / grid));
    };
    run("window rows, 16 loads in flight, 2 WG/CU", burst<0, 16>, 60 * 1024);
    run("window rows, 16 in flight, 1 WG/CU", burst<0, 16>, 100 * 1024);
    run("window rows, 8 in flight x 2 passes, 2 WG/CU", burst<0, 8>, 60 * 1024);
    run("window rows, 4 in flight x 4 passes, 2 WG/CU", burst<0, 4>, 60 * 1024);
    run("contiguous 64 KiB, 16 in flight, 2 WG/CU", burst<1, 16>, 60 * 1024);
    run("contiguous 64 KiB, 16 in flight, 4 WG/CU", burst<1, 16>, 36 * 1024);
    run("window rows, nontemporal, 2 WG/CU", burst<2, 16>, 60 * 1024);
    run("window rows, 16 in flight, 4 WG/CU", burst<0, 16>, 36 * 1024);
    return 0;
}
