// Microbenchmark: an encoder stage lives in the SECOND HALF of the decoder's concat buffer (row stride 2C floats, columns [C, 2C) used), so
// its kernels touch only every other C x 4-byte piece of memory.  Does HBM deliver less when only half of each row is touched (channel
// interleaving)?  Streams the same number of useful bytes three ways: compact rows, the upper half of double-width rows, and the lower half.
//   hipcc --offload-arch=gfx950 -O3 halfrow.hip -o halfrow && ./halfrow
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: compact rows of RB bytes; 1: upper half of rows of 2*RB bytes; 2: lower half.  WRITE 0/1/2: read, write, read-modify-write
template <int MODE, int WRITE>
__global__ __launch_bounds__(256) void walk(u32x4* __restrict__ buf, unsigned* sink, long long rows, int rb) {
    const int ppr = rb / 16;                                             // 16-byte pieces per useful row
    const long long total = rows * ppr;
    u32x4 acc = {0, 0, 0, 0};
    for (long long q0 = (long long)blockIdx.x * 2048 + threadIdx.x; q0 < total; q0 += (long long)gridDim.x * 2048) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long q = q0 + u * 256;
            if (q >= total) break;
            const long long r = q / ppr; const int p = (int)(q - r * ppr);
            long long off = MODE == 0 ? r * rb + p * 16 : r * 2LL * rb + (MODE == 1 ? rb : 0) + p * 16;
            u32x4* ptr = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(buf) + off);
            if (WRITE == 1) *ptr = u32x4{(unsigned)q, 1u, 2u, 3u};
            else if (WRITE == 2) { u32x4 v = *ptr; v[0] += 1; *ptr = v; }
            else acc += *ptr;
        }
    }
    if (WRITE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 0x12345u) sink[blockIdx.x] = acc[0];
}

int main() {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    unsigned* sink; (void)hipMalloc(&sink, 1 << 22);
    const size_t useful = 1ull << 30;                                     // 1 GiB of useful bytes per pass
    u32x4* buf; (void)hipMalloc(&buf, 2 * useful); (void)hipMemset(buf, 1, 2 * useful);
    for (int rb : {128, 256, 512, 1024, 2048}) {
        const long long rows = (long long)(useful / rb);
        auto run = [&](const char* name, auto kern) {
            float best = 1e30f;
            for (int r = 0; r < 4; ++r) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(8192), dim3(256), 0, 0, buf, sink, rows, rb);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (r > 0 && ms < best) best = ms;
            }
            printf("row %4d B  %-34s %7.3f ms  %7.1f GB/s (useful bytes)\n", rb, name, best, useful / 1e9 / best * 1e3);
        };
        run("read  compact", walk<0, 0>); run("read  upper half of 2x rows", walk<1, 0>); run("read  lower half of 2x rows", walk<2, 0>);
        run("write compact", walk<0, 1>); run("write upper half of 2x rows", walk<1, 1>);
        run("rmw   compact", walk<0, 2>); run("rmw   upper half of 2x rows", walk<1, 2>);
    }
    return 0;
}
