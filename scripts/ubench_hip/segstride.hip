// Microbenchmark: does HBM care HOW the 4C-wide hidden tensor h1 is walked?  The fused block kernels move it channel chunk by channel
// chunk: a workgroup owns 64 tokens and, per step, touches ONE 128-byte piece (64 channels x 2 B) of each token's row -- pieces that
// sit ROWB = 4C x 2 bytes apart in the token-major layout h1[M][4C].  A chunk-major ("planar") layout h1[4C/64][M][64] makes the
// 64 pieces of a step 8 KiB contiguous.  This program times both walks, reading and writing, with every byte moved exactly once and
// the same parallelism (workgroups x loads in flight), against a plain contiguous stream.
//   hipcc --offload-arch=gfx950 -O3 segstride.hip -o segstride && ./segstride
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: token-major rows (piece stride ROWB), 1: planar (piece stride 128 B, chunk stride M * 128), 2: contiguous stream
// WRITE 0: read (sum into a register, defeat DCE), 1: write
template <int MODE, int WRITE>
__global__ __launch_bounds__(256) void walk(u32x4* __restrict__ buf, unsigned* sink, long long M, int chunks) {
    const int tid = threadIdx.x;
    const long long tok0 = (long long)blockIdx.x * 64;                  // this workgroup's 64 tokens
    const int piece = tid & 7, trow = tid >> 3;                          // 8 lanes x 16 B = one 128-byte piece; 32 tokens per pass
    const long long rowb = (long long)chunks * 128;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll 4
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const long long tok = tok0 + trow + 32 * half;
            long long off;                                               // byte offset
            if (MODE == 0) off = tok * rowb + (long long)c * 128 + piece * 16;
            else if (MODE == 1) off = ((long long)c * M + tok) * 128 + piece * 16;
            else off = ((long long)blockIdx.x * chunks + c) * 8192 + (half * 256 + tid) * 16;
            u32x4* p = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(buf) + off);
            if (WRITE) *p = u32x4{(unsigned)tid, (unsigned)c, 0u, 1u};
            else { const u32x4 v = *p; acc += v; }
        }
    }
    if (!WRITE && acc[0] + acc[1] + acc[2] + acc[3] == 0x12345u) sink[blockIdx.x] = acc[0];
}

int main() {
    const int Cs[] = {64, 128, 256};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned* sink; hipMalloc(&sink, 1 << 22);
    for (int C : Cs) {
        const int chunks = 4 * C / 64;
        const long long M = (1LL << 30) / (4LL * C * 2);                // 1 GiB tensor: far beyond the 256 MiB Infinity Cache
        u32x4* buf; hipMalloc(&buf, (size_t)M * 4 * C * 2);
        hipMemset(buf, 1, (size_t)M * 4 * C * 2);
        const double gb = (double)M * 4 * C * 2 / 1e9;
        auto run = [&](const char* name, auto kern) {
            float best = 1e30f;
            for (int r = 0; r < 4; ++r) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3((unsigned)(M / 64)), dim3(256), 0, 0, buf, sink, M, chunks);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r > 0 && ms < best) best = ms;
            }
            printf("C=%3d (row %4d B) %-28s %7.3f ms  %7.1f GB/s\n", C, chunks * 128, name, best, gb / best * 1e3);
        };
        run("read  token-major pieces", walk<0, 0>);
        run("read  planar chunks", walk<1, 0>);
        run("read  contiguous", walk<2, 0>);
        run("write token-major pieces", walk<0, 1>);
        run("write planar chunks", walk<1, 1>);
        run("write contiguous", walk<2, 1>);
        hipFree(buf);
    }
    return 0;
}
