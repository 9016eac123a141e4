// Prototype for VERDICT r04 "next" 1: the fc1 phase of attn_block (LN2 output tile in LDS -> h1 = GELU(Z W1^T + b1), the largest phase of the kernel:
// 35 K of the 89 K cycles of a C = 256 window) as a ONE-window (64-row) and as a TWO-window (128-row) workgroup.  In the two-window form every weight
// fragment fetched from L2 feeds two windows' MFMAs: half the L2 -> CU stream per token, twice the accumulators per wave.  Same everything else:
// fragment-major weights streamed through a register ring, operand fragments from LDS, weights as the MFMA A operand, sigmoid-form GELU, bf16 stores.
//   C = 256: 4 waves per workgroup, 2 workgroups per CU in both forms (as attn_block<256,256>);  C = 512: 8 waves, 1 workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 fc1_2w.hip -o fc1_2w && ./fc1_2w
// Prints, per (C, tokens): time of both forms with and without the weight loads (NOW = 1: fragments from the lane id), TFLOP/s, ns per window.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const bf16x2_t v = {static_cast<__bf16>(lo), static_cast<__bf16>(hi)};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void mma(f32x4& d, u32x4 a, u32x4 b) {
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), d, 0, 0, 0);
}
__device__ __forceinline__ float gelu(float x) {      // the sigmoid form of the bf16 kernels (uf_common.h)
    const float u = x * (x * x * -0.10294324f + -2.3022081985f);
    return x * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(u) + 1.0f);
}

// Z: bf16 [M][C] token-major (stands for the LN2 output), W1: fragment-major bf16 [4C][C], h1: bf16 [M][4C]
template <int C, int ROWS, int NT, int RING, int NOW>
__global__ __launch_bounds__(NT, C == 256 ? 2 : 1) void fc1_kernel(const uint16_t* __restrict__ Z, const uint16_t* __restrict__ W1, const float* __restrict__ b1,
                                                                 uint16_t* __restrict__ h1) {
    constexpr int WAVES = NT / 64, KS = C / 32, N4 = 4 * C, UNITS = N4 / 64, RT = ROWS / 16, SA = C * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const size_t row0 = (size_t)blockIdx.x * ROWS;
    // tile -> LDS (stands for phase 2's LN2 store; 16-byte pieces)
    for (int i = tid; i < ROWS * (C / 8); i += NT) {
        const int r = i / (C / 8), c8 = i - r * (C / 8);
        *reinterpret_cast<u32x4*>(smem + r * SA + c8 * 16) = *reinterpret_cast<const u32x4*>(Z + (row0 + r) * C + c8 * 8);
    }
    __syncthreads();
#pragma unroll 1
    for (int u = wave; u < UNITS; u += WAVES) {
        f32x4 acc[4][RT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < RT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 wf[RING][4];
        auto wload = [&](int ks, int slot) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint16_t* p = W1 + (((size_t)(u * 4 + i) * KS + ks) * 64 + lane) * 8;
                if (NOW) { const unsigned v = 0x3c003c00u ^ ((unsigned)(uintptr_t)p & 0x00ff00ffu); wf[slot][i] = u32x4{v, v, v, v}; }
                else wf[slot][i] = *reinterpret_cast<const u32x4*>(p);
            }
        };
#pragma unroll
        for (int s = 0; s < RING - 1; ++s)
            if (s < KS) wload(s, s);
        f32x4 bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4*>(b1 + u * 64 + i * 16 + fg * 4);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + RING - 1 < KS) wload(ks + RING - 1, (ks + RING - 1) % RING);
            u32x4 af[RT];
#pragma unroll
            for (int j = 0; j < RT; ++j) af[j] = *reinterpret_cast<const u32x4*>(smem + (j * 16 + fr) * SA + (ks * 32 + fg * 8) * 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < RT; ++j) mma(acc[i][j], wf[ks % RING][i], af[j]);      // weight as A: lane = 4 channels of one token
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            uint16_t* orow = h1 + (row0 + j * 16 + fr) * N4 + u * 64 + fg * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = acc[i][j] + bv[i];
                *reinterpret_cast<u32x2*>(orow + i * 16) = u32x2{pack2(gelu(v[0]), gelu(v[1])), pack2(gelu(v[2]), gelu(v[3]))};
            }
        }
    }
}

template <int C, int ROWS, int NT, int RING, int NOW>
float run(const uint16_t* Z, const uint16_t* W1, const float* b1, uint16_t* h1, int M, int reps) {
    auto k = fc1_kernel<C, ROWS, NT, RING, NOW>;
    const int smem = ROWS * (C * 2 + 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(M / ROWS), dim3(NT), smem, 0, Z, W1, b1, h1);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(M / ROWS), dim3(NT), smem, 0, Z, W1, b1, h1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); exit(1); }
    return 1e3f * ms / reps;
}

template <int C, int NT>
void bench(int M) {
    const size_t nz = (size_t)M * C, nw = (size_t)4 * C * C, nh = (size_t)M * 4 * C;
    std::vector<uint16_t> hz(nz), hw(nw);
    srand(7);
    for (auto& v : hz) v = (uint16_t)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));      // bf16 in +-[0.5, 1)
    for (auto& v : hw) v = (uint16_t)(0x3c00 + (rand() & 0xff) + ((rand() & 1) << 15));      // +-[0.0078, 0.0156)
    uint16_t *Z, *W1, *h1; float* b1;
    hipMalloc(&Z, nz * 2); hipMalloc(&W1, nw * 2); hipMalloc(&h1, nh * 2); hipMalloc(&b1, 4 * C * 4);
    hipMemcpy(Z, hz.data(), nz * 2, hipMemcpyHostToDevice); hipMemcpy(W1, hw.data(), nw * 2, hipMemcpyHostToDevice); hipMemset(b1, 0, 4 * C * 4);
    const double flop = 2.0 * M * C * 4.0 * C;
    // one window per workgroup: ring of 5 as Fc1Walk at KS >= 8; two windows: ring of 3 (128 accumulator registers leave room for 48)
    const float a = run<C, 64, NT, 5, 0>(Z, W1, b1, h1, M, 20), an = run<C, 64, NT, 5, 1>(Z, W1, b1, h1, M, 20);
    const float b = run<C, 128, NT, 3, 0>(Z, W1, b1, h1, M, 20), bn = run<C, 128, NT, 3, 1>(Z, W1, b1, h1, M, 20);
    const float a3 = run<C, 64, NT, 3, 0>(Z, W1, b1, h1, M, 20);
    printf("C=%d tokens=%d (%d windows): one-window form %7.1f us (%6.1f TFLOP/s; ring 3: %7.1f us; no weight loads %7.1f us) | two-window form %7.1f us (%6.1f TFLOP/s; no weight loads %7.1f us) | two / one = %.3f\n",
           C, M, M / 64, a, flop / a / 1e6, a3, an, b, flop / b / 1e6, bn, b / a);
    hipFree(Z); hipFree(W1); hipFree(h1); hipFree(b1);
}

int main() {
    bench<256, 256>(65536);      // dec1 at batch 16: 1024 windows
    bench<256, 256>(131072);     // dec1 at batch 32
    bench<512, 512>(16384);      // dec0 at batch 16: 256 windows = 256 / 128 workgroups for 256 CUs
    bench<512, 512>(32768);      // dec0 at batch 32
    bench<512, 512>(65536);      // dec0 at batch 64
    return 0;
}
