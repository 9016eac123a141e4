// Microbenchmark / prototype for DESIGN.md section 7.1, second half: the token tiles of the bf16 weight-gradient kernel (linear_wgrad2 in
// uf_bwd.hip: dW[n][k] = sum_m dY[m][n] X[m][k], 128 x 128 output tiles, token steps of 32, operands transposed on the way out of LDS by
// ds_read_b64_tr_b16) staged by
//   MODE 0  registers -> ds_write_b128 (what the library does)
//   MODE 1  LDS-DMA (global_load_lds_dwordx4): the library's XOR placement of the 16-byte pieces (piece p of row r at position
//           p ^ (rho(r) << 1)) is obtained by permuting which global piece a lane fetches
// No bias sums, full tiles only (N, K multiples of 128, M a multiple of 32 x chunks).  Partial tiles per token chunk, as in the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 wgrad_dma.hip -o wgrad_dma && ./wgrad_dma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ void dma_global_to_lds(const void* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(src) : "memory");
}
__device__ __forceinline__ int rho_of(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ unsigned wg2_off(int row, int chunk) { return (unsigned)(row * 256 + ((chunk ^ (rho_of(row) << 2)) << 3)); }

template <int MODE, int TOK>                                  // TOK = tokens per step (barrier): 32 (the library) or 64 (MODE 1 only)
__global__ __launch_bounds__(256, 2) void wgrad_proto(const uint16_t* __restrict__ dY, const uint16_t* __restrict__ X, float* __restrict__ ws_w, int M, int N, int K, int S) {
    constexpr int TB = TOK * 256;                              // bytes of one operand tile
    __shared__ __attribute__((aligned(1024))) char Ys[2][TB];
    __shared__ __attribute__((aligned(1024))) char Xs[2][TB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int k_tiles = K / 128, tiles = (N / 128) * k_tiles;
    const int xcd = (int)blockIdx.x & 7, seq = (int)blockIdx.x >> 3;
    const int tile = seq % tiles, chunk = (seq / tiles) * 8 + xcd;
    if (chunk >= S) return;
    const int n0 = (tile / k_tiles) * 128, k0 = (tile % k_tiles) * 128;
    const int wn = wave >> 1, wk = wave & 1;
    const int steps_all = M / TOK;
    const int s0 = (int)((long long)steps_all * chunk / S), s1 = (int)((long long)steps_all * (chunk + 1) / S);
    const int prow = tid >> 4, pseg = tid & 15;
    const uint16_t* ysrc = dY + n0 + pseg * 8;
    const uint16_t* xsrc = X + k0 + pseg * 8;
    u32x4 ry[2], rx[2];
    auto fetch = [&](int s) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = s * 32 + prow + 16 * q;
            ry[q] = *reinterpret_cast<const u32x4*>(ysrc + (size_t)m * N);
            rx[q] = *reinterpret_cast<const u32x4*>(xsrc + (size_t)m * K);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = prow + 16 * q;
            *reinterpret_cast<u32x4*>(Ys[buf] + wg2_off(row, pseg * 2)) = ry[q];
            *reinterpret_cast<u32x4*>(Xs[buf] + wg2_off(row, pseg * 2)) = rx[q];
        }
    };
    const unsigned ybase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&Ys[0][0];
    const unsigned xbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&Xs[0][0];
    auto dma_step = [&](int s, int buf) {                   // wave w: rows [TOK/4 w, TOK/4 (w + 1)) of both tiles, instructions of 4 rows each
#pragma unroll
        for (int q = 0; q < TOK / 16; ++q) {
            const int rb = (wave * (TOK / 16) + q) * 4, row = rb + (lane >> 4);
            const int p = (lane & 15) ^ (rho_of(row & 31) << 1);
            const size_t m = (size_t)s * TOK + row;
            dma_global_to_lds(dY + m * N + n0 + p * 8, ybase + buf * TB + rb * 256);
            dma_global_to_lds(X + m * K + k0 + p * 8, xbase + buf * TB + rb * 256);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int trow = 8 * fg + (fr >> 2);
    unsigned yaddr[4], xaddr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        yaddr[i] = ybase + wg2_off(trow, (wn * 64 + i * 16) / 4 + (fr & 3));
        xaddr[i] = xbase + wg2_off(trow, (wk * 64 + i * 16) / 4 + (fr & 3));
    }
    if (s0 < s1) {
        if (MODE == 0) { fetch(s0); stash(0); }
        else { dma_step(s0, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const int buf = (s - s0) & 1;
        if (s + 1 < s1) { if (MODE == 0) fetch(s + 1); else dma_step(s + 1, buf ^ 1); }
#pragma unroll
        for (int sub = 0; sub < TOK / 32; ++sub) {
        u32x2 y0[4], y1[4], x0[4], x1[4];
        const unsigned bo = (unsigned)buf * (unsigned)TB + (unsigned)sub * 8192u;
        const unsigned ya0 = yaddr[0] + bo, ya1 = yaddr[1] + bo, ya2 = yaddr[2] + bo, ya3 = yaddr[3] + bo;
        const unsigned xa0 = xaddr[0] + bo, xa1 = xaddr[1] + bo, xa2 = xaddr[2] + bo, xa3 = xaddr[3] + bo;
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:1024\n\t"
            "ds_read_b64_tr_b16 %2, %17\n\tds_read_b64_tr_b16 %3, %17 offset:1024\n\t"
            "ds_read_b64_tr_b16 %4, %18\n\tds_read_b64_tr_b16 %5, %18 offset:1024\n\t"
            "ds_read_b64_tr_b16 %6, %19\n\tds_read_b64_tr_b16 %7, %19 offset:1024\n\t"
            "ds_read_b64_tr_b16 %8, %20\n\tds_read_b64_tr_b16 %9, %20 offset:1024\n\t"
            "ds_read_b64_tr_b16 %10, %21\n\tds_read_b64_tr_b16 %11, %21 offset:1024\n\t"
            "ds_read_b64_tr_b16 %12, %22\n\tds_read_b64_tr_b16 %13, %22 offset:1024\n\t"
            "ds_read_b64_tr_b16 %14, %23\n\tds_read_b64_tr_b16 %15, %23 offset:1024\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(y0[0]), "=&v"(y1[0]), "=&v"(y0[1]), "=&v"(y1[1]), "=&v"(y0[2]), "=&v"(y1[2]), "=&v"(y0[3]), "=&v"(y1[3]),
              "=&v"(x0[0]), "=&v"(x1[0]), "=&v"(x0[1]), "=&v"(x1[1]), "=&v"(x0[2]), "=&v"(x1[2]), "=&v"(x0[3]), "=&v"(x1[3])
            : "v"(ya0), "v"(ya1), "v"(ya2), "v"(ya3), "v"(xa0), "v"(xa1), "v"(xa2), "v"(xa3)
            : "memory");
        u32x4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = u32x4{y0[i][0], y0[i][1], y1[i][0], y1[i][1]};
            b[i] = u32x4{x0[i][0], x0[i][1], x1[i][0], x1[i][1]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        }
        if (s + 1 < s1) { if (MODE == 0) stash(buf ^ 1); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
    }
    float* wp = ws_w + (size_t)chunk * N * K;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + i * 16 + fg * 4 + r, k = k0 + wk * 64 + j * 16 + fr;
                wp[(size_t)n * K + k] = acc[i][j][r];
            }
}

static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main() {
    struct Shape { int M, N, K; const char* what; };
    const Shape shapes[] = {{131072, 1024, 256, "lin1 dec1"}, {131072, 256, 1024, "lin2 dec1"}, {32768, 2048, 512, "lin1 dec0"}, {32768, 512, 2048, "lin2 dec0"},
                            {131072, 768, 256, "qkv dec1"}, {524288, 512, 128, "lin1 dec2"}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Shape& s : shapes) {
        const size_t ny = (size_t)s.M * s.N, nx = (size_t)s.M * s.K;
        std::vector<uint16_t> hy(ny), hx(nx);
        uint32_t seed = 777u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hy) v = f2bf(rnd());
        for (auto& v : hx) v = f2bf(rnd());
        const int tiles = (s.N / 128) * (s.K / 128);
        int S = 512 / tiles; if (S > 256) S = 256; if (S < 1) S = 1;
        uint16_t *dY, *dX; float* dW;
        hipMalloc(&dY, ny * 2); hipMalloc(&dX, nx * 2); hipMalloc(&dW, (size_t)S * s.N * s.K * 4);
        hipMemcpy(dY, hy.data(), ny * 2, hipMemcpyHostToDevice); hipMemcpy(dX, hx.data(), nx * 2, hipMemcpyHostToDevice);
        const dim3 grid((unsigned)(tiles * ((S + 7) / 8 * 8)));
        auto run = [&](int mode, const char* name) {
            auto kern = mode == 0 ? wgrad_proto<0, 32> : (mode == 1 ? wgrad_proto<1, 32> : wgrad_proto<1, 64>);
            hipMemset(dW, 0, (size_t)S * s.N * s.K * 4);
            float best = 1e30f;
            for (int r = 0; r < 5; ++r) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, dY, dX, dW, s.M, s.N, s.K, S);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r > 0 && ms < best) best = ms;
            }
            std::vector<float> hw((size_t)S * s.N * s.K);
            hipMemcpy(hw.data(), dW, hw.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int q = 0; q < 12; ++q) {
                const size_t n = ((size_t)q * 2654435761u) % s.N, k = ((size_t)q * 40503u + 17) % s.K;
                double ref = 0, got = 0;
                for (size_t m = 0; m < (size_t)s.M; ++m) ref += (double)bf2f(hy[m * s.N + n]) * bf2f(hx[m * s.K + k]);
                for (int c = 0; c < S; ++c) got += hw[((size_t)c * s.N + n) * s.K + k];
                const double err = fabs(ref - got) / (fabs(ref) + 1.0);
                if (err > worst) worst = err;
            }
            printf("%-10s %7dx%5dx%5d S=%3d  %-30s %8.1f us  %7.1f TFLOP/s  rel err %.1e%s\n", s.what, s.M, s.N, s.K, S, name, best * 1e3, 2.0 * s.M * s.N * s.K / best / 1e9, worst,
                   worst < 1e-3 ? "" : "  <-- WRONG");
        };
        run(0, "registers -> ds_write_b128");
        run(1, "LDS-DMA, XOR-placed pieces");
        run(2, "LDS-DMA, 64-token steps");
        hipFree(dY); hipFree(dX); hipFree(dW);
    }
    return 0;
}
