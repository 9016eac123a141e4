// Backward building blocks, first members of the family.  The reference has no backward
// code (torch autograd over model.py); the closed forms below are the ones the test suite pins against that autograd.
//   uf_gelu_bwd            dx = dy * GELU'(a)                                   (nn.GELU, model.py:657-660)
//   uf_layernorm_bwd       dx, dgamma, dbeta of nn.LayerNorm over the last dim   (model.py:881,888,952,987)
//   uf_dwconv3x3_wgrad     tap and bias gradients of the depthwise 3x3           (LeFF dwconv, model.py:659)
// (the INPUT gradient of the depthwise conv is the forward stencil with flipped taps: uf_dwconv3x3_fwd, gelu = 0).
// All reductions over tokens are two-stage (per-thread partials in a workspace, then a finalize kernel that adds them
// in a fixed order): bit-reproducible, no atomics.
#include "uf_internal.h"

namespace uf {
namespace {

// ---------------------------------------------------------------------------------------------------------------
// GELU'(a) = Phi(a) + a phi(a)   (nn.GELU default = erf form, model.py:657-660)
// ---------------------------------------------------------------------------------------------------------------
// (uf_common.h gelu_grad_t: the erf form for f32 operands, the derivative of the bf16 forward's own GELU for bf16 operands)

template <typename T> struct Vec {   // primary: the 2-byte operand types (bf16, f16)
    static_assert(sizeof(T) == 2, "2-byte operand type");
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const T* p, float* f) { unpack8<T>(*reinterpret_cast<const u32x4*>(p), f); }
    static __device__ __forceinline__ void store(T* p, const float* f) { *reinterpret_cast<u32x4*>(p) = pack8<T>(f); }
};
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float* f) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p);
        f[0] = r[0]; f[1] = r[1]; f[2] = r[2]; f[3] = r[3];
    }
    static __device__ __forceinline__ void store(float* p, const float* f) { *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]}; }
};

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ a, const T* __restrict__ dy, T* __restrict__ dx, long long nvec) {
    constexpr int N = Vec<T>::N;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    float fa[N], fd[N];
    Vec<T>::load(a + i * N, fa);
    Vec<T>::load(dy + i * N, fd);
#pragma unroll
    for (int k = 0; k < N; ++k) fd[k] *= gelu_grad_t<T>(fa[k]);
    Vec<T>::store(dx + i * N, fd);
}

// y = GELU(a) as a separate pass (training forward: the pre-activation a is kept for GELU', so the activation cannot stay
// fused in the GEMM epilogue without writing both).  Same flavour as the fused epilogues: erf form for f32 operands, the
// packed sigmoid form for results stored as bf16 (uf_common.h gelu_n).
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ a, T* __restrict__ y, long long nvec) {
    constexpr int N = Vec<T>::N;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    float f[N];
    Vec<T>::load(a + i * N, f);
    gelu_n<T, N>(f);
    Vec<T>::store(y + i * N, f);
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  xhat = (x-mu)*rstd, g = dy*gamma:
//   dx = rstd * (g - mean(g) - xhat * mean(g*xhat));  dgamma = sum_rows dy*xhat;  dbeta = sum_rows dy.
// Row layout as the forward kernel: LPR lanes share a row (DPP all-reduce), RPB rows per workgroup pass; a thread keeps
// the same channels over all its rows, so dgamma/dbeta accumulate in registers and land in partial[slot][2][C].
// ---------------------------------------------------------------------------------------------------------------
// dy: f32 or the operand type (TD); rows of dy are indexed by m, rows of x / add / dx by tok(m) = m, or -- win_h > 0 -- the token of
// window-order row m (dy still in the order the attention half produced it: window_reverse + roll back folded in); add (optional):
// a second gradient of the same tensor summed into dx (the residual path).
template <typename TD> __device__ __forceinline__ f32x4 load_dy4(const TD* p) {   // primary: the 2-byte operand types
    const u32x2 r = *reinterpret_cast<const u32x2*>(p);
    float a, b, c, d;
    unpack2<TD>(r[0], a, b);
    unpack2<TD>(r[1], c, d);
    return f32x4{a, b, c, d};
}
template <> __device__ __forceinline__ f32x4 load_dy4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// optional second output of the LayerNorm backward: mode 0 none, 1 = row tok, 2 = the window-order row of tok (H, W, shift of the partition);
// scale: per-image factors (hw rows per image) or NULL
template <typename TD> struct LnCast { TD* out; const float* scale; int mode, hw, H, W, shift; };
__device__ __forceinline__ int token_to_window_row(int tok, int H, int W, int shift) {      // inverse of window_row_to_token
    const int hw = H * W, b = tok / hw, r = tok - b * hw;
    const int h = r / W, w = r - h * W;
    int hs = h - shift; if (hs < 0) hs += H;
    int ws = w - shift; if (ws < 0) ws += W;
    return (((b * (H >> 3) + (hs >> 3)) * (W >> 3) + (ws >> 3)) << 6) + ((hs & 7) << 3) + (ws & 7);
}
template <int C, typename TD>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ gamma,
                                                            const TD* __restrict__ dy, int ld_dy, const float* __restrict__ add, float* __restrict__ dx, int ld_dx,
                                                            float* __restrict__ partial, int rows, int win_h, int win_w, int shift, LnCast<TD> cast) {
    constexpr int LPR = (C / 4) < 64 ? (C / 4) : 64;
    constexpr int V4 = C / (4 * LPR);
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    const int nblk = (rows + RPB - 1) / RPB;
    f32x4 gm[V4], adg[V4], adb[V4];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        gm[i] = *reinterpret_cast<const f32x4*>(gamma + (i * LPR + sub) * 4);
        adg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        adb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int mb = blockIdx.x; mb < nblk; mb += gridDim.x) {      // workgroup-uniform trip count
        const int m = mb * RPB + rl;
        const bool live = m < rows;
        const int mc = live ? m : rows - 1;                        // clamped: loads stay unconditional
        const int tok = win_h > 0 ? window_row_to_token(mc, win_h, win_w, shift) : mc;
        f32x4 v[V4], d[V4], ad[V4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)tok * ld_x + (i * LPR + sub) * 4);
            d[i] = load_dy4<TD>(dy + (size_t)mc * ld_dy + (i * LPR + sub) * 4);
            // the residual-path gradient is requested with the row, not where it is added (a dependent round trip per row block)
            ad[i] = add ? *reinterpret_cast<const f32x4*>(add + (size_t)tok * ld_dx + (i * LPR + sub) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
        const float mean = allreduce<RedSum, LPR>(sum) * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            v[i] -= mean;
            sq += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
        }
        const float rstd = 1.0f / sqrtf(allreduce<RedSum, LPR>(sq) * (1.0f / C) + 1e-5f);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            v[i] *= rstd;                                          // xhat
            const f32x4 g = d[i] * gm[i];
            s1 += (g[0] + g[1]) + (g[2] + g[3]);
            const f32x4 gx = g * v[i];
            s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
        }
        s1 = allreduce<RedSum, LPR>(s1) * (1.0f / C);
        s2 = allreduce<RedSum, LPR>(s2) * (1.0f / C);
        if (live) {
            // the copy of dx the NEXT GEMM of the backward reads (grad_fork's job, one pass over dx less): operand type, times the per-image
            // DropPath scale of the branch it enters, at the token's own row or at its window-order row
            const size_t crow = cast.mode == 2 ? (size_t)token_to_window_row(tok, cast.H, cast.W, cast.shift) : (size_t)tok;
            const float cs = (cast.mode && cast.scale) ? cast.scale[mc / cast.hw] : 1.0f;
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                f32x4 r = (d[i] * gm[i] - s1 - v[i] * s2) * rstd;
                if (add) r = r + ad[i];
                *reinterpret_cast<f32x4*>(dx + (size_t)tok * ld_dx + (i * LPR + sub) * 4) = r;
                if (cast.mode) store4(cast.out + crow * C + (i * LPR + sub) * 4, r * cs);
                adg[i] += d[i] * v[i];
                adb[i] += d[i];
            }
        }
    }
    // the RPB row lanes of the workgroup meet in LDS and are added in lane order: ONE partial row per workgroup (round 6; there were RPB of them, which is
    // what kept the grid at 512 workgroups = 2 waves per SIMD of a kernel that is a pure stream: with 2048 it holds 8 and four times the bytes in flight)
    __shared__ float red[RPB][2 * C];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        *reinterpret_cast<f32x4*>(&red[rl][(i * LPR + sub) * 4]) = adg[i];
        *reinterpret_cast<f32x4*>(&red[rl][C + (i * LPR + sub) * 4]) = adb[i];
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * 2 * C;
    for (int j = threadIdx.x; j < 2 * C; j += 256) {
        float t = red[0][j];
#pragma unroll
        for (int r = 1; r < RPB; ++r) t += red[r][j];
        out[j] = t;
    }
}

// out[j] = sum_{p < P} partial[p * stride + j], j < n.  A workgroup owns 32 columns; 8 "p-lanes" per column each add every 8th
// partial in index order (loads of a wave: 32 consecutive columns = 128 B per p), then the 8 lane sums are added in lane order
// through LDS.  The order depends only on P: bit-reproducible.  (The first version gave every column ONE thread that walked all P
// partials -- up to 32 K dependent steps on a handful of threads; it was 65 % of the GPU time of a training step.)
__device__ __forceinline__ void column_sum_block(const float* __restrict__ partial, int P, size_t stride, float* __restrict__ out, int n, int blk) {
    __shared__ float sh[8][33];
    const int col = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int j = blk * 32 + col;
    float s = 0.f;
    if (j < n) {
        const float* src = partial + j;
        int p = pl;
        for (; p + 56 < P; p += 64) {       // eight independent loads in flight per thread: the kernel is a chain of L2 round trips (P / 8 partials
            float v[8];                     // per thread), 560 launches per training step, 8 ms of it with four in flight
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(p + 8 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; p + 24 < P; p += 32) {
            const float a = src[(size_t)p * stride], b = src[(size_t)(p + 8) * stride], c = src[(size_t)(p + 16) * stride], d = src[(size_t)(p + 24) * stride];
            s = (((s + a) + b) + c) + d;
        }
        for (; p < P; p += 8) s += src[(size_t)p * stride];
    }
    sh[pl][col] = s;
    __syncthreads();
    if (pl == 0 && j < n) {
        float t = sh[0][col];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sh[k][col];
        out[j] = t;
    }
}
__global__ __launch_bounds__(256) void column_sum_kernel(const float* __restrict__ partial, int P, size_t stride, float* __restrict__ out, int n) {
    column_sum_block(partial, P, stride, out, n, blockIdx.x);
}
// two independent sums in one launch (weight + bias gradient of a linear layer, dgamma + dbeta of a LayerNorm): these kernels are
// launch-latency bound (~10 us each, 560 per training step), the arithmetic per output is the same as in the single form
// The same sum for MANY columns and few partials (the weight gradient of a wide linear layer: 1 M columns, 8 partials): a thread owns four
// consecutive columns (16-byte loads, a wave reads 1 KiB per partial), PL p-lanes per column group add every PL-th partial in index order and
// are then added in lane order.  With one column per thread these launches read 4 bytes per lane and were 4.5 ms of a 76 ms training step.
template <int PL>
__device__ __forceinline__ void column_sum_wide_block(const float* __restrict__ partial, int P, size_t stride, float* __restrict__ out, int n, int blk) {
    constexpr int G = 256 / PL;
    __shared__ f32x4 shw[PL][G];
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    const int j = (blk * G + g) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (j < n) {
        const float* src = partial + j;
        int p = pl;
        for (; p + 7 * PL < P; p += 8 * PL) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(p + PL * u) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; p < P; p += PL) s += *reinterpret_cast<const f32x4*>(src + (size_t)p * stride);
    }
    if (PL > 1) {
        shw[pl][g] = s;
        __syncthreads();
        if (pl == 0 && j < n) {
            f32x4 t = shw[0][g];
#pragma unroll
            for (int k = 1; k < PL; ++k) t += shw[k][g];
            *reinterpret_cast<f32x4*>(out + j) = t;
        }
    } else if (j < n) {
        *reinterpret_cast<f32x4*>(out + j) = s;
    }
}
// wide1: 0 = one column per thread (column_sum_block), 1 / 4 = four columns per thread with that many p-lanes
__global__ __launch_bounds__(256) void column_sum2_kernel(const float* __restrict__ p1, int P1, size_t s1, float* __restrict__ o1, int n1, int nb1,
                                                          const float* __restrict__ p2, int P2, size_t s2, float* __restrict__ o2, int n2, int wide1) {
    if ((int)blockIdx.x < nb1) {                                                      // workgroup-uniform branches (the blocks hold barriers)
        if (wide1 == 1) column_sum_wide_block<1>(p1, P1, s1, o1, n1, blockIdx.x);
        else if (wide1 == 4) column_sum_wide_block<4>(p1, P1, s1, o1, n1, blockIdx.x);
        else if (wide1 == 16) column_sum_wide_block<16>(p1, P1, s1, o1, n1, blockIdx.x);
        else column_sum_block(p1, P1, s1, o1, n1, blockIdx.x);
    } else column_sum_block(p2, P2, s2, o2, n2, blockIdx.x - nb1);
}
static inline void launch_column_sum2(hipStream_t st, const float* p1, int P1, size_t s1, float* o1, int n1, const float* p2, int P2, size_t s2, float* o2, int n2) {
    const bool vec = n1 % 4 == 0 && s1 % 4 == 0 && ((uintptr_t)p1 % 16) == 0 && ((uintptr_t)o1 % 16) == 0;
    // p-lanes so that about 256 K threads share the work (a thread of the one-lane form walks all P partials: 4-8 dependent round trips
    // of 8 loads on 256 workgroups); the summation order depends only on (n, P)
    int wide1 = 0;
    if (vec) wide1 = n1 >= (1 << 20) ? 1 : (n1 >= (1 << 18) ? (P1 >= 8 ? 4 : 1) : (n1 >= (1 << 14) && P1 >= 32 ? 16 : (n1 >= (1 << 15) && P1 >= 8 ? 4 : 0)));
    const int nb1 = wide1 ? (n1 / 4 + 256 / wide1 - 1) / (256 / wide1) : (n1 + 31) / 32, nb2 = (n2 + 31) / 32;
    hipLaunchKernelGGL(column_sum2_kernel, dim3(nb1 + nb2), dim3(256), 0, st, p1, P1, s1, o1, n1, nb1, p2, P2, s2, o2, n2, wide1);
}
constexpr int COLSUM_COLS = 32;   // columns per workgroup of column_sum_kernel

constexpr int LN_BWD_MAX_BLOCKS = 2048;   // 8 workgroups per CU (62 registers: 8 waves per SIMD)

// ---------------------------------------------------------------------------------------------------------------
// depthwise 3x3 weight gradient: dw[t][c] = sum_{b,y,x} dc[b,y,x,c] * h[b, y+ky-1, x+kx-1, c],  db[c] = sum dc.
// Thread = N channels x a 4-row column strip (as the forward stencil), grid-strided over strips with a stride that is
// a multiple of the channel-group count, so a thread's channels never change: 10 x N register accumulators per thread,
// written to partial[thread][10][N]; wgrad_finalize adds the threads that share a channel group.
// ---------------------------------------------------------------------------------------------------------------
constexpr int DWB_R = 4;
// workgroups of dwconv3x3_wgrad_kernel: 176 VGPRs = 2 workgroups per CU resident -> 512 fills the chip once (256 left every SIMD with
// ONE wave of a memory-bound kernel)
static int dw_wgrad_blocks() { return 512; }

template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const T* __restrict__ h, const T* __restrict__ dc, float* __restrict__ partial,
                                                              int B, int H, int W, int C) {
    constexpr int N = Vec<T>::N;
    const int cv = C / N, strips = H / DWB_R;
    const long long total = (long long)B * strips * W * cv;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;       // multiple of cv (checked by the launcher)
    const int c = (int)(gtid % cv) * N;
    float acc[10][N];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int i = 0; i < N; ++i) acc[t][i] = 0.f;
    for (long long idx = gtid; idx < total; idx += stride) {
        long long rest = idx / cv;
        const int xw = (int)(rest % W); rest /= W;
        const int y0 = (int)(rest % strips) * DWB_R;
        const int b = (int)(rest / strips);
        const size_t img = (size_t)b * H * W * C + c;
        float d[DWB_R][N];
#pragma unroll
        for (int r = 0; r < DWB_R; ++r) {
            Vec<T>::load(dc + img + ((size_t)(y0 + r) * W + xw) * C, d[r]);
#pragma unroll
            for (int i = 0; i < N; ++i) acc[9][i] += d[r][i];
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ixr = xw + kx - 1;
            const float mx = (ixr >= 0 && ixr < W) ? 1.0f : 0.0f;
            const int ix = ixr < 0 ? 0 : (ixr >= W ? W - 1 : ixr);
#pragma unroll
            for (int r = -1; r <= DWB_R; ++r) {            // input row y0 + r pairs with output rows r+1-ky
                const int iyr = y0 + r;
                const float m = (iyr >= 0 && iyr < H) ? mx : 0.0f;
                const int iy = iyr < 0 ? 0 : (iyr >= H ? H - 1 : iyr);
                float f[N];
                Vec<T>::load(h + img + ((size_t)iy * W + ix) * C, f);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int orow = r + 1 - ky;
                    if (orow < 0 || orow >= DWB_R) continue;
#pragma unroll
                    for (int i = 0; i < N; ++i) acc[ky * 3 + kx][i] = fmaf(f[i] * m, d[orow][i], acc[ky * 3 + kx][i]);
                }
            }
        }
    }
    float* out = partial + (size_t)gtid * 10 * N;
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int i = 0; i < N; ++i) out[t * N + i] = acc[t][i];
}

// dw9[t][c] (t < 9) and dbias[c] (t == 9) = sum over the threads j = cg, cg + cv, ... of partial[j][t][c % N].  32 outputs per
// workgroup, 8 p-lanes per output (every 8th contributing thread, in order), lane sums added in lane order: fixed order.
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_finalize(const float* __restrict__ partial, long long nthreads, int cv, int N, float* __restrict__ dw9,
                                                               float* __restrict__ dbias, int C) {
    __shared__ float sh[8][33];
    const int col = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + col;   // over 10 * C outputs
    float s = 0.f;
    int t = 0, c = 0;
    if (j < 10 * C) {
        t = j / C; c = j % C;
        const int cg = c / N, i = c % N;
        const float* src = partial + t * N + i;
        const long long step = 8LL * cv;
        long long th = cg + (long long)pl * cv;
        for (; th + 7 * step < nthreads; th += 8 * step) {      // eight loads in flight (it was one dependent round trip per partial: 58 us per call)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(th + u * step) * 10 * N];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; th < nthreads; th += step) s += src[(size_t)th * 10 * N];
    }
    sh[pl][col] = s;
    __syncthreads();
    if (pl == 0 && j < 10 * C) {
        float r = sh[0][col];
#pragma unroll
        for (int k = 1; k < 8; ++k) r += sh[k][col];
        if (t < 9) dw9[(size_t)t * C + c] = r; else dbias[c] = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Depthwise 3x3 backward in ONE pass over dc (the gradient at the stencil's output):
//   da1 = T(stencil(dc; flipped taps)) * GELU'(a1)       -- what uf_dwconv3x3_mul_dgelu writes, same FMA order (2-byte types: bit-identical)
//   dw[a][b][c], db[c]                                      -- what uf_dwconv3x3_wgrad computes from h1 and dc
// The tap gradient is taken centred on the thread's own pixels:
//   dw[a][b] = sum_{y,x} dc[y][x] h1[y+a-1][x+b-1] = sum_{y',x'} h1[y'][x'] dc[y'-a+1][x'-b+1],
// so the dc neighbours are the ones the input-gradient stencil loads anyway, and h1 = T(GELU(a1 as stored)) -- exactly what the
// forward wrote -- is recomputed from the pre-activation the thread holds for GELU'.  h1 and dc are not read a second time (the
// separate tap-gradient kernel moved 2 x M x 4C operands per block again and wrote 42 MB of per-thread partial sums per call).
// A workgroup = `cg` channel groups (N channels each) x 256/cg pixel columns, strips of R rows; it walks the (image, strip, x)
// positions with a fixed stride and keeps its channels, so the 10 x N accumulators stay in registers; at the end the pixel threads
// of a channel group meet in LDS (fixed order) and ONE partial per workgroup and channel goes to the workspace;
// dwconv3x3_bwd_finalize adds the workgroups in order.  Bit-reproducible, no atomics.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int N> __device__ __forceinline__ void round_t(float* f) {   // to the operand type and back (what a later pass would read)
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < N; i += 2) unpack2<T>(pack2<T>(f[i], f[i + 1]), f[i], f[i + 1]);
    }
}

// byte offsets are 32-bit (the launcher checks the tensor is under 4 GiB): one VGPR per address next to the uniform base pointer
template <typename T, int N, int R>
__global__ __launch_bounds__(256, 2) void dwconv3x3_bwd_kernel(const T* __restrict__ dc, const float* __restrict__ w9f, const T* __restrict__ a1, T* __restrict__ da1,
                                                               float* __restrict__ partial, int B, int H, int W, int C, int cg_log2) {
    using CH = Chunk<T, N>;
    using Raw = typename CH::Raw;
    __shared__ float red[256][N + 1];
    const int tid = threadIdx.x;
    const int cg = 1 << cg_log2, cv = C / N, cb = cv >> cg_log2, px = 256 >> cg_log2;
    const int cgi = tid & (cg - 1), pi = tid >> cg_log2;
    const int c = (((int)blockIdx.x % cb) * cg + cgi) * N;
    const int strips = H / R;
    const int P = B * strips * W;
    const int pstep = ((int)gridDim.x / cb) * px;
    const unsigned rowb = (unsigned)W * C * (unsigned)sizeof(T), pixb = (unsigned)C * (unsigned)sizeof(T);
    const char* dcb = reinterpret_cast<const char*>(dc);
    const char* a1b = reinterpret_cast<const char*>(a1);
    char* dab = reinterpret_cast<char*>(da1);
    float wacc[10][N];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int i = 0; i < N; ++i) wacc[t][i] = 0.f;
    for (int pos = ((int)blockIdx.x / cb) * px + pi; pos < P; pos += pstep) {
        const int xw = pos % W, rest = pos / W;
        const int y0 = (rest % strips) * R, b = rest / strips;
        const unsigned o00 = ((unsigned)(b * H + y0) * W + xw) * pixb + (unsigned)c * (unsigned)sizeof(T);   // this thread's first pixel
        Raw araw[R];
#pragma unroll
        for (int r = 0; r < R; ++r) araw[r] = *reinterpret_cast<const Raw*>(a1b + (o00 + r * rowb));
        float hc[R][N], dg[R][N], acc[R][N];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float av[N];
            CH::unpack(araw[r], av);
#pragma unroll
            for (int i = 0; i < N; ++i) gelu_and_grad_t<T>(av[i], hc[r][i], dg[r][i]);      // one exp2 / rcp pair for both (uf_common.h)
            round_t<T, N>(hc[r]);
#pragma unroll
            for (int i = 0; i < N; ++i) acc[r][i] = 0.f;
        }
        // rows y0 - 1 and y0 + R clamped into the image (their values are masked below)
        const unsigned otop = y0 > 0 ? o00 - rowb : o00, obot = y0 + R < H ? o00 + R * rowb : o00 + (R - 1) * rowb;
        const float mtop = y0 > 0 ? 1.0f : 0.0f, mbot = y0 + R < H ? 1.0f : 0.0f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ixr = xw + kx - 1;
            const float mx = (ixr >= 0 && ixr < W) ? 1.0f : 0.0f;
            const int dxp = ixr < 0 ? 0 : (ixr >= W ? 0 : kx - 1);     // clamped column, relative to xw
            const unsigned oshift = (unsigned)(dxp * (int)pixb);
            // the taps are the same in every position step: re-read them (L1) through an offset the compiler cannot see through, or it
            // keeps all 9 N of them live across the loop next to the 10 N accumulators and spills
            int wo = kx * C + c;
            asm volatile("" : "+v"(wo));
            float wk[3][N];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w9f + (size_t)(ky * 3) * C + wo);
                static_assert(N == 4 || N == 8, "4 or 8 channels per thread");
                wk[ky][0] = wv[0] * mx; wk[ky][1] = wv[1] * mx; wk[ky][2] = wv[2] * mx; wk[ky][3] = wv[3] * mx;
                if constexpr (N == 8) {
                    const f32x4 wu = *reinterpret_cast<const f32x4*>(w9f + (size_t)(ky * 3) * C + wo + 4);
                    wk[ky][4] = wu[0] * mx; wk[ky][5] = wu[1] * mx; wk[ky][6] = wu[2] * mx; wk[ky][7] = wu[3] * mx;
                }
            }
            Raw fr[R + 2];
#pragma unroll
            for (int r = -1; r <= R; ++r) {
                const unsigned o = (r == -1 ? otop : (r == R ? obot : o00 + r * rowb)) + oshift;
                fr[r + 1] = *reinterpret_cast<const Raw*>(dcb + o);
            }
#pragma unroll
            for (int r = -1; r <= R; ++r) {                 // dc row y0 + r feeds output rows r + 1 - ky
                float f[N];
                CH::unpack(fr[r + 1], f);
                if (r == -1) {
#pragma unroll
                    for (int i = 0; i < N; ++i) f[i] *= mtop;
                }
                if (r == R) {
#pragma unroll
                    for (int i = 0; i < N; ++i) f[i] *= mbot;
                }
                if (kx == 1 && r >= 0 && r < R) {
#pragma unroll
                    for (int i = 0; i < N; ++i) wacc[9][i] += f[i];
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int orow = r + 1 - ky;
                    if (orow < 0 || orow >= R) continue;
#pragma unroll
                    for (int i = 0; i < N; ++i) acc[orow][i] = fmaf(f[i], wk[ky][i], acc[orow][i]);
                }
                if (kx != 1) {                               // the column mask of the tap gradient (the stencil has it inside wk)
#pragma unroll
                    for (int i = 0; i < N; ++i) f[i] *= mx;
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int orow = r + 1 - ky;
                    if (orow < 0 || orow >= R) continue;
#pragma unroll
                    for (int i = 0; i < N; ++i) wacc[(2 - ky) * 3 + (2 - kx)][i] = fmaf(hc[orow][i], f[i], wacc[(2 - ky) * 3 + (2 - kx)][i]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);              // one column tap at a time: all 3 (R + 2) loads hoisted = the register file
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            round_t<T, N>(acc[r]);
#pragma unroll
            for (int i = 0; i < N; ++i) acc[r][i] *= dg[r][i];
            *reinterpret_cast<Raw*>(dab + (o00 + r * rowb)) = CH::pack(acc[r]);
        }
    }
    // the pixel threads of a channel group meet in LDS: thread (cgi, i) adds its px partners in pixel order
    float* out = partial + (size_t)blockIdx.x * 10 * cg * N;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[tid][i] = wacc[t][i];
        __syncthreads();
        if (tid < cg * N) {
            const int g = tid / N, i = tid % N;
            float s = 0.f;
            for (int q = 0; q < px; ++q) s += red[q * cg + g][i];
            out[t * cg * N + tid] = s;
        }
        __syncthreads();
    }
}

// The same pass WALKING along x (W a multiple of 8): a thread owns 4 channels x a strip of R rows x SEG consecutive pixel columns per
// position step and keeps the last three dc columns ((R + 2) rows each) in registers as f32 -- one new column per step instead of
// three -- exactly as dwconv3x3_walk_kernel does for the forward (uf_elementwise.hip: the 3x re-read of the input through L1 / L2 was
// what held the one-column-per-thread form at 2.8 TB/s).  Positions are (image, strip, segment); same partial-sum layout.
template <typename T, int SEG>
__global__ __launch_bounds__(256, 2) void dwconv3x3_bwd_walk_kernel(const T* __restrict__ dc, const float* __restrict__ w9f, const T* __restrict__ a1,
                                                                    T* __restrict__ da1, float* __restrict__ partial, int B, int H, int W, int C, int cg_log2) {
    constexpr int N = 4, R = DWB_R;
    using CH = Chunk<T, N>;
    using Raw = typename CH::Raw;
    __shared__ float red[256][N + 1];
    const int tid = threadIdx.x;
    const int cg = 1 << cg_log2, cv = C / N, cb = cv >> cg_log2, px = 256 >> cg_log2;
    const int cgi = tid & (cg - 1), pi = tid >> cg_log2;
    const int c = (((int)blockIdx.x % cb) * cg + cgi) * N;
    const int strips = H / R, segs = W / SEG;
    const int P = B * strips * segs;
    const int pstep = ((int)gridDim.x / cb) * px;
    const unsigned pixb = (unsigned)C * (unsigned)sizeof(T), rowb = (unsigned)W * pixb;
    const char* dcb = reinterpret_cast<const char*>(dc);
    const char* a1b = reinterpret_cast<const char*>(a1);
    char* dab = reinterpret_cast<char*>(da1);
    f32x4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const f32x4*>(w9f + (size_t)t * C + c);
    float wacc[10][N];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int i = 0; i < N; ++i) wacc[t][i] = 0.f;
    for (int pos = ((int)blockIdx.x / cb) * px + pi; pos < P; pos += pstep) {
        const int x0 = (pos % segs) * SEG, rest = pos / segs;
        const int y0 = (rest % strips) * R, b = rest / strips;
        const unsigned o00 = ((unsigned)(b * H + y0) * W + x0) * pixb + (unsigned)c * (unsigned)sizeof(T);
        unsigned ro[R + 2];
        ro[0] = y0 > 0 ? o00 - rowb : o00;
#pragma unroll
        for (int r = 0; r < R; ++r) ro[r + 1] = o00 + r * rowb;
        ro[R + 1] = y0 + R < H ? o00 + R * rowb : o00 + (R - 1) * rowb;
        const float mtop = y0 > 0 ? 1.0f : 0.0f, mbot = y0 + R < H ? 1.0f : 0.0f;
        float col[3][R + 2][N];        // dc columns xx - 1, xx, xx + 1 of the current pixel column xx, rotating
        auto load_col = [&](float (&cl)[R + 2][N], int dx, float m) {
            Raw raw[R + 2];
#pragma unroll
            for (int r = 0; r < R + 2; ++r) raw[r] = *reinterpret_cast<const Raw*>(dcb + (ro[r] + (unsigned)(dx * (int)pixb)));
#pragma unroll
            for (int r = 0; r < R + 2; ++r) {
                CH::unpack(raw[r], cl[r]);
                const float mr = (r == 0 ? mtop : (r == R + 1 ? mbot : 1.0f)) * m;
                if (r == 0 || r == R + 1) {
#pragma unroll
                    for (int i = 0; i < N; ++i) cl[r][i] *= mr;
                } else if (m != 1.0f) {
#pragma unroll
                    for (int i = 0; i < N; ++i) cl[r][i] *= m;
                }
            }
        };
        auto emit = [&](const float (&cL)[R + 2][N], const float (&cM)[R + 2][N], const float (&cR)[R + 2][N], int dx) {
            Raw araw[R];
#pragma unroll
            for (int r = 0; r < R; ++r) araw[r] = *reinterpret_cast<const Raw*>(a1b + (ro[r + 1] + (unsigned)(dx * (int)pixb)));
            float hc[R][N], dg[R][N], acc[R][N];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float av[N];
                CH::unpack(araw[r], av);
#pragma unroll
                for (int i = 0; i < N; ++i) gelu_and_grad_t<T>(av[i], hc[r][i], dg[r][i]);  // one exp2 / rcp pair for both (uf_common.h)
                round_t<T, N>(hc[r]);
#pragma unroll
                for (int i = 0; i < N; ++i) { acc[r][i] = 0.f; wacc[9][i] += cM[r + 1][i]; }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float (&cl)[R + 2][N] = kx == 0 ? cL : (kx == 1 ? cM : cR);
#pragma unroll
                for (int r = -1; r <= R; ++r)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int orow = r + 1 - ky;
                        if (orow < 0 || orow >= R) continue;
#pragma unroll
                        for (int i = 0; i < N; ++i) {
                            acc[orow][i] = fmaf(cl[r + 1][i], wt[ky * 3 + kx][i], acc[orow][i]);
                            wacc[(2 - ky) * 3 + (2 - kx)][i] = fmaf(hc[orow][i], cl[r + 1][i], wacc[(2 - ky) * 3 + (2 - kx)][i]);
                        }
                    }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                round_t<T, N>(acc[r]);
#pragma unroll
                for (int i = 0; i < N; ++i) acc[r][i] *= dg[r][i];
                *reinterpret_cast<Raw*>(dab + (ro[r + 1] + (unsigned)(dx * (int)pixb))) = CH::pack(acc[r]);
            }
        };
        load_col(col[0], x0 > 0 ? -1 : 0, x0 > 0 ? 1.0f : 0.0f);
        load_col(col[1], 0, 1.0f);
        const bool last_seg = x0 + SEG >= W;
#pragma unroll 1
        for (int xs = 0; xs < SEG; xs += 3) {          // three steps per turn: the column registers rotate by name
            {
                const bool edge = last_seg && xs + 1 >= SEG;
                load_col(col[2], edge ? xs : xs + 1, edge ? 0.0f : 1.0f);
                emit(col[0], col[1], col[2], xs);
            }
            if (xs + 1 < SEG) {
                const bool edge = last_seg && xs + 2 >= SEG;
                load_col(col[0], edge ? xs + 1 : xs + 2, edge ? 0.0f : 1.0f);
                emit(col[1], col[2], col[0], xs + 1);
            }
            if (xs + 2 < SEG) {
                const bool edge = last_seg && xs + 3 >= SEG;
                load_col(col[1], edge ? xs + 2 : xs + 3, edge ? 0.0f : 1.0f);
                emit(col[2], col[0], col[1], xs + 2);
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * 10 * cg * N;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[tid][i] = wacc[t][i];
        __syncthreads();
        if (tid < cg * N) {
            const int g = tid / N, i = tid % N;
            float s = 0.f;
            for (int q = 0; q < px; ++q) s += red[q * cg + g][i];
            out[t * cg * N + tid] = s;
        }
        __syncthreads();
    }
}

// dw9[t][c] (t < 9) and dbias[c] (t == 9) = sum over the workgroups w = cblk, cblk + cb, ... of partial[w][t][c within the block].
// 32 outputs per workgroup, 8 p-lanes per output (every 8th workgroup, in order), lane sums added in lane order: fixed order.
__global__ __launch_bounds__(256) void dwconv3x3_bwd_finalize(const float* __restrict__ partial, int blocks, int cb, int cgN, float* __restrict__ dw9,
                                                             float* __restrict__ dbias, int C, int accumulate) {
    __shared__ float sh[8][33];
    const int col = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + col;   // over 10 * C outputs
    float s = 0.f;
    int t = 0, c = 0;
    if (j < 10 * C) {
        t = j / C; c = j % C;
        const int cblk = c / cgN, cl = c % cgN;
        const float* src = partial + (size_t)t * cgN + cl;
        const size_t wstride = (size_t)10 * cgN;
        const int slots = blocks / cb;
        int q = pl;
        for (; q + 56 < slots; q += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)((q + 8 * u) * cb + cblk) * wstride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; q < slots; q += 8) s += src[(size_t)(q * cb + cblk) * wstride];
    }
    sh[pl][col] = s;
    __syncthreads();
    if (pl == 0 && j < 10 * C) {
        float r = sh[0][col];
#pragma unroll
        for (int k = 1; k < 8; ++k) r += sh[k][col];
        float* dst = t < 9 ? dw9 + (size_t)t * C + c : dbias + c;
        *dst = accumulate ? *dst + r : r;           // accumulate: a later image chunk of a tensor of 4 GiB or more (fixed chunk order)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// nn.Linear weight / bias gradients:  dW[n][k] = sum_m dY[m][n] X[m][k],  db[n] = sum_m dY[m][n]   (linear_bwd).
// The contraction runs over TOKENS, so both MFMA operands need 8 consecutive tokens of ONE column per lane: a step
// stages dY[32 tokens][64 n] and X[32 tokens][64 k] transposed into LDS ([column][token]) and each wave issues 2 x 2
// MFMAs on its 32 x 32 quadrant of the 64 x 64 output tile.  blockIdx.y cuts the tokens into gridDim.y chunks; partial
// tiles go to ws[chunk][N][K] (db: ws_b[chunk][N]) and column_sum_kernel adds the chunks in order (bit-reproducible).
// First version: LDS-write bound (2-byte transposing stores); the transposing LDS reads of gfx950 are the next step.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const T* __restrict__ dY, int ldy, const T* __restrict__ X, int ldx,
                                                           float* __restrict__ ws_w, float* __restrict__ ws_b, int M, int N, int K) {
    constexpr int SZ = sizeof(T), EP = 16 / SZ;            // elements per 16-byte piece
    constexpr int PPR = 64 / EP;                            // pieces per 64-column row
    constexpr int PPT = 32 * PPR / 256;                     // pieces per thread and operand per step (1 bf16, 2 f32)
    constexpr int STR = 32 * SZ + 16;                       // LDS row stride of a [column][32 tokens] tile
    __shared__ __attribute__((aligned(16))) char Yt[64 * STR];
    __shared__ __attribute__((aligned(16))) char Xt[64 * STR];
    __shared__ float Bs[32][64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int k_tiles = (K + 63) / 64;
    const int n0 = (blockIdx.x / k_tiles) * 64, k0 = (blockIdx.x % k_tiles) * 64;
    const int wn = wave >> 1, wk = wave & 1;                // 32 x 32 quadrant of this wave
    const bool do_bias = (blockIdx.x % k_tiles) == 0;       // one k-tile column of blocks also sums dY
    // token range of this chunk, in steps of 32
    const int steps_all = (M + 31) / 32;
    const int s0 = (int)((long long)steps_all * blockIdx.y / gridDim.y), s1 = (int)((long long)steps_all * (blockIdx.y + 1) / gridDim.y);

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[PPT][EP];
#pragma unroll
    for (int q = 0; q < PPT; ++q)
#pragma unroll
        for (int e = 0; e < EP; ++e) bsum[q][e] = 0.f;

    for (int s = s0; s < s1; ++s) {
        const int m0 = s * 32;
        // ---- stage: every thread moves PPT 16-byte pieces of each operand, transposing on the way into LDS
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int pc = tid + 256 * q, r = pc / PPR, pp = pc % PPR;      // token row in the step, piece in the row
            const int m = m0 + r;
            const float live = m < M ? 1.0f : 0.0f;
            const int mc = m < M ? m : M - 1;
            const int cn = n0 + pp * EP, ck = k0 + pp * EP;
            float fy[EP], fx[EP];
            Vec<T>::load(dY + (size_t)mc * ldy + (cn < N ? cn : N - EP), fy);   // clamped, unconditional
            Vec<T>::load(X + (size_t)mc * ldx + (ck < K ? ck : K - EP), fx);
            const float my = (cn < N) ? live : 0.0f, mx = (ck < K) ? live : 0.0f;
#pragma unroll
            for (int e = 0; e < EP; ++e) {
                store1(reinterpret_cast<T*>(Yt + (pp * EP + e) * STR) + r, fy[e] * my);
                store1(reinterpret_cast<T*>(Xt + (pp * EP + e) * STR) + r, fx[e] * mx);
                bsum[q][e] += fy[e] * my;
            }
        }
        __syncthreads();
        // ---- 2 x 2 MFMAs: A = dY^T rows n (8 tokens of lane group fg), B = X^T rows k, same token slots
        Frag<T> a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) load_frag(a[i], reinterpret_cast<const T*>(Yt + (wn * 32 + i * 16 + fr) * STR) + fg * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) load_frag(b[j], reinterpret_cast<const T*>(Xt + (wk * 32 + j * 16 + fr) * STR) + fg * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma16(acc[i][j], a[i], b[j]);
        __syncthreads();
    }
    // ---- partial tile: D row = 4*fg + reg -> n, col = fr -> k
    float* wp = ws_w + (size_t)blockIdx.y * N * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 32 + i * 16 + fg * 4 + r, k = k0 + wk * 32 + j * 16 + fr;
                if (n < N && k < K) wp[(size_t)n * K + k] = acc[i][j][r];
            }
    if (do_bias) {   // column sums of this chunk's dY tile: the 32 token-row threads of a piece meet in LDS
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int pc = tid + 256 * q, r = pc / PPR, pp = pc % PPR;
#pragma unroll
            for (int e = 0; e < EP; ++e) Bs[r][pp * EP + e] = bsum[q][e];
        }
        __syncthreads();
        if (tid < 64 && n0 + tid < N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) t += Bs[r][tid];
            ws_b[(size_t)blockIdx.y * N + n0 + tid] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 weight gradient, second version: 128 x 128 output tile per workgroup (4 waves x 64 x 64), tokens in steps of 32.
// dY[32][128] and X[32][128] go to LDS exactly as they sit in HBM (token-major rows, 16-byte pieces: coalesced loads, 16-byte LDS
// stores, register-staged double buffer); the MFMA operands -- 8 consecutive TOKENS of one column per lane -- come out of LDS
// through gfx950's transposing read ds_read_b64_tr_b16: within a 16-lane group, lane q addresses the 8-byte chunk
// [row q>>2][4 columns (q&3)*4..] of a 4 x 16 block and receives column q of its 4 rows (mapping probed on the hardware:
// scripts/ubench_hip/trread.hip).  Two reads give a lane its 8 tokens.  No transposing stores, 16 MFMAs per wave per step
// instead of 4, and every dY / X byte is read by N/128 resp. K/128 workgroups instead of N/64, K/64.
// LDS layout [32 rows][256 B] with the 8-byte chunk index XORed by 4 * ((row & 3) | ((row >> 3) & 1) << 2): the 8 rows x 4 chunks a
// half-wave reads in one instruction land on 32 distinct bank pairs (conflict-free), and 16-byte stores stay contiguous.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned wg2_off(int row, int chunk) {     // byte offset of 8-byte chunk `chunk` (0..31) of token row `row`
    const int rho = (row & 3) | (((row >> 3) & 1) << 2);
    return (unsigned)(row * 256 + ((chunk ^ (rho << 2)) << 3));
}

// Workgroup -> (tile, token chunk): the tiles of ONE chunk read the same dY / X rows (dY once per K tile column, X once per N tile
// row), and workgroup L runs on XCD L % 8 with its own L2.  As a (tiles, chunks) grid the tiles of a chunk landed on eight different
// XCDs and every one of them fetched the rows for itself: 460 MB of HBM traffic per launch against 129 MB algorithmic, 77 GB of the
// 337 GB a training step moved (profiles/r03_pmc_traffic_train.json, first version).  Now the grid is linear and XCD x walks
// chunks x, x + 8, ... with the tiles of a chunk on consecutive workgroups of that XCD.  Placement only: same partials, same sums.
template <typename T>
__global__ __launch_bounds__(256, 2) void linear_wgrad2_kernel(const T* __restrict__ dY, int ldy, const T* __restrict__ X, int ldx,
                                                               float* __restrict__ ws_w, float* __restrict__ ws_b, int M, int N, int K, int S, int xcd_map) {
    __shared__ __attribute__((aligned(16))) char Ys[2][32 * 256];
    __shared__ __attribute__((aligned(16))) char Xs[2][32 * 256];
    __shared__ float Bs[16][128 + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int k_tiles = (K + 127) / 128, tiles = ((N + 127) / 128) * k_tiles;
    int tile, chunk;
    if (xcd_map) {
        const int xcd = (int)blockIdx.x & 7, seq = (int)blockIdx.x >> 3;
        tile = seq % tiles;
        chunk = (seq / tiles) * 8 + xcd;
        if (chunk >= S) return;                              // whole workgroup (the grid is padded to 8 chunk lanes)
    } else {
        tile = (int)blockIdx.x % tiles;
        chunk = (int)blockIdx.x / tiles;
    }
    const int n0 = (tile / k_tiles) * 128, k0 = (tile % k_tiles) * 128;
    const int wn = wave >> 1, wk = wave & 1;                 // 64 x 64 quadrant of this wave
    const bool do_bias = (tile % k_tiles) == 0;
    const int steps_all = (M + 31) / 32;
    const int s0 = (int)((long long)steps_all * chunk / S), s1 = (int)((long long)steps_all * (chunk + 1) / S);

    // loader role: pieces pc = tid and tid + 256 of each [32][128] tile: row pc >> 4, 16-byte piece pc & 15 (same piece for both)
    const int prow = tid >> 4, pseg = tid & 15;
    const int cn = n0 + pseg * 8, ck = k0 + pseg * 8;
    const bool okn = cn < N, okk = ck < K;                   // N, K are multiples of 8: a piece is entirely in or out
    const T* ysrc = dY + (okn ? cn : 0);
    const T* xsrc = X + (okk ? ck : 0);
    u32x4 ry[2], rx[2];
    auto fetch = [&](int s) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = s * 32 + prow + 16 * q;
            const int mc = m < M ? m : M - 1;                // clamped, unconditional; masked below
            ry[q] = *reinterpret_cast<const u32x4*>(ysrc + (size_t)mc * ldy);
            rx[q] = *reinterpret_cast<const u32x4*>(xsrc + (size_t)mc * ldx);
        }
    };
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    auto stash = [&](int s, int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = prow + 16 * q;
            const bool live = s * 32 + row < M;
            const u32x4 zy = (live && okn) ? ry[q] : u32x4{0, 0, 0, 0}, zx = (live && okk) ? rx[q] : u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(Ys[buf] + wg2_off(row, pseg * 2)) = zy;      // chunks 2*pseg, 2*pseg+1 stay adjacent under the XOR
            *reinterpret_cast<u32x4*>(Xs[buf] + wg2_off(row, pseg * 2)) = zx;
            if (do_bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { float lo, hi; unpack2<T>(zy[e], lo, hi); bsum[2 * e] += lo; bsum[2 * e + 1] += hi; }
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned ybase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&Ys[0][0];
    const unsigned xbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&Xs[0][0];
    // operand addressing: lane (group fg, q = fr) reads token rows 8*fg + (fr >> 2) [+4 = +1024 B for the second half: same XOR
    // pattern], chunk (col0 / 4) + (fr & 3) of the 16-column tile i
    const int trow = 8 * fg + (fr >> 2);
    unsigned yaddr[4], xaddr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        yaddr[i] = ybase + wg2_off(trow, (wn * 64 + i * 16) / 4 + (fr & 3));
        xaddr[i] = xbase + wg2_off(trow, (wk * 64 + i * 16) / 4 + (fr & 3));
    }

    if (s0 < s1) { fetch(s0); stash(s0, 0); }
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const int buf = (s - s0) & 1;
        if (s + 1 < s1) fetch(s + 1);                       // global loads of the next step fly under this step's MFMAs
        // all 16 transposing reads of the step and their wait in ONE asm statement (early-clobber outputs): hipcc does not track
        // asm loads, so nothing may touch the destinations before the s_waitcnt inside the string (cdna_hip_programming.md 5.7)
        u32x2 y0[4], y1[4], x0[4], x1[4];
        const unsigned bo = (unsigned)buf * 8192u;
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
        Frag<T> a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i].v = u32x4{y0[i][0], y0[i][1], y1[i][0], y1[i][1]};
            b[i].v = u32x4{x0[i][0], x0[i][1], x1[i][0], x1[i][1]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma16(acc[i][j], a[i], b[j]);
        if (s + 1 < s1) stash(s + 1, buf ^ 1);              // the other buffer: last read in step s-1, all waves passed the barrier since
        __syncthreads();
    }
    // ---- partial tile: D row = 4*fg + reg -> n, col = fr -> k
    float* wp = ws_w + (size_t)chunk * N * K;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + i * 16 + fg * 4 + r, k = k0 + wk * 64 + j * 16 + fr;
                if (n < N && k < K) wp[(size_t)n * K + k] = acc[i][j][r];
            }
    if (do_bias) {   // column sums of this chunk's dY slab: the 16 row-threads of a piece meet in LDS, fixed order
#pragma unroll
        for (int e = 0; e < 8; ++e) Bs[prow][pseg * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 128 && n0 + tid < N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += Bs[r][tid];
            ws_b[(size_t)chunk * N + n0 + tid] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient, third version (round 4): linear_wgrad2 with the token tiles staged by LDS-DMA and 64-token steps.
//   * `buffer_load_dwordx4 ... lds` moves dY[64][128] and X[64][128] straight into LDS (8 instructions per wave and step, each 4 token
//     rows x 256 bytes); the XOR placement of the 16-byte pieces that keeps the transposing reads conflict-free (wg2_off) is obtained by
//     permuting WHICH global piece a lane fetches.  No staging registers, no ds_write pass; token rows past M and columns past N / K
//     read zeros through the buffer descriptor.
//   * 64 tokens per barrier (two 32-row sub-tiles per buffer, 64 KiB of LDS for the double buffer, two workgroups per CU): 32 MFMAs per
//     wave between barriers instead of 16.
// Measured as a prototype in round 3 (scripts/ubench_hip/wgrad_dma.hip, profiles/r03_wgrad_dma.txt): +23 % at dec1, +38 % at dec0 over
// linear_wgrad2.  The bias-summing workgroups (first K tile column) read their dY pieces back from LDS, same thread -> row mapping and
// the same token order as the register path.  Chunks are ranges of 64-token steps, partial tiles as before.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void linear_wgrad3_kernel(const T* __restrict__ dY, int ldy, const T* __restrict__ X, int ldx,
                                                               float* __restrict__ ws_w, float* __restrict__ ws_b, int M, int N, int K, int S) {
    constexpr int TOK = 64, TB = TOK * 256;                  // tokens per step, bytes of one operand tile
    __shared__ __attribute__((aligned(1024))) char Ys[2][TB];
    __shared__ __attribute__((aligned(1024))) char Xs[2][TB];
    __shared__ float Bs[16][128 + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int k_tiles = (K + 127) / 128, tiles = ((N + 127) / 128) * k_tiles;
    const int xcd = (int)blockIdx.x & 7, seq = (int)blockIdx.x >> 3;      // XCD x walks chunks x, x + 8, ...: see linear_wgrad2_kernel
    const int tile = seq % tiles, chunk = (seq / tiles) * 8 + xcd;
    if (chunk >= S) return;
    const int n0 = (tile / k_tiles) * 128, k0 = (tile % k_tiles) * 128;
    const int wn = wave >> 1, wk = wave & 1;
    const bool do_bias = (tile % k_tiles) == 0;
    const int steps_all = (M + TOK - 1) / TOK;
    const int s0 = (int)((long long)steps_all * chunk / S), s1 = (int)((long long)steps_all * (chunk + 1) / S);

    const unsigned ybase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&Ys[0][0];
    const unsigned xbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&Xs[0][0];
    const unsigned long long ya = (unsigned long long)(uintptr_t)dY, xa = (unsigned long long)(uintptr_t)X;
    const u32x4 rsy = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ya), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ya >> 32)) & 0xffffu,
                       (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(M - 1) * (unsigned)ldy + (unsigned)N) * 2u)), 0x00020000u};
    const u32x4 rsx = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xa), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xa >> 32)) & 0xffffu,
                       (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(M - 1) * (unsigned)ldx + (unsigned)K) * 2u)), 0x00020000u};
    // wave w moves rows [16 w, 16 w + 16) of both tiles: instruction q = 4 rows; lane l lands at row 4 q + l / 16, position l % 16 and
    // fetches piece (l % 16) ^ (rho(row % 32) << 1), rho as in wg2_off
    unsigned voy[4], vox[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (wave * 4 + q) * 4 + (lane >> 4), r32 = row & 31;
        const int rho = (r32 & 3) | (((r32 >> 3) & 1) << 2);
        const int pc = (lane & 15) ^ (rho << 1);
        voy[q] = (n0 + pc * 8 < N) ? ((unsigned)row * (unsigned)ldy + (unsigned)(n0 + pc * 8)) * 2u : 0xffffff00u;
        vox[q] = (k0 + pc * 8 < K) ? ((unsigned)row * (unsigned)ldx + (unsigned)(k0 + pc * 8)) * 2u : 0xffffff00u;
    }
    auto dma_step = [&](int s, int buf) {
        const unsigned sy = (unsigned)s * (unsigned)(TOK * 2) * (unsigned)ldy, sx = (unsigned)s * (unsigned)(TOK * 2) * (unsigned)ldx;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned rb = (unsigned)((wave * 4 + q) * 4) * 256u;
            dma_buffer_to_lds(rsy, voy[q], sy, ybase + (unsigned)buf * TB + rb);
            dma_buffer_to_lds(rsx, vox[q], sx, xbase + (unsigned)buf * TB + rb);
        }
    };
    // bias sums (first K tile column only): the thread that staged piece pseg of rows prow + 16 q in the register path reads them back
    const int prow = tid >> 4, pseg = tid & 15;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

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
    if (s0 < s1) { dma_step(s0, 0); wait_dma<0>(); }
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const int buf = (s - s0) & 1;
        if (s + 1 < s1) dma_step(s + 1, buf ^ 1);           // the other buffer: last read in step s-1, all waves passed the barrier since
        __builtin_amdgcn_sched_barrier(0);
        if (do_bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = prow + 16 * q;
                const u32x4 zy = *reinterpret_cast<const u32x4*>(Ys[buf] + (row >> 5) * 8192 + wg2_off(row & 31, pseg * 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) { float lo, hi; unpack2<T>(zy[e], lo, hi); bsum[2 * e] += lo; bsum[2 * e + 1] += hi; }
            }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
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
            Frag<T> a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i].v = u32x4{y0[i][0], y0[i][1], y1[i][0], y1[i][1]};
                b[i].v = u32x4{x0[i][0], x0[i][1], x1[i][0], x1[i][1]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(acc[i][j], a[i], b[j]);
        }
        if (s + 1 < s1) wait_dma<0>();
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
                if (n < N && k < K) wp[(size_t)n * K + k] = acc[i][j][r];
            }
    if (do_bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) Bs[prow][pseg * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 128 && n0 + tid < N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += Bs[r][tid];
            ws_b[(size_t)chunk * N + n0 + tid] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient, fourth version (round 4): 256 x 256 output tile per workgroup, eight waves, four-deep LDS-DMA ring.
// linear_wgrad3 ran at 13 % of the MFMA rate on the wide layers (280-330 TFLOP/s at dec1 / dec0, profiles/r04_run11.txt) with HBM at
// 1.4 TB/s and LDS at half its rate: every 64-token step waits for ONE step of prefetch (64 KiB in flight per CU), and a 128 x 128 tile
// needs 512 bytes of operands per token for 32 K MACs -- the L2 -> LDS fill ran at 4.4-5 TB/s and that was the limit.  Here
//   * the tile is 256 x 256 (wave (wn, wk) owns 64 rows x 128 columns: 32 MFMAs per 24 transposing reads): 1 KiB of operands per token
//     for 128 K MACs, half the fill per flop;
//   * a stage is 32 tokens = two dY panels + two X panels of [32][128] in the layout of the third version (wg2_off, so DMA placement and
//     transposing reads are the proven ones), four stages of 32 KiB: three in flight while one is read, `s_waitcnt vmcnt(8/4/0)` counts
//     the wave's own 4 instructions per stage, one barrier per stage;
//   * the bias gradient is the product with a fragment of ones on the matrix pipe (4 MFMAs per stage in the wk = 0 waves of the first
//     K tile column) instead of 70 VALU instructions per step in half of all workgroups.
// Shapes with N or K not a multiple of 256 keep the third version.  Same partial-tile + ordered column sum as before (bit-reproducible;
// not bit-identical to the third version: the chunking and the bias summation order differ).
// ---------------------------------------------------------------------------------------------------------------
typedef short s16x4_t __attribute__((ext_vector_type(4)));
template <typename T> struct OnesWord;
template <> struct OnesWord<bf16> { static constexpr unsigned v = 0x3f803f80u; };
template <> struct OnesWord<f16> { static constexpr unsigned v = 0x3c003c00u; };
constexpr int WG4_TOK = 32, WG4_PANEL = WG4_TOK * 256, WG4_STAGE = 4 * WG4_PANEL, WG4_NST = 4;
template <typename T>
__global__ __launch_bounds__(512, 1) void linear_wgrad4_kernel(const T* __restrict__ dY, int ldy, const T* __restrict__ X, int ldx,
                                                               float* __restrict__ ws_w, float* __restrict__ ws_b, int M, int N, int K, int S) {
    __shared__ __attribute__((aligned(1024))) char smem[WG4_NST * WG4_STAGE];
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int k_tiles = K / 256, tiles = (N / 256) * k_tiles;
    const int xcd = (int)blockIdx.x & 7, seq = (int)blockIdx.x >> 3;      // XCD x walks chunks x, x + 8, ...: see linear_wgrad2_kernel
    const int tile = seq % tiles, chunk = (seq / tiles) * 8 + xcd;
    if (chunk >= S) return;
    const int n0 = (tile / k_tiles) * 256, k0 = (tile % k_tiles) * 256;
    const int wn = wave >> 1, wk = wave & 1;
    const bool do_bias = (tile % k_tiles) == 0 && wk == 0;               // wave-uniform
    const int steps_all = (M + WG4_TOK - 1) / WG4_TOK;
    const int s0 = (int)((long long)steps_all * chunk / S), s1 = (int)((long long)steps_all * (chunk + 1) / S);

    lds_char* const lds = (lds_char*)&smem[0];
    const unsigned lbase = (unsigned)(uintptr_t)lds;
    // ---- DMA role: wave w moves rows [16 (w & 1), +16) of panel w >> 1 (panels 0, 1: dY columns n0 + 0 / + 128; 2, 3: X columns k0 + 0 / + 128),
    // instruction q = 4 rows; lane l lands at row 4 q + l / 16, position l % 16 and fetches piece (l % 16) ^ (rho(row) << 1)
    const int panel = wave >> 1;
    const bool isy = panel < 2;
    const T* src = isy ? dY : X;
    const int ld = isy ? ldy : ldx, lim = isy ? N : K, c0 = (isy ? n0 : k0) + (panel & 1) * 128;
    const unsigned long long sa = (unsigned long long)(uintptr_t)src;
    const u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sa >> 32)) & 0xffffu,
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(M - 1) * (unsigned)ld + (unsigned)lim) * 2u)), 0x00020000u};
    unsigned vo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (wave & 1) * 16 + q * 4 + (lane >> 4);
        const int rho = (row & 3) | (((row >> 3) & 1) << 2);
        const int pc = (lane & 15) ^ (rho << 1);
        vo[q] = (c0 + pc * 8 < lim) ? ((unsigned)row * (unsigned)ld + (unsigned)(c0 + pc * 8)) * 2u : 0xffffff00u;
    }
    const unsigned drow = lbase + (unsigned)panel * WG4_PANEL + (unsigned)((wave & 1) * 16) * 256u;
    auto issue = [&](int s) {
        const unsigned so = (unsigned)s * (unsigned)(WG4_TOK * 2) * (unsigned)ld;
        const unsigned db = drow + (unsigned)((s - s0) & (WG4_NST - 1)) * WG4_STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_buffer_to_lds(rs, vo[q], so, db + (unsigned)q * 1024u);
    };

    f32x4 acc[4][8], bacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    Frag<T> ones;
    ones.v = u32x4{OnesWord<T>::v, OnesWord<T>::v, OnesWord<T>::v, OnesWord<T>::v};
    // ---- operand addressing (as in the third version): lane (fg, fr) reads token rows 8 fg + (fr >> 2) and + 4 (= + 1024 bytes, same rho),
    // 8-byte chunk (column / 4) + (fr & 3) of a 16-column fragment; the chunk index is XORed with rho << 2
    const int trow = 8 * fg + (fr >> 2);
    const int rho = (trow & 3) | (((trow >> 3) & 1) << 2);
    unsigned ao[4], bo[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) ao[i] = (unsigned)(wn >> 1) * WG4_PANEL + wg2_off(trow, (wn & 1) * 16 + i * 4 + (fr & 3));
#pragma unroll
    for (int j = 0; j < 8; ++j) bo[j] = (unsigned)(2 + wk) * WG4_PANEL + wg2_off(trow, j * 4 + (fr & 3));
    (void)rho;
    auto frag = [&](unsigned off) {
        Frag<T> f;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off + 1024));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        f.v = u32x4{l2[0], l2[1], h2[0], h2[1]};
        return f;
    };

    const int ns = s1 - s0;
    for (int d = 0; d < WG4_NST - 1 && d < ns; ++d) issue(s0 + d);
    for (int s = s0; s < s1; ++s) {
        const int rem = s1 - 1 - s;                          // stages issued after s: min(rem, 2) are still in flight behind it
        if (rem >= 2) wait_dma<8>();
        else if (rem == 1) wait_dma<4>();
        else wait_dma<0>();
        __syncthreads();                                     // stage s landed for every wave; every wave is done reading stage s - 1
        if (s + WG4_NST - 1 < s1) issue(s + WG4_NST - 1);    // ... whose buffer this refills
        const unsigned sb = (unsigned)((s - s0) & (WG4_NST - 1)) * WG4_STAGE;
        Frag<T> a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = frag(sb + ao[i]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            Frag<T> b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = frag(sb + bo[h * 4 + j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(acc[i][h * 4 + j], a[i], b[j]);
        }
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) mma16(bacc[i], a[i], ones);       // D[row n][every column] = sum over the stage's tokens of dY[token][n]
        }
    }
    // ---- partial tile: D row = 4 fg + reg -> n, col = fr -> k
    float* wp = ws_w + (size_t)chunk * N * K;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + i * 16 + fg * 4 + r, k = k0 + wk * 128 + j * 16 + fr;
                wp[(size_t)n * K + k] = acc[i][j][r];
            }
    if (do_bias && fr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) ws_b[(size_t)chunk * N + n0 + wn * 64 + i * 16 + fg * 4 + r] = bacc[i][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Window attention backward (WindowAttention.forward, model.py:494-519, without the projections):
//   P = softmax(q k^T + bias + mask);   dV = P^T dO;   dP = dO V^T;   dS = P o (dP - rowsum(dP o P));
//   dq = dS k;   dk = dS^T q;   dbias[h] = sum over windows of dS        (q is the SCALED query the forward stores)
// One workgroup per (head, chunk of windows), 4 waves; everything is recomputed from q, k, v^T.  All five products run
// on the MFMA with BOTH operands read from LDS tiles kept in the orientation the product needs ([row][contraction index]),
// so each tile exists token-major and d-major, and P / dS in both orientations -- simple and layout-safe; chaining the
// accumulators into the next operands as the forward kernel does is the follow-up.  Wave w owns query rows 16w..16w+15 of
// S, P, dP, dS (softmax row statistics: 4 in-lane tiles + a 16-lane DPP reduction) and keeps its slice of dbias in
// registers across the windows of the chunk.
// ---------------------------------------------------------------------------------------------------------------
// HD = 32 (every stage of Uformer_S / _B) or 16 (Uformer_T, utils/model_utils.py:66-67: embed_dim 16 with the same head counts).  The
// token-major tiles keep 32-wide rows for both: with HD = 16 columns [16, 32) are zero-filled once, so the one 32-deep MFMA k-step of
// S = q k^T and dP = dO v^T contracts over 16 real and 16 zero slots; the d-major tiles and the output tiles simply have HD / 16 of them.
template <typename T, int HD>
__global__ __launch_bounds__(256) void window_attn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt,
                                                              const float* __restrict__ bias_dense, const float* __restrict__ mask, int n_mask,
                                                              const T* __restrict__ dO, int ldo, T* __restrict__ dq, T* __restrict__ dk,
                                                              T* __restrict__ dvt, T* __restrict__ dqkv, float qscale, float* __restrict__ ws_bias, int n_windows,
                                                              int heads, int H, int W, int shift) {
    constexpr int SZ = sizeof(T), EP = 16 / SZ, NDT = HD / 16;
    static_assert(HD == 16 || HD == 32, "head_dim 16 or 32");
    constexpr int SD = 32 * SZ + 16, ST = 64 * SZ + 16;       // row strides: [token][32 d slots] tiles, [d or token][token] tiles
    constexpr int NPC = 64 * HD / EP;                         // 16-byte pieces per tile (HD = 32: 256 bf16 / 512 f32; HD = 16: half of that)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;                 char* Ks = Qs + 64 * SD;  char* Vs = Ks + 64 * SD;  char* Gs = Vs + 64 * SD;   // token-major
    char* QT = Gs + 64 * SD;         char* KT = QT + 32 * ST;  char* GT = KT + 32 * ST;                              // d-major
    char* Ps = GT + 32 * ST;         char* PT = Ps + 64 * ST;  char* Ds = PT + 64 * ST;  char* DT = Ds + 64 * ST;   // 64 x 64
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.x;
    const int w0 = (int)((long long)n_windows * blockIdx.y / gridDim.y), w1 = (int)((long long)n_windows * (blockIdx.y + 1) / gridDim.y);
    const int i0 = wave * 16;
    const int nWc = W >> 3, nW = (H >> 3) * nWc;

    // bias of this wave's rows, cached for all windows: element (t, r) = bias[h][i0 + 4fg + r][16t + fr]
    float br[4][4];
    f32x4 adb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        adb[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) br[t][r] = bias_dense[(size_t)h * 4096 + (i0 + 4 * fg + r) * 64 + 16 * t + fr];
    }

    if constexpr (HD == 16) {        // the zero half of the 32-slot rows of the four token-major tiles (never written again)
        for (int e = tid; e < 4 * 64 * 16; e += 256) {
            const int tile = e >> 10, row = (e >> 4) & 63, col = 16 + (e & 15);
            store1(reinterpret_cast<T*>(smem + tile * 64 * SD + row * SD) + col, 0.0f);
        }
    }
    for (int bw = w0; bw < w1; ++bw) {
        const size_t base = ((size_t)bw * heads + h) * (64 * HD);
        // ---- stage q, k, dO (token-major + transposed) and v (from v^T, transposed back to token-major)
#pragma unroll
        for (int pc = tid; pc < NPC; pc += 256) {
            {
                const int i = pc / (HD / EP), pp = pc % (HD / EP);           // token row, piece of the 32 d
                float fq[EP], fk[EP], fgd[EP];
                Vec<T>::load(q + base + i * HD + pp * EP, fq);
                Vec<T>::load(k + base + i * HD + pp * EP, fk);
                Vec<T>::load(dO + ((size_t)bw * 64 + i) * ldo + h * HD + pp * EP, fgd);
                Vec<T>::store(reinterpret_cast<T*>(Qs + i * SD) + pp * EP, fq);
                Vec<T>::store(reinterpret_cast<T*>(Ks + i * SD) + pp * EP, fk);
                Vec<T>::store(reinterpret_cast<T*>(Gs + i * SD) + pp * EP, fgd);
#pragma unroll
                for (int e = 0; e < EP; ++e) {
                    store1(reinterpret_cast<T*>(QT + (pp * EP + e) * ST) + i, fq[e]);
                    store1(reinterpret_cast<T*>(KT + (pp * EP + e) * ST) + i, fk[e]);
                    store1(reinterpret_cast<T*>(GT + (pp * EP + e) * ST) + i, fgd[e]);
                }
            }
            {
                const int d = pc / (64 / EP), pp = pc % (64 / EP);           // d row of v^T, piece of the 64 tokens
                float fv[EP];
                Vec<T>::load(vt + base + d * 64 + pp * EP, fv);
#pragma unroll
                for (int e = 0; e < EP; ++e) store1(reinterpret_cast<T*>(Vs + (pp * EP + e) * SD) + d, fv[e]);
            }
        }
        __syncthreads();

        // ---- this wave's 16 query rows: S = q k^T and dP = dO v^T (one MFMA k-step: d = 32), then softmax algebra
        Frag<T> aq, ag, bk[4], bv[4];
        load_frag(aq, reinterpret_cast<const T*>(Qs + (i0 + fr) * SD) + fg * 8);
        load_frag(ag, reinterpret_cast<const T*>(Gs + (i0 + fr) * SD) + fg * 8);
        f32x4 sc[4], dp[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            load_frag(bk[t], reinterpret_cast<const T*>(Ks + (16 * t + fr) * SD) + fg * 8);
            load_frag(bv[t], reinterpret_cast<const T*>(Vs + (16 * t + fr) * SD) + fg * 8);
            sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16(sc[t], aq, bk[t]);       // D[row = 4fg + r -> query i0+4fg+r][col = fr -> key 16t+fr]
            mma16(dp[t], ag, bv[t]);
        }
        const int wi = bw % nW;
        const bool last_r = shift > 0 && (wi / nWc) == (H >> 3) - 1;
        const bool last_c = shift > 0 && (wi % nWc) == nWc - 1;
        const float* mk = mask ? mask + (size_t)(bw % n_mask) * 4096 : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = i0 + 4 * fg + r;
            const bool q_lo_y = (qi >> 3) >= 4, q_lo_x = (qi & 7) >= 4;
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kj = 16 * t + fr;
                float v = sc[t][r] + br[t][r];
                if (mk) v += mk[qi * 64 + kj];
                // SW-MSA mask, model.py:924-942: -100 where the region ids of query and key differ
                const bool k_lo_y = (kj >> 3) >= 4, k_lo_x = (kj & 7) >= 4;
                if ((last_r && (k_lo_y != q_lo_y)) || (last_c && (k_lo_x != q_lo_x))) v += -100.0f;
                sc[t][r] = v;
                mx = fmaxf(mx, v);
            }
            mx = allreduce<RedMax, 16>(mx);                       // the 16 lanes of a lane group hold one query row
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) { sc[t][r] = __expf(sc[t][r] - mx); sum += sc[t][r]; }
            const float inv = 1.0f / allreduce<RedSum, 16>(sum);
            float dot = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) { sc[t][r] *= inv; dot += sc[t][r] * dp[t][r]; }
            dot = allreduce<RedSum, 16>(dot);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float ds = sc[t][r] * (dp[t][r] - dot);
                adb[t][r] += ds;
                const int kj = 16 * t + fr;
                store1(reinterpret_cast<T*>(Ps + qi * ST) + kj, sc[t][r]);
                store1(reinterpret_cast<T*>(PT + kj * ST) + qi, sc[t][r]);
                store1(reinterpret_cast<T*>(Ds + qi * ST) + kj, ds);
                store1(reinterpret_cast<T*>(DT + kj * ST) + qi, ds);
            }
        }
        __syncthreads();

        // ---- dq[i][d] = sum_j dS[i][j] k[j][d];  dk[j][d] = sum_i dS[i][j] q[i][d];  dv^T[d][j] = sum_i dO[i][d] P[i][j]
        // wave w: 16-row tile w of dq and dk (both d tiles), 16-column tile w of dv^T (both d tiles); 64-long contraction
        f32x4 oq[NDT], ok[NDT], ov[NDT];
#pragma unroll
        for (int c = 0; c < NDT; ++c) { oq[c] = f32x4{0.f, 0.f, 0.f, 0.f}; ok[c] = oq[c]; ov[c] = oq[c]; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag<T> a_ds, a_dst, b_pt;
            load_frag(a_ds, reinterpret_cast<const T*>(Ds + (i0 + fr) * ST) + ks * 32 + fg * 8);      // rows i, slots j
            load_frag(a_dst, reinterpret_cast<const T*>(DT + (i0 + fr) * ST) + ks * 32 + fg * 8);     // rows j, slots i
            load_frag(b_pt, reinterpret_cast<const T*>(PT + (i0 + fr) * ST) + ks * 32 + fg * 8);      // cols j, slots i
#pragma unroll
            for (int c = 0; c < NDT; ++c) {
                Frag<T> b_kt, b_qt, a_gt;
                load_frag(b_kt, reinterpret_cast<const T*>(KT + (16 * c + fr) * ST) + ks * 32 + fg * 8);   // cols d, slots j
                load_frag(b_qt, reinterpret_cast<const T*>(QT + (16 * c + fr) * ST) + ks * 32 + fg * 8);   // cols d, slots i
                load_frag(a_gt, reinterpret_cast<const T*>(GT + (16 * c + fr) * ST) + ks * 32 + fg * 8);   // rows d, slots i
                mma16(oq[c], a_ds, b_kt);      // D[row i][col d]
                mma16(ok[c], a_dst, b_qt);     // D[row j][col d]
                mma16(ov[c], a_gt, b_pt);      // D[row d][col j]
            }
        }
        if (dqkv) {
            // merged form: the gradient of the fused q|k|v projection output, T[n_windows*64][3C] in window-row order, channel
            // h*32 + d of each third; dq is multiplied by the query scale here (it is the gradient wrt the SCALED query, model.py:497)
            const int C3 = 3 * heads * HD;
            T* row0 = dqkv + (size_t)bw * 64 * C3 + h * HD;
#pragma unroll
            for (int c = 0; c < NDT; ++c) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    store1(row0 + (size_t)(i0 + 4 * fg + r) * C3 + 16 * c + fr, oq[c][r] * qscale);
                    store1(row0 + (size_t)(i0 + 4 * fg + r) * C3 + heads * HD + 16 * c + fr, ok[c][r]);
                }
                // dv^T tile: lane holds d = 16c + 4fg + 0..3 of key token i0 + fr: four consecutive channels of one row
                T* dst = row0 + (size_t)(i0 + fr) * C3 + 2 * heads * HD + 16 * c + 4 * fg;
                if constexpr (SZ == 2) *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<T>(ov[c][0], ov[c][1]), pack2<T>(ov[c][2], ov[c][3])};
                else *reinterpret_cast<f32x4*>(dst) = ov[c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < NDT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    store1(dq + base + (i0 + 4 * fg + r) * HD + 16 * c + fr, oq[c][r]);
                    store1(dk + base + (i0 + 4 * fg + r) * HD + 16 * c + fr, ok[c][r]);
                    store1(dvt + base + (16 * c + 4 * fg + r) * 64 + i0 + fr, ov[c][r]);
                }
        }
        __syncthreads();   // the next window overwrites the tiles
    }
    float* wb = ws_bias + ((size_t)blockIdx.y * heads + h) * 4096;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) wb[(i0 + 4 * fg + r) * 64 + 16 * t + fr] = adb[t][r];
}


// ---------------------------------------------------------------------------------------------------------------
// Window attention backward, second version (round 4; 2-byte operand types): the accumulators of S / dP are chained straight into the
// operands of the three output products, as the forward kernel does -- P and dS never touch LDS, nothing is stored transposed.
//
// A chained accumulator D[row][col] can only be the B operand of a product that contracts over its ROWS with output column = its column
// (FragFromAcc).  dq contracts over keys, dk and dv over queries, so every wave evaluates BOTH orientations of the 64 x 64 tiles:
//   phase A (wave w = query tile w):  S^T[k][q] = K Q^T,  dP^T[k][q] = V dO^T  -> softmax statistics per query (in-lane + xor 16 / 32),
//            dS^T -> dbias slice, row statistics (max, 1 / sum, dot) to LDS, and dq^T[d][q] = sum_k K^T[d][k] dS^T[k][q]  (dS chained);
//   phase B (wave w = key tile w):    S[q][k] = Q K^T,  dP[q][k] = dO V^T with the statistics of phase A -> P, dS, then
//            dk^T[d][k] = sum_q Q^T[d][q] dS[q][k]  and  dv^T[d][k] = sum_q dO^T[d][q] P[q][k]                      (dS, P chained).
// The 2 x 16 extra MFMAs per window are free next to what they replace: the first version wrote q, k, dO transposed and P, dS in both
// orientations into LDS with 2-byte scalar stores (~24 K per window and head) and ran at ~100 TFLOP/s (profiles/r03_train_final_kernel_stats.csv:
// 6.6 ms of a 79 ms training step).  LDS holds only the four operand tiles as they sit in HBM (q, k, dO token-major [64][32], v^T [32][64],
// 16-byte staging stores); an operand that is needed transposed (K^T, Q^T, dO^T as A operands with 8 tokens per lane; V from v^T) comes
// out of LDS through gfx950's transposing read ds_read_b64_tr_b16 (mapping: linear_wgrad2 above).  Two barriers per window.
// Same products, the same softmax algebra; the order of the 32-deep / 64-deep sums inside an MFMA chain is the hardware's in both versions:
// results agree with the first version to operand rounding (tests/test_gpu_bwd.py compares both with the oracle).
// ---------------------------------------------------------------------------------------------------------------
#ifndef UF_ATTN_BWD2_WPS
#define UF_ATTN_BWD2_WPS 2      // waves per SIMD the kernel is bounded for: 3 (168 registers) spills 11 dwords and measured 2374 us over the nine stage shapes against 1777 at 2 (profiles/r04_run9.txt)
#endif
__device__ __forceinline__ u32x4 tr_frag(unsigned a0, unsigned a1) {     // 8 contraction slots of one column: rows a0 .. +3 and a1 .. +3 of the lane group
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1) : "memory");
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

template <typename T, int HD, bool MK>
__global__ __launch_bounds__(256, MK ? 2 : UF_ATTN_BWD2_WPS) void window_attn_bwd2_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt,
                                                               const float* __restrict__ bias_dense, const float* __restrict__ mask, int n_mask,
                                                               const T* __restrict__ dO, int ldo, T* __restrict__ dq, T* __restrict__ dk,
                                                               T* __restrict__ dvt, T* __restrict__ dqkv, float qscale, float* __restrict__ ws_bias, int n_windows,
                                                               int heads, int H, int W, int shift, int pair) {
    static_assert(sizeof(T) == 2 && (HD == 16 || HD == 32), "2-byte operand types, head_dim 16 or 32");
    constexpr int NDT = HD / 16;
    constexpr int SD = 32 * 2 + 16, ST = 64 * 2 + 16;            // row strides: [token][32 d slots] tiles, v^T [32 d][64 tokens]
    __shared__ __attribute__((aligned(16))) char Qs[64 * SD];
    __shared__ __attribute__((aligned(16))) char Ks[64 * SD];
    __shared__ __attribute__((aligned(16))) char Gs[64 * SD];
    __shared__ __attribute__((aligned(16))) char Vt[32 * ST];
    __shared__ __attribute__((aligned(16))) float Stat[3][64];     // per query row: max, 1 / sum, sum_k P dP
    __shared__ __attribute__((aligned(16))) float Bs[64][68];       // this head's relative-position bias [query][key], rows padded to 272 bytes (conflict-free 16-byte reads)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    // (head, window chunk) of this workgroup.  Launch order is head-fastest and workgroup i runs on XCD i % 8, so two neighbouring heads -- whose 64-byte pieces of
    // a dO / dqkv token row share one 128-byte line -- sat on two XCDs and each L2 fetched the whole line: dO came from memory twice (fetch 10.0 GB per step for
    // 8.1 algorithmic, profiles/r06_pmc_traffic_train.json).  Round 6: units of two heads, unit u on XCD u % 8, its two workgroups in consecutive launch rounds.
    int h = blockIdx.x, cy = blockIdx.y;
    if (pair && (heads & 1) == 0 && ((heads * gridDim.y) & 15) == 0) {
        const unsigned i = blockIdx.x + heads * blockIdx.y, u = (i >> 4) * 8 + (i & 7), hp = (unsigned)heads >> 1;
        h = (int)(2 * (u % hp) + ((i >> 3) & 1));
        cy = (int)(u / hp);
    }
    const int w0 = (int)((long long)n_windows * cy / gridDim.y), w1 = (int)((long long)n_windows * (cy + 1) / gridDim.y);
    const int i0 = wave * 16;                                      // this wave's query tile (phase A) and key tile (phase B)
    const int nWc = W >> 3, nW = (H >> 3) * nWc;
    const unsigned qb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)Qs, kb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)Ks;
    const unsigned gb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)Gs, vb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)Vt;

    // relative-position bias of this head: staged once in LDS (it cost 32 registers per lane as two per-lane caches: the kernel spilled at three waves
    // per SIMD); phase A reads 4 consecutive keys of query i0 + fr, phase B the key i0 + fr of 4 consecutive queries
    for (int e = tid; e < 64 * 16; e += 256) *reinterpret_cast<f32x4*>(&Bs[e >> 4][(e & 15) * 4]) = *reinterpret_cast<const f32x4*>(bias_dense + (size_t)h * 4096 + e * 4);
    f32x4 adb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) adb[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (HD == 16) {        // the zero halves of the 32-slot tiles (never written again): d slots 16..31 of q, k, dO and rows 16..31 of v^T
        for (int e = tid; e < 3 * 64 * 2; e += 256) {
            const int tile = e / 128, row = (e >> 1) & 63, pc = 2 + (e & 1);
            char* base = tile == 0 ? Qs : (tile == 1 ? Ks : Gs);
            *reinterpret_cast<u32x4*>(base + row * SD + pc * 16) = u32x4{0, 0, 0, 0};
        }
        for (int e = tid; e < 16 * 8; e += 256) *reinterpret_cast<u32x4*>(Vt + (16 + (e >> 3)) * ST + (e & 7) * 16) = u32x4{0, 0, 0, 0};
    }
    // tr-read addresses (tile-relative): an A operand "rows = 16 channels d of tile c, 8 contraction slots = tokens" out of a token-major tile:
    // lane (fg, fr) addresses token row (base + fr / 4) and the 4 channels 16 c + 4 (fr % 4) .. of it.  For operands that meet a CHAINED accumulator the
    // lane group's 8 slots are tokens 32 sk + 4 fg + {0..3} and 32 sk + 16 + 4 fg + {0..3} (FragFromAcc order).
    auto tr_tok = [&](unsigned base, int sk, int c) -> u32x4 {
        const unsigned a0 = base + (unsigned)((32 * sk + 4 * fg + (fr >> 2)) * SD + (16 * c + 4 * (fr & 3)) * 2);
        return tr_frag(a0, a0 + 16 * SD);
    };
    // the four operand pieces of a window travel HBM -> registers one window AHEAD (requested right after the staging barrier, stored to LDS at the top of
    // the next iteration): the global round trip hides under the previous window's products instead of standing in front of every window
    constexpr int PPR = HD / 8;                                           // 16-byte pieces per token row
    const bool ld_tok = tid < 64 * PPR, ld_vt = tid < HD * 8;
    const int ti = tid / PPR, tp = tid % PPR, vd = tid >> 3, vp = tid & 7;
    u32x4 rq = {0, 0, 0, 0}, rk = rq, rg = rq, rv = rq;
    auto fetch = [&](int bw) {
        const size_t base = ((size_t)bw * heads + h) * (64 * HD);
        if (ld_tok) {
            rq = *reinterpret_cast<const u32x4*>(q + base + ti * HD + tp * 8);
            rk = *reinterpret_cast<const u32x4*>(k + base + ti * HD + tp * 8);
            rg = *reinterpret_cast<const u32x4*>(dO + ((size_t)bw * 64 + ti) * ldo + h * HD + tp * 8);
        }
        if (ld_vt) rv = *reinterpret_cast<const u32x4*>(vt + base + vd * 64 + vp * 8);
    };
    if (w0 < w1) fetch(w0);
#pragma unroll 1
    for (int bw = w0; bw < w1; ++bw) {
        const size_t base = ((size_t)bw * heads + h) * (64 * HD);
        // ---- stage q, k, dO (token-major) and v^T as they sit in HBM: one 16-byte piece per thread and tile (HD = 16: half the threads)
        if (ld_tok) {
            *reinterpret_cast<u32x4*>(Qs + ti * SD + tp * 16) = rq;
            *reinterpret_cast<u32x4*>(Ks + ti * SD + tp * 16) = rk;
            *reinterpret_cast<u32x4*>(Gs + ti * SD + tp * 16) = rg;
        }
        if (ld_vt) *reinterpret_cast<u32x4*>(Vt + vd * ST + vp * 16) = rv;
        __syncthreads();
        if (bw + 1 < w1) fetch(bw + 1);
        const int wi = bw % nW;
        const bool last_r = shift > 0 && (wi / nWc) == (H >> 3) - 1;
        const bool last_c = shift > 0 && (wi % nWc) == nWc - 1;
        const bool edge = last_r || last_c;
        const float* mk = MK ? mask + (size_t)(bw % n_mask) * 4096 : nullptr;      // dense caller-supplied mask (inference-only argument of the reference): own instantiation

        // ================= phase A: query tile i0 .. i0 + 15 in the columns =================
        {
            Frag<T> qf, gf;                                               // B operands: column = query i0 + fr, slots d = 8 fg ..
            load_frag(qf, reinterpret_cast<const T*>(Qs + (i0 + fr) * SD) + fg * 8);
            load_frag(gf, reinterpret_cast<const T*>(Gs + (i0 + fr) * SD) + fg * 8);
            f32x4 s[4], dp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Frag<T> kf, vf;                                           // A operands: row = key 16 t + fr, slots d
                load_frag(kf, reinterpret_cast<const T*>(Ks + (16 * t + fr) * SD) + fg * 8);
                const unsigned va = vb + (unsigned)((8 * fg + (fr >> 2)) * ST + (16 * t + 4 * (fr & 3)) * 2);     // V[k][d] out of v^T[d][k]
                vf.v = tr_frag(va, va + 4 * ST);
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = s[t];
                mma16(s[t], kf, qf);                                      // D[row = key 16 t + 4 fg + r][col = query i0 + fr]
                mma16(dp[t], vf, gf);
            }
            const int qi = i0 + fr;
            const bool q_lo_y = (qi >> 3) >= 4, q_lo_x = (qi & 7) >= 4;
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 brA = *reinterpret_cast<const f32x4*>(&Bs[qi][16 * t + 4 * fg]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kj = 16 * t + 4 * fg + r;
                    float v = s[t][r] + brA[r];
                    if constexpr (MK) v += mk[qi * 64 + kj];
                    if (edge) {                                                       // SW-MSA mask, model.py:924-942: only the windows of the last row / column have one (wave-uniform)
                        const bool k_lo_y = (kj >> 3) >= 4, k_lo_x = (kj & 7) >= 4;
                        if ((last_r && (k_lo_y != q_lo_y)) || (last_c && (k_lo_x != q_lo_x))) v += -100.0f;
                    }
                    s[t][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = red_xor32<RedMax>(red_xor16<RedMax>(mx));               // the four lane groups hold the other keys of this query
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[t][r] = __expf(s[t][r] - mx); sum += s[t][r]; }
            const float inv = 1.0f / red_xor32<RedSum>(red_xor16<RedSum>(sum));
            float dot = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[t][r] *= inv; dot += s[t][r] * dp[t][r]; }
            dot = red_xor32<RedSum>(red_xor16<RedSum>(dot));
            if (fg == 0) { Stat[0][qi] = mx; Stat[1][qi] = inv; Stat[2][qi] = dot; }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[t][r] = s[t][r] * (dp[t][r] - dot); adb[t][r] += s[t][r]; }      // s = dS^T now
            // dq^T[d][q] = sum_k K^T[d][k] dS^T[k][q]
            f32x4 oq[NDT];
#pragma unroll
            for (int c = 0; c < NDT; ++c) oq[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                Frag<T> ds;
                ds.v = u32x4{pack2<T>(s[2 * sk][0], s[2 * sk][1]), pack2<T>(s[2 * sk][2], s[2 * sk][3]), pack2<T>(s[2 * sk + 1][0], s[2 * sk + 1][1]), pack2<T>(s[2 * sk + 1][2], s[2 * sk + 1][3])};
#pragma unroll
                for (int c = 0; c < NDT; ++c) {
                    Frag<T> kt_;
                    kt_.v = tr_tok(kb, sk, c);
                    mma16(oq[c], kt_, ds);                                // D[row d = 16 c + 4 fg + r][col q = i0 + fr]
                }
            }
#pragma unroll
            for (int c = 0; c < NDT; ++c) {
                if (dqkv) {      // merged form: dq (times the query scale) in the first third of the row of token i0 + fr
                    T* dst = dqkv + ((size_t)bw * 64 + qi) * (3 * heads * HD) + h * HD + 16 * c + 4 * fg;
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<T>(oq[c][0] * qscale, oq[c][1] * qscale), pack2<T>(oq[c][2] * qscale, oq[c][3] * qscale)};
                } else {
                    T* dst = dq + base + qi * HD + 16 * c + 4 * fg;
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<T>(oq[c][0], oq[c][1]), pack2<T>(oq[c][2], oq[c][3])};
                }
            }
        }
        __syncthreads();                                                  // the row statistics of all four query tiles
        // ================= phase B: key tile i0 .. i0 + 15 in the columns =================
        {
            Frag<T> kf, vf;                                               // B operands: column = key i0 + fr, slots d
            load_frag(kf, reinterpret_cast<const T*>(Ks + (i0 + fr) * SD) + fg * 8);
            const unsigned va = vb + (unsigned)((8 * fg + (fr >> 2)) * ST + (i0 + 4 * (fr & 3)) * 2);
            vf.v = tr_frag(va, va + 4 * ST);
            f32x4 s[4], dp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Frag<T> qf, gf;                                           // A operands: row = query 16 t + fr, slots d
                load_frag(qf, reinterpret_cast<const T*>(Qs + (16 * t + fr) * SD) + fg * 8);
                load_frag(gf, reinterpret_cast<const T*>(Gs + (16 * t + fr) * SD) + fg * 8);
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = s[t];
                mma16(s[t], qf, kf);                                      // D[row = query 16 t + 4 fg + r][col = key i0 + fr]
                mma16(dp[t], gf, vf);
            }
            const int kj = i0 + fr;
            const bool k_lo_y = (kj >> 3) >= 4, k_lo_x = (kj & 7) >= 4;
            f32x4 pr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 mxs = *reinterpret_cast<const f32x4*>(&Stat[0][16 * t + 4 * fg]), ivs = *reinterpret_cast<const f32x4*>(&Stat[1][16 * t + 4 * fg]);
                const f32x4 dts = *reinterpret_cast<const f32x4*>(&Stat[2][16 * t + 4 * fg]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = 16 * t + 4 * fg + r;
                    float v = s[t][r] + Bs[qi][kj];
                    if constexpr (MK) v += mk[qi * 64 + kj];
                    if (edge) {
                        const bool q_lo_y = (qi >> 3) >= 4, q_lo_x = (qi & 7) >= 4;
                        if ((last_r && (k_lo_y != q_lo_y)) || (last_c && (k_lo_x != q_lo_x))) v += -100.0f;
                    }
                    const float pv = __expf(v - mxs[r]) * ivs[r];
                    pr[t][r] = pv;
                    s[t][r] = pv * (dp[t][r] - dts[r]);                   // dS
                }
            }
            f32x4 ok[NDT], ov[NDT];
#pragma unroll
            for (int c = 0; c < NDT; ++c) { ok[c] = f32x4{0.f, 0.f, 0.f, 0.f}; ov[c] = ok[c]; }
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                Frag<T> ds, pf;
                ds.v = u32x4{pack2<T>(s[2 * sk][0], s[2 * sk][1]), pack2<T>(s[2 * sk][2], s[2 * sk][3]), pack2<T>(s[2 * sk + 1][0], s[2 * sk + 1][1]), pack2<T>(s[2 * sk + 1][2], s[2 * sk + 1][3])};
                pf.v = u32x4{pack2<T>(pr[2 * sk][0], pr[2 * sk][1]), pack2<T>(pr[2 * sk][2], pr[2 * sk][3]), pack2<T>(pr[2 * sk + 1][0], pr[2 * sk + 1][1]), pack2<T>(pr[2 * sk + 1][2], pr[2 * sk + 1][3])};
#pragma unroll
                for (int c = 0; c < NDT; ++c) {
                    Frag<T> qt_, gt_;
                    qt_.v = tr_tok(qb, sk, c);
                    gt_.v = tr_tok(gb, sk, c);
                    mma16(ok[c], qt_, ds);                                // dk^T: D[row d][col key i0 + fr]
                    mma16(ov[c], gt_, pf);                                // dv^T
                }
            }
#pragma unroll
            for (int c = 0; c < NDT; ++c) {
                const u32x2 pk = u32x2{pack2<T>(ok[c][0], ok[c][1]), pack2<T>(ok[c][2], ok[c][3])}, pv = u32x2{pack2<T>(ov[c][0], ov[c][1]), pack2<T>(ov[c][2], ov[c][3])};
                if (dqkv) {
                    T* row = dqkv + ((size_t)bw * 64 + kj) * (3 * heads * HD) + h * HD + 16 * c + 4 * fg;
                    *reinterpret_cast<u32x2*>(row + heads * HD) = pk;
                    *reinterpret_cast<u32x2*>(row + 2 * heads * HD) = pv;
                } else {
                    *reinterpret_cast<u32x2*>(dk + base + kj * HD + 16 * c + 4 * fg) = pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) store1(dvt + base + (16 * c + 4 * fg + r) * 64 + kj, ov[c][r]);
                }
            }
        }
        __syncthreads();   // the next window overwrites the tiles
    }
    float* wb = ws_bias + ((size_t)cy * heads + h) * 4096;
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(wb + (i0 + fr) * 64 + 16 * t + 4 * fg) = adb[t];
}

}  // namespace
}  // namespace uf

using namespace uf;

extern "C" int uf_gelu_bwd(const void* a, const void* dy, void* dx, long long n, uf_dtype dtype, void* stream) {
    UF_REQUIRE(a && dy && dx, UF_ERR_NULL, "uf_gelu_bwd: null pointer");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "uf_gelu_bwd: dtype %d", (int)dtype);
    const int N = dtype_half(dtype) ? 8 : 4;
    UF_REQUIRE(n > 0 && n % N == 0, UF_ERR_SHAPE, "uf_gelu_bwd: n=%lld must be a positive multiple of %d", n, N);
    UF_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0, UF_ERR_ALIGN, "uf_gelu_bwd: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long long nvec = n / N;
    const dim3 grid((unsigned)((nvec + 255) / 256));
    UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(gelu_bwd_kernel<TT>, grid, dim3(256), 0, st, (const TT*)a, (const TT*)dy, (TT*)dx, nvec));
    return check_launch("gelu_bwd");
}

extern "C" int uf_gelu_fwd(const void* a, void* y, long long n, uf_dtype dtype, void* stream) {
    UF_REQUIRE(a && y, UF_ERR_NULL, "uf_gelu_fwd: null pointer");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "uf_gelu_fwd: dtype %d", (int)dtype);
    const int N = dtype_half(dtype) ? 8 : 4;
    UF_REQUIRE(n > 0 && n % N == 0, UF_ERR_SHAPE, "uf_gelu_fwd: n=%lld must be a positive multiple of %d", n, N);
    UF_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)y % 16) == 0, UF_ERR_ALIGN, "uf_gelu_fwd: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long long nvec = n / N;
    const dim3 grid((unsigned)((nvec + 255) / 256));
    UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(gelu_fwd_kernel<TT>, grid, dim3(256), 0, st, (const TT*)a, (TT*)y, nvec));
    return check_launch("gelu_fwd");
}

extern "C" size_t uf_layernorm_bwd_workspace_bytes(int rows, int C) {
    if (rows <= 0 || C < 16) return 0;
    const int LPR = (C / 4) < 64 ? (C / 4) : 64;
    (void)LPR;
    return (size_t)LN_BWD_MAX_BLOCKS * 2 * C * sizeof(float);
}

struct LnCastArgs { void* out; const float* scale; int mode, hw, H, W, shift; };
static int layernorm_bwd_any(const char* fn, const float* x, int ld_x, const float* gamma, const void* dy, int ld_dy, int dy_is_f32, uf_dtype dtype, const float* add,
                             float* dx, int ld_dx, float* dgamma, float* dbeta, int rows, int C, int win_h, int win_w, int shift, void* ws, size_t ws_bytes, void* stream,
                             LnCastArgs ca = LnCastArgs{nullptr, nullptr, 0, 1, 8, 8, 0}) {
    UF_REQUIRE(x && gamma && dy && dx && dgamma && dbeta && ws, UF_ERR_NULL, "%s: null pointer", fn);
    UF_REQUIRE(rows > 0 && ld_x >= C && ld_dy >= C && ld_dx >= C && ld_x % 4 == 0 && ld_dy % 4 == 0 && ld_dx % 4 == 0, UF_ERR_SHAPE,
               "%s: rows=%d C=%d ld=(%d,%d,%d)", fn, rows, C, ld_x, ld_dy, ld_dx);
    UF_REQUIRE(win_h == 0 || (win_h % 8 == 0 && win_w % 8 == 0 && win_w > 0 && rows % (win_h * win_w) == 0 && (shift == 0 || shift == 4)), UF_ERR_SHAPE,
               "%s: window geometry H=%d W=%d shift=%d rows=%d", fn, win_h, win_w, shift, rows);
    UF_REQUIRE(ws_bytes >= uf_layernorm_bwd_workspace_bytes(rows, C), UF_ERR_WORKSPACE, "%s: workspace too small", fn);
    const bool f32dy = dy_is_f32 || dtype == UF_F32;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)ws;
    int slots = 0;
#define UF_LNB_CASE(CV)                                                                                                       \
    case CV: {                                                                                                                \
        constexpr int LPR = (CV / 4) < 64 ? (CV / 4) : 64, RPB = 256 / LPR;                                                   \
        const int nblk = (rows + RPB - 1) / RPB, grid = nblk < LN_BWD_MAX_BLOCKS ? nblk : LN_BWD_MAX_BLOCKS;                  \
        slots = grid;                                                                                                         \
        if (f32dy) hipLaunchKernelGGL((layernorm_bwd_kernel<CV, float>), dim3(grid), dim3(256), 0, st, x, ld_x, gamma, (const float*)dy, ld_dy, add, dx, ld_dx, partial, rows, win_h, win_w, shift, \
                                      (LnCast<float>{(float*)ca.out, ca.scale, ca.mode, ca.hw, ca.H, ca.W, ca.shift})); \
        else if (dtype == UF_F16) hipLaunchKernelGGL((layernorm_bwd_kernel<CV, f16>), dim3(grid), dim3(256), 0, st, x, ld_x, gamma, (const f16*)dy, ld_dy, add, dx, ld_dx, partial, rows, win_h, win_w, shift, \
                                                     (LnCast<f16>{(f16*)ca.out, ca.scale, ca.mode, ca.hw, ca.H, ca.W, ca.shift})); \
        else hipLaunchKernelGGL((layernorm_bwd_kernel<CV, bf16>), dim3(grid), dim3(256), 0, st, x, ld_x, gamma, (const bf16*)dy, ld_dy, add, dx, ld_dx, partial, rows, win_h, win_w, shift, \
                                (LnCast<bf16>{(bf16*)ca.out, ca.scale, ca.mode, ca.hw, ca.H, ca.W, ca.shift})); \
        break;                                                                                                                \
    }
    switch (C) {
        UF_LNB_CASE(16)
        UF_LNB_CASE(32)
        UF_LNB_CASE(64)
        UF_LNB_CASE(128)
        UF_LNB_CASE(256)
        UF_LNB_CASE(512)
        UF_LNB_CASE(1024)
        default:
            set_error("%s: C=%d unsupported (16,32,64,128,256,512,1024)", fn, C);
            return UF_ERR_UNSUPPORTED;
    }
#undef UF_LNB_CASE
    int rc = check_launch("layernorm_bwd");
    if (rc) return rc;
    launch_column_sum2(st, partial, slots, (size_t)2 * C, dgamma, C, partial + C, slots, (size_t)2 * C, dbeta, C);
    return check_launch("layernorm_bwd_finalize");
}

extern "C" int uf_layernorm_bwd(const float* x, int ld_x, const float* gamma, const float* dy, int ld_dy, float* dx, int ld_dx,
                                float* dgamma, float* dbeta, int rows, int C, void* ws, size_t ws_bytes, void* stream) {
    return layernorm_bwd_any("uf_layernorm_bwd", x, ld_x, gamma, dy, ld_dy, 1, UF_F32, nullptr, dx, ld_dx, dgamma, dbeta, rows, C, 0, 0, 0, ws, ws_bytes, stream);
}

// the same backward reading the output gradient where the block backward has it: dy of the operand type (dtype; dy_is_f32 = 1: f32) in
// window order when windowed (B, H, W, shift: x, add and dx rows are then the tokens of those window rows: window_reverse + roll back folded
// in), plus an optional second gradient `add` f32[rows][ld_dx] summed into dx (the residual path).  Saves the cast and add passes.
extern "C" int uf_layernorm_bwd_fused(const float* x, int ld_x, const float* gamma, const void* dy, int ld_dy, int dy_is_f32, const float* add, float* dx, int ld_dx,
                                      float* dgamma, float* dbeta, int B, int H, int W, int C, int windowed, int shift, uf_dtype dtype, void* ws, size_t ws_bytes,
                                      void* stream) {
    UF_REQUIRE(B > 0 && H > 0 && W > 0 && dtype_ok(dtype), UF_ERR_SHAPE, "uf_layernorm_bwd_fused: B=%d H=%d W=%d dtype=%d", B, H, W, (int)dtype);
    UF_REQUIRE(dy_is_f32 || dtype == UF_F32 || ld_dy % 8 == 0, UF_ERR_ALIGN, "uf_layernorm_bwd_fused: ld_dy=%d", ld_dy);
    return layernorm_bwd_any("uf_layernorm_bwd_fused", x, ld_x, gamma, dy, ld_dy, dy_is_f32, dtype, add, dx, ld_dx, dgamma, dbeta, B * H * W, C, windowed ? H : 0, windowed ? W : 0,
                             shift, ws, ws_bytes, stream);
}

// uf_layernorm_bwd_fused that also writes the copy of dx the next GEMM of the backward reads -- cast_out T[rows][C] = T(dx * cast_scale[image]) at
// the token's row, or at its window-order row when cast_windowed (partition geometry H, W, cast_shift) -- which is what uf_grad_fork did in a
// pass of its own (read dx, read the residual gradient, write both: 3.7 ms of a 76 ms Uformer-B training step, profiles/r04_run13.txt).  dy must
// be of the operand type (dy_is_f32 = 0, or dtype = f32).  Same dx, dgamma, dbeta as uf_layernorm_bwd_fused, bit for bit.
extern "C" int uf_layernorm_bwd_cast(const float* x, int ld_x, const float* gamma, const void* dy, int ld_dy, int dy_is_f32, const float* add, float* dx, int ld_dx,
                                     float* dgamma, float* dbeta, int B, int H, int W, int C, int windowed, int shift, uf_dtype dtype, void* cast_out,
                                     const float* cast_scale, int cast_windowed, int cast_shift, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(B > 0 && H > 0 && W > 0 && dtype_ok(dtype), UF_ERR_SHAPE, "uf_layernorm_bwd_cast: B=%d H=%d W=%d dtype=%d", B, H, W, (int)dtype);
    UF_REQUIRE(dy_is_f32 || dtype == UF_F32 || ld_dy % 8 == 0, UF_ERR_ALIGN, "uf_layernorm_bwd_cast: ld_dy=%d", ld_dy);
    UF_REQUIRE(cast_out, UF_ERR_NULL, "uf_layernorm_bwd_cast: cast_out is NULL (use uf_layernorm_bwd_fused)");
    UF_REQUIRE(!dy_is_f32 || dtype == UF_F32, UF_ERR_UNSUPPORTED, "uf_layernorm_bwd_cast: dy must be of the operand type");
    UF_REQUIRE(!cast_windowed || (H % 8 == 0 && W % 8 == 0 && (cast_shift == 0 || cast_shift == 4)), UF_ERR_SHAPE, "uf_layernorm_bwd_cast: window geometry H=%d W=%d shift=%d", H, W,
               cast_shift);
    UF_REQUIRE(((uintptr_t)cast_out % 16) == 0 && C % 4 == 0, UF_ERR_ALIGN, "uf_layernorm_bwd_cast: cast_out alignment");
    return layernorm_bwd_any("uf_layernorm_bwd_cast", x, ld_x, gamma, dy, ld_dy, dy_is_f32, dtype, add, dx, ld_dx, dgamma, dbeta, B * H * W, C, windowed ? H : 0, windowed ? W : 0,
                             shift, ws, ws_bytes, stream, LnCastArgs{cast_out, cast_scale, cast_windowed ? 2 : 1, H * W, H, W, cast_shift});
}

extern "C" size_t uf_dwconv3x3_wgrad_workspace_bytes(int C, uf_dtype dtype) {
    const int N = dtype_half(dtype) ? 8 : 4;
    if (C <= 0 || C % N) return 0;
    const int cv = C / N;
    int blocks = dw_wgrad_blocks();
    while (((long long)blocks * 256) % cv) ++blocks;   // stride must be a multiple of the channel-group count
    return (size_t)blocks * 256 * 10 * N * sizeof(float);
}

extern "C" int uf_dwconv3x3_wgrad(const void* h, const void* dc, float* dw9, float* dbias, int B, int H, int W, int C, uf_dtype dtype,
                                  void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(h && dc && dw9 && dbias && ws, UF_ERR_NULL, "uf_dwconv3x3_wgrad: null pointer");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "uf_dwconv3x3_wgrad: dtype %d", (int)dtype);
    const int N = dtype_half(dtype) ? 8 : 4;
    UF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % N == 0 && H % DWB_R == 0, UF_ERR_SHAPE,
               "uf_dwconv3x3_wgrad: B=%d H=%d W=%d C=%d (C multiple of %d, H multiple of %d)", B, H, W, C, N, DWB_R);
    UF_REQUIRE((long long)B * H * W * C < 0x7fffffffLL, UF_ERR_SHAPE, "uf_dwconv3x3_wgrad: tensor too large");
    const size_t need = uf_dwconv3x3_wgrad_workspace_bytes(C, dtype);
    UF_REQUIRE(ws_bytes >= need, UF_ERR_WORKSPACE, "uf_dwconv3x3_wgrad: workspace too small: %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int cv = C / N;
    const int blocks = (int)(need / (256 * 10 * N * sizeof(float)));
    UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(dwconv3x3_wgrad_kernel<TT>, dim3(blocks), dim3(256), 0, st, (const TT*)h, (const TT*)dc, (float*)ws, B, H, W, C));
    int rc = check_launch("dwconv3x3_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(dwconv3x3_wgrad_finalize, dim3((10 * C + 31) / 32), dim3(256), 0, st, (const float*)ws, (long long)blocks * 256, cv, N, dw9, dbias, C);
    return check_launch("dwconv3x3_wgrad_finalize");
}

// geometry of the fused depthwise backward for C channels: 4 channels per thread (8-byte loads for the 2-byte types; with 8 the
// accumulators spill), channel groups per workgroup (a power of two dividing C / 4, at most 32), workgroups (1024 = four per CU,
// rounded up to whole channel-block rounds)
static void dw_bwd_geometry(int C, int* cg_log2, int* blocks) {
    const int cv = C / 4;
    int lg = 0;
    while (lg < 5 && cv % (2 << lg) == 0) ++lg;
    const int cb = cv >> lg, want = 1024;
    *cg_log2 = lg; *blocks = (want + cb - 1) / cb * cb;
}

extern "C" size_t uf_dwconv3x3_bwd_workspace_bytes(int C, uf_dtype dtype) {
    if (C <= 0 || !dtype_ok(dtype) || C % (dtype_half(dtype) ? 8 : 4)) return 0;
    int lg, blocks;
    dw_bwd_geometry(C, &lg, &blocks);
    return (size_t)blocks * 10 * (4 << lg) * sizeof(float);
}

extern "C" int uf_dwconv3x3_bwd(const void* dc, const float* w9_flipped, const void* pre, void* da, float* dw9, float* dbias, int B, int H, int W, int C,
                                uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(dc && w9_flipped && pre && da && dw9 && dbias && ws, UF_ERR_NULL, "uf_dwconv3x3_bwd: null pointer");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "uf_dwconv3x3_bwd: dtype %d", (int)dtype);
    const int Nn = dtype_half(dtype) ? 8 : 4;
    UF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % Nn == 0 && H % DWB_R == 0, UF_ERR_SHAPE,
               "uf_dwconv3x3_bwd: B=%d H=%d W=%d C=%d (C multiple of %d, H multiple of %d)", B, H, W, C, Nn, DWB_R);
    // The kernels address the tensor with 32-bit byte offsets.  A tensor of 4 GiB or more (f32 256 x 256 at batch 64 in the last decoder stage, 512 x 512
    // at f32 batch 16 / bf16 batch 32 -- shapes the two-kernel form handled) is walked in chunks of whole images, each under the limit; the tap / bias
    // gradients of the chunks are added in chunk order by the finalize kernel (the one-pass form used to fail with UF_ERR_SHAPE there).
    // UF_DWBWD_MAX_BYTES lowers the limit (tests).
    static const unsigned long long lim_env = getenv("UF_DWBWD_MAX_BYTES") ? strtoull(getenv("UF_DWBWD_MAX_BYTES"), nullptr, 10) : 0ULL;
    const unsigned long long limit = lim_env ? lim_env : 0xffffffffULL, per_img = (unsigned long long)H * W * C * dtype_size(dtype);
    UF_REQUIRE(per_img < limit, UF_ERR_SHAPE, "uf_dwconv3x3_bwd: one image of the tensor has %llu bytes (32-bit offsets: under %llu)", per_img, limit);
    const int chunkB = (int)((limit - 1) / per_img) < B ? (int)((limit - 1) / per_img) : B;
    UF_REQUIRE(((uintptr_t)dc % 16) == 0 && ((uintptr_t)pre % 16) == 0 && ((uintptr_t)da % 16) == 0 && ((uintptr_t)w9_flipped % 16) == 0, UF_ERR_ALIGN,
               "uf_dwconv3x3_bwd: operands must be 16-byte aligned");
    const size_t need = uf_dwconv3x3_bwd_workspace_bytes(C, dtype);
    UF_REQUIRE(ws_bytes >= need, UF_ERR_WORKSPACE, "uf_dwconv3x3_bwd: workspace too small: %zu < %zu", ws_bytes, need);
    int lg, blocks;
    dw_bwd_geometry(C, &lg, &blocks);
    const int cb = (C / 4) >> lg;
    hipStream_t st = (hipStream_t)stream;
    // the walking form for the 2-byte types (with f32 operands its 72 column registers on top of the erf-form GELU spill)
    const int seg = (dtype_half(dtype) && W % 8 == 0) ? (W % 16 == 0 ? 16 : 8) : 0;
    for (int b0 = 0; b0 < B; b0 += chunkB) {
        const int Bc = B - b0 < chunkB ? B - b0 : chunkB;
        const size_t off = (size_t)b0 * per_img;
        const char* dcc = (const char*)dc + off; const char* prec = (const char*)pre + off; char* dac = (char*)da + off;
#define UF_DWBWD_ARGS dim3(blocks), dim3(256), 0, st, (const TT*)dcc, w9_flipped, (const TT*)prec, (TT*)dac, (float*)ws, Bc, H, W, C, lg
        if (dtype == UF_F32) { using TT = float; hipLaunchKernelGGL((dwconv3x3_bwd_kernel<TT, 4, DWB_R>), UF_DWBWD_ARGS); }
        else if (dtype == UF_BF16) {
            using TT = bf16;
            if (seg == 16) hipLaunchKernelGGL((dwconv3x3_bwd_walk_kernel<TT, 16>), UF_DWBWD_ARGS);
            else if (seg == 8) hipLaunchKernelGGL((dwconv3x3_bwd_walk_kernel<TT, 8>), UF_DWBWD_ARGS);
            else hipLaunchKernelGGL((dwconv3x3_bwd_kernel<TT, 4, DWB_R>), UF_DWBWD_ARGS);
        } else {
            using TT = f16;
            if (seg == 16) hipLaunchKernelGGL((dwconv3x3_bwd_walk_kernel<TT, 16>), UF_DWBWD_ARGS);
            else if (seg == 8) hipLaunchKernelGGL((dwconv3x3_bwd_walk_kernel<TT, 8>), UF_DWBWD_ARGS);
            else hipLaunchKernelGGL((dwconv3x3_bwd_kernel<TT, 4, DWB_R>), UF_DWBWD_ARGS);
        }
#undef UF_DWBWD_ARGS
        int rc = check_launch("dwconv3x3_bwd");
        if (rc) return rc;
        hipLaunchKernelGGL(dwconv3x3_bwd_finalize, dim3((10 * C + 31) / 32), dim3(256), 0, st, (const float*)ws, blocks, cb, 4 << lg, dw9, dbias, C, b0 > 0 ? 1 : 0);
        rc = check_launch("dwconv3x3_bwd_finalize");
        if (rc) return rc;
    }
    return UF_OK;
}

static bool wgrad_v2() { return true; }   // 2-byte operand types: the 128 x 128-tile kernels (the first version serves f32)
static int wgrad_chunks(int M, int N, int K, uf_dtype dtype = UF_F32) {
    const int T = (dtype_half(dtype) && wgrad_v2()) ? 128 : 64;
    const int tiles = ((N + T - 1) / T) * ((K + T - 1) / T), steps = (M + 31) / 32;
    // workgroups per launch to aim for: every chunk costs one N x K f32 partial written and read again by the ordered sum, so no more
    // chunks than it takes to fill the chip (128-wide tiles: 2 workgroups per CU resident)
    const int target = T == 128 ? 512 : 2048;   // 512 vs 1024 vs 2048 measured: 148.8 / 151.0 / 151.1 ms per training step (round 2); 256 / 384 / 512 / 768: 64.0 / 63.3 / 63.0 / 63.7 (profiles/r06_run20_ab.txt)
    int S = target / tiles;
    if (S > steps) S = steps;
    if (S > 512) S = 512;       // one-tile shapes (C = 32: every weight; C = 64: the projection) had 256 workgroups for 512 slots under the old cap of 256
    // XCD x owns chunks x, x + 8, ... (the launch rounds the chunk count up to a multiple of 8 and the surplus workgroups exit): with S % 8 != 0 the first S % 8
    // XCDs carry one chunk = `tiles` workgroups more than the others and -- whenever that pushes them past their 64 resident workgroups -- run a second, nearly
    // empty round while six XCDs idle.  Round 6: the q|k|v gradient (N = 3C: 12 tiles x 42 chunks = 72 workgroups on XCDs 0 and 1, 60 on the rest) took 125 us
    // where linear1 (N = 4C: 16 x 32 = 64 everywhere) took 93 us for 4/3 of the work (profiles/r06_run12_wgrad.txt).  Whole multiples of 8 only.
    if (S >= 8) S = S / 8 * 8;
    return S < 1 ? 1 : S;
}

// fourth version (256 x 256 tiles, one workgroup of 8 waves per CU): N and K multiples of 256.  Measured against the third version on the
// Uformer-B shapes at batch 32 (profiles/r04_run12.txt, microseconds third / fourth): 131072 x 1024 x 256: 105 / 90, 131072 x 768 x 256: 125 / 121,
// 131072 x 256 x 256: 35 / 45, 32768 x 2048 x 512: 82 / 83, 32768 x 1024 x 256: 34 / 41, 8192 x 2048 x 512: 33 / 43 -- it pays with many tokens per
// chunk (its pipeline is three stages deep before the first MFMA, and a chunk's partial tile is four times as large), so that is where it runs.
// UF_VARIANT="wgrad4=0" turns it off, "wgrad4=1" runs it on every shape it supports (A/B runs, tests).
static bool wgrad4_shape(int M, int N, int K) {
    const int e = variant("wgrad4", -1);
    if (e == 0 || !wgrad_v2() || N % 256 || K % 256 || M < 256) return false;
    if (e == 1) return true;
    return M >= 65536 && (long long)N * K >= 196608;
}
static int wgrad4_chunks(int M, int N, int K) {
    const int tiles = (N / 256) * (K / 256), steps = (M + WG4_TOK - 1) / WG4_TOK;
    int S = 256 / tiles;     // one workgroup per CU (128 KiB of LDS each)
    if (S > steps / 4) S = steps / 4;                         // at least four stages per chunk
    if (S > 256) S = 256;
    if (S >= 8) S = S / 8 * 8;                                // the same number of chunks on every XCD (see wgrad_chunks): 3 tiles x 85 chunks put 33 workgroups on five XCDs of 32 CUs
    return S < 1 ? 1 : S;
}

extern "C" size_t uf_linear_wgrad_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int s = wgrad_chunks(M, N, K, UF_F32);                                                   // the dtype is not an argument: size for any kernel
    const int s2 = wgrad_chunks(M, N, K, UF_BF16), s4 = N % 256 == 0 && K % 256 == 0 ? wgrad4_chunks(M, N, K) : 0;
    s = s2 > s ? s2 : s;
    s = s4 > s ? s4 : s;
    return (size_t)s * ((size_t)N * K + N) * sizeof(float);
}

extern "C" int uf_linear_wgrad(const void* dY, int ldy, const void* X, int ldx, float* dW, float* db, int M, int N, int K,
                               uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(dY && X && dW && ws, UF_ERR_NULL, "uf_linear_wgrad: null pointer");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "uf_linear_wgrad: dtype %d", (int)dtype);
    const int EP = dtype_half(dtype) ? 8 : 4;
    UF_REQUIRE(M > 0 && N >= EP && K >= EP && N % EP == 0 && K % EP == 0 && ldy >= N && ldx >= K && ldy % EP == 0 && ldx % EP == 0, UF_ERR_SHAPE,
               "uf_linear_wgrad: M=%d N=%d K=%d ld=(%d,%d) (N, K, ld multiples of %d)", M, N, K, ldy, ldx, EP);
    UF_REQUIRE(((uintptr_t)dY % 16) == 0 && ((uintptr_t)X % 16) == 0, UF_ERR_ALIGN, "uf_linear_wgrad: operands must be 16-byte aligned");
    const size_t need = uf_linear_wgrad_workspace_bytes(M, N, K);
    UF_REQUIRE(ws_bytes >= need, UF_ERR_WORKSPACE, "uf_linear_wgrad: workspace too small: %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const bool v2 = dtype_half(dtype) && wgrad_v2();
    const bool v4 = v2 && wgrad4_shape(M, N, K) && ldy % 8 == 0 && ldx % 8 == 0 && ((long long)M + 64) * ldy * 2 < 0xffffff00LL && ((long long)M + 64) * ldx * 2 < 0xffffff00LL;   // + 64: the rows of the last stage past M
    const int S = v4 ? wgrad4_chunks(M, N, K) : wgrad_chunks(M, N, K, dtype);
    float* ws_w = (float*)ws;
    float* ws_b = ws_w + (size_t)S * N * K;
    const int TT = v2 ? 128 : 64;
    const dim3 grid(((N + TT - 1) / TT) * ((K + TT - 1) / TT), S);
    char name[96] = "";
    if (timing_enabled()) snprintf(name, sizeof(name), "linear_wgrad_%s %dx%dx%d", dtype_name(dtype), M, N, K);
    {
        ScopedTimer tm(name, 2.0 * M * N * K, (double)M * (N + K) * dtype_size(dtype) + 4.0 * N * K, st);
        constexpr int xcd_map = 1;          // XCD-aware chunk order (profiles/r03_wgrad_xcd.txt)
        const dim3 grid2((unsigned)(grid.x * (xcd_map ? (S + 7) / 8 * 8 : S)));
        // third version (LDS-DMA staging, 64-token steps) where both operands are addressable with 32-bit byte offsets, else the second version
        const bool v3 = v2 && M >= 256 && ((long long)M + 64) * ldy * 2 < 0xffffff00LL && ((long long)M + 64) * ldx * 2 < 0xffffff00LL;
        const dim3 grid3((unsigned)(grid.x * ((S + 7) / 8 * 8)));
        const dim3 grid4((unsigned)((N / 256) * (K / 256) * ((S + 7) / 8 * 8)));
        if (v4 && dtype == UF_BF16) hipLaunchKernelGGL(linear_wgrad4_kernel<bf16>, grid4, dim3(512), 0, st, (const bf16*)dY, ldy, (const bf16*)X, ldx, ws_w, ws_b, M, N, K, S);
        else if (v4) hipLaunchKernelGGL(linear_wgrad4_kernel<f16>, grid4, dim3(512), 0, st, (const f16*)dY, ldy, (const f16*)X, ldx, ws_w, ws_b, M, N, K, S);
        else if (v3 && dtype == UF_BF16) hipLaunchKernelGGL(linear_wgrad3_kernel<bf16>, grid3, dim3(256), 0, st, (const bf16*)dY, ldy, (const bf16*)X, ldx, ws_w, ws_b, M, N, K, S);
        else if (v3) hipLaunchKernelGGL(linear_wgrad3_kernel<f16>, grid3, dim3(256), 0, st, (const f16*)dY, ldy, (const f16*)X, ldx, ws_w, ws_b, M, N, K, S);
        else if (v2 && dtype == UF_BF16) hipLaunchKernelGGL(linear_wgrad2_kernel<bf16>, grid2, dim3(256), 0, st, (const bf16*)dY, ldy, (const bf16*)X, ldx, ws_w, ws_b, M, N, K, S, xcd_map);
        else if (v2) hipLaunchKernelGGL(linear_wgrad2_kernel<f16>, grid2, dim3(256), 0, st, (const f16*)dY, ldy, (const f16*)X, ldx, ws_w, ws_b, M, N, K, S, xcd_map);
        else UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(linear_wgrad_kernel<TT>, grid, dim3(256), 0, st, (const TT*)dY, ldy, (const TT*)X, ldx, ws_w, ws_b, M, N, K));
    }
    int rc = check_launch("linear_wgrad");
    if (rc) return rc;
    const int nk = N * K;
    if (db) launch_column_sum2(st, ws_w, S, (size_t)N * K, dW, nk, ws_b, S, (size_t)N, db, N);
    else launch_column_sum2(st, ws_w, S, (size_t)N * K, dW, nk, nullptr, 0, 0, nullptr, 0);
    return check_launch("linear_wgrad_finalize");
}

static int attn_bwd_chunks(int n_windows, int heads) {
    int G = 1024 / heads;
    if (G > n_windows) G = n_windows;
    return G < 1 ? 1 : G;
}

extern "C" size_t uf_window_attention_bwd_workspace_bytes(int n_windows, int heads) {
    if (n_windows <= 0 || heads <= 0) return 0;
    return (size_t)attn_bwd_chunks(n_windows, heads) * heads * 4096 * sizeof(float);
}

static int window_attention_bwd_any(const void* q, const void* k, const void* vt, const float* bias_dense, const float* mask, int n_mask,
                                    const void* dO, int ldo, void* dq, void* dk, void* dvt, void* dqkv, float* dbias, int n_windows, int heads,
                                    int head_dim, int H, int W, int shift, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(q && k && vt && bias_dense && dO && ((dq && dk && dvt) || dqkv) && dbias && ws, UF_ERR_NULL, "uf_window_attention_bwd: null pointer");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "uf_window_attention_bwd: dtype %d", (int)dtype);
    UF_REQUIRE(n_windows > 0 && heads > 0, UF_ERR_SHAPE, "uf_window_attention_bwd: n_windows=%d heads=%d", n_windows, heads);
    UF_REQUIRE(head_dim == 32 || head_dim == 16, UF_ERR_UNSUPPORTED, "uf_window_attention_bwd: head_dim %d (16 or 32)", head_dim);
    UF_REQUIRE(H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8 && n_windows % ((H / 8) * (W / 8)) == 0, UF_ERR_SHAPE,
               "uf_window_attention_bwd: H=%d W=%d n_windows=%d", H, W, n_windows);
    UF_REQUIRE(shift == 0 || shift == 4, UF_ERR_UNSUPPORTED, "uf_window_attention_bwd: shift %d (0 or 4)", shift);
    UF_REQUIRE(!mask || n_mask > 0, UF_ERR_SHAPE, "uf_window_attention_bwd: mask given with n_mask=%d", n_mask);
    const int EP = dtype_half(dtype) ? 8 : 4;
    UF_REQUIRE(ldo >= heads * head_dim && ldo % EP == 0, UF_ERR_ALIGN, "uf_window_attention_bwd: ldo=%d", ldo);
    const size_t need = uf_window_attention_bwd_workspace_bytes(n_windows, heads);
    UF_REQUIRE(ws_bytes >= need, UF_ERR_WORKSPACE, "uf_window_attention_bwd: workspace too small: %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int G = attn_bwd_chunks(n_windows, heads);
    const float qscale = (float)(1.0 / sqrt((double)head_dim));   // python: head_dim ** -0.5, rounded once to f32
    const int SZ = (int)dtype_size(dtype);
    const int smem = 4 * 64 * (32 * SZ + 16) + 3 * 32 * (64 * SZ + 16) + 4 * 64 * (64 * SZ + 16);
    char name[96] = "";
    if (timing_enabled()) snprintf(name, sizeof(name), "window_attn_bwd_%s %dx%d", dtype_name(dtype), n_windows, heads);
    {
        const double pairs = (double)n_windows * heads;
        ScopedTimer tm(name, 2.0 * 5 * 64 * 64 * 32 * pairs, pairs * 64 * 32 * 7.0 * SZ, st);
#define UF_ATTN_BWD(TT, HDV)                                                                                                                        \
        {                                                                                                                                              \
            static bool done[64] = {};                                                                                                                \
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(window_attn_bwd_kernel<TT, HDV>), smem, done, "window_attn_bwd")) return rc;  \
            hipLaunchKernelGGL((window_attn_bwd_kernel<TT, HDV>), dim3(heads, G), dim3(256), smem, st, (const TT*)q, (const TT*)k, (const TT*)vt,       \
                               bias_dense, mask, n_mask, (const TT*)dO, ldo, (TT*)dq, (TT*)dk, (TT*)dvt, (TT*)dqkv, qscale, (float*)ws, n_windows,    \
                               heads, H, W, shift);                                                                                                   \
        }
        // second version (accumulators chained into the operands, nothing stored transposed) for the 2-byte types (the first version serves f32)
        const bool v2 = dtype_half(dtype) && ((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)vt % 16) == 0 && ((uintptr_t)dO % 16) == 0;
#define UF_ATTN_BWD2(TT, HDV)                                                                                                                       \
        if (mask) hipLaunchKernelGGL((window_attn_bwd2_kernel<TT, HDV, true>), dim3(heads, G), dim3(256), 0, st, (const TT*)q, (const TT*)k, (const TT*)vt, bias_dense, mask,      \
                           n_mask, (const TT*)dO, ldo, (TT*)dq, (TT*)dk, (TT*)dvt, (TT*)dqkv, qscale, (float*)ws, n_windows, heads, H, W, shift, pair);                      \
        else hipLaunchKernelGGL((window_attn_bwd2_kernel<TT, HDV, false>), dim3(heads, G), dim3(256), 0, st, (const TT*)q, (const TT*)k, (const TT*)vt, bias_dense, mask,      \
                           n_mask, (const TT*)dO, ldo, (TT*)dq, (TT*)dk, (TT*)dvt, (TT*)dqkv, qscale, (float*)ws, n_windows, heads, H, W, shift, pair);
        const int pair = variant("attpair", 1);      // UF_VARIANT="attpair=0": (head, chunk) in launch order (A/B runs); same bits either way
        if (v2 && dtype == UF_BF16) { if (head_dim == 32) { UF_ATTN_BWD2(bf16, 32) } else { UF_ATTN_BWD2(bf16, 16) } }
        else if (v2) { if (head_dim == 32) { UF_ATTN_BWD2(f16, 32) } else { UF_ATTN_BWD2(f16, 16) } }
        else
        UF_DISPATCH(dtype, TT, { if (head_dim == 32) UF_ATTN_BWD(TT, 32) else UF_ATTN_BWD(TT, 16) });
#undef UF_ATTN_BWD2
#undef UF_ATTN_BWD
    }
    int rc = check_launch("window_attn_bwd");
    if (rc) return rc;
    const int nb = heads * 4096;
    hipLaunchKernelGGL(column_sum_kernel, dim3((nb + COLSUM_COLS - 1) / COLSUM_COLS), dim3(256), 0, st, (const float*)ws, G, (size_t)nb, dbias, nb);
    return check_launch("window_attn_bwd_finalize");
}

extern "C" int uf_window_attention_bwd(const void* q, const void* k, const void* vt, const float* bias_dense, const float* mask, int n_mask,
                                       const void* dO, int ldo, void* dq, void* dk, void* dvt, float* dbias, int n_windows, int heads,
                                       int head_dim, int H, int W, int shift, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(dq && dk && dvt, UF_ERR_NULL, "uf_window_attention_bwd: null output");
    return window_attention_bwd_any(q, k, vt, bias_dense, mask, n_mask, dO, ldo, dq, dk, dvt, nullptr, dbias, n_windows, heads, head_dim, H, W, shift, dtype, ws, ws_bytes, stream);
}

// the same backward writing ONE tensor: dqkv T[n_windows*64][3C], the gradient of the fused q|k|v projection output (heads merged,
// dq times head_dim^-0.5) -- what uf_linear_wgrad / the input-gradient GEMM of the projection read next.
extern "C" int uf_window_attention_bwd_qkv(const void* q, const void* k, const void* vt, const float* bias_dense, const float* mask, int n_mask,
                                           const void* dO, int ldo, void* dqkv, float* dbias, int n_windows, int heads, int head_dim, int H, int W,
                                           int shift, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(dqkv && ((uintptr_t)dqkv % 16) == 0, UF_ERR_NULL, "uf_window_attention_bwd_qkv: dqkv must be a 16-byte aligned pointer");
    return window_attention_bwd_any(q, k, vt, bias_dense, mask, n_mask, dO, ldo, nullptr, nullptr, nullptr, dqkv, dbias, n_windows, heads, head_dim, H, W, shift, dtype, ws, ws_bytes, stream);
}

// ===============================================================================================================================
// Reductions / patch gathers of the training path that used to be ATen glue: the relative-position
// bias-table gradient (index_add_), the modulator gradient (sum over windows) and the im2col / col2im of the strided
// convolutions (torch unfold / fold).  All deterministic.
// ===============================================================================================================================
namespace uf {
namespace {

// partial[p][n] = sum over rows m = p, p + P, ... of X[m][n]  (then column_sum_kernel over p)
template <typename T>
__global__ __launch_bounds__(256) void rows_partial_kernel(const T* __restrict__ X, int ld, float* __restrict__ partial, int M, int N, int P) {
    constexpr int V = Vec<T>::N;
    const int nv = N / V;
    const int j = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
    if (j >= nv) return;
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    for (int m = p; m < M; m += P) {
        float f[V];
        Vec<T>::load(X + (size_t)m * ld + (size_t)j * V, f);
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] += f[i];
    }
#pragma unroll
    for (int i = 0; i < V; ++i) partial[(size_t)p * N + (size_t)j * V + i] = acc[i];
}

// dtable[e][h] = sum of dbias[h][i][j] over the (i, j) with relative_position_index[i][j] == e, e = (dy+7)*15 + (dx+7)
// (model.py:467-477, :500-502): for a fixed (dy, dx) the pairs are yi in [max(0,dy), min(8,8+dy)), xi likewise -- walked in
// row-major order, one thread per table entry and head.
__global__ void rpb_table_grad_kernel(const float* __restrict__ dbias, float* __restrict__ dtable, int heads) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 225 * heads) return;
    const int e = t / heads, h = t - e * heads;
    const int dy = e / 15 - 7, dx = e % 15 - 7;
    float s = 0.f;
    for (int yi = (dy > 0 ? dy : 0); yi < (dy < 0 ? 8 + dy : 8); ++yi)
        for (int xi = (dx > 0 ? dx : 0); xi < (dx < 0 ? 8 + dx : 8); ++xi) {
            const int i = yi * 8 + xi, j = (yi - dy) * 8 + (xi - dx);
            s += dbias[((size_t)h * 64 + i) * 64 + j];
        }
    dtable[(size_t)e * heads + h] = s;
}

// cols[m][(ky*k + kx)*Cin + c] = x[b][oy*s + ky - pad][ox*s + kx - pad][c] (0 outside the map), m = (b*Ho + oy)*Wo + ox; columns
// [k*k*Cin, ldc) are zero.  x: f32 token rows (stride ld_x) or NCHW planes.
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, int ld_x, T* __restrict__ cols, int ldc, int B, int H, int W, int Cin, int k,
                                                     int stride, int pad, int Ho, int Wo, int nchw) {
    const long long total = (long long)B * Ho * Wo * ldc;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int col = (int)(t % ldc);
        const long long m = t / ldc;
        float v = 0.f;
        if (col < k * k * Cin) {
            const int c = col % Cin, tap = col / Cin, ky = tap / k, kx = tap - ky * k;
            const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long long)Wo * Ho));
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                v = nchw ? x[(((size_t)b * Cin + c) * H + iy) * W + ix] : x[((size_t)(b * H + iy) * W + ix) * ld_x + c];
        }
        store1(cols + t, v);
    }
}

// dx[b][y][x][c] (+)= sum over the taps (ky, kx) whose output pixel oy = (y + pad - ky) / stride, ox = ... is integral and inside:
// dcols[(b*Ho + oy)*Wo + ox][(ky*k + kx)*Cin + c].  Gather form: one thread per input element, taps in (ky, kx) order.
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcols, int ldc, float* __restrict__ dx, int ld_dx, int B, int H, int W, int Cin, int k,
                                                     int stride, int pad, int Ho, int Wo, int nchw, int accumulate) {
    const long long total = (long long)B * H * W * Cin;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        int c, xx, yy, b;
        if (nchw) { xx = (int)(t % W); yy = (int)((t / W) % H); c = (int)((t / ((long long)W * H)) % Cin); b = (int)(t / ((long long)W * H * Cin)); }
        else { c = (int)(t % Cin); xx = (int)((t / Cin) % W); yy = (int)((t / ((long long)Cin * W)) % H); b = (int)(t / ((long long)Cin * W * H)); }
        float s = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int ny = yy + pad - ky;
            if (ny < 0 || ny % stride) continue;
            const int oy = ny / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int nx = xx + pad - kx;
                if (nx < 0 || nx % stride) continue;
                const int ox = nx / stride;
                if (ox >= Wo) continue;
                s += load1(dcols + ((size_t)(b * Ho + oy) * Wo + ox) * ldc + (ky * k + kx) * Cin + c);
            }
        }
        float* o = nchw ? dx + t : dx + ((size_t)(b * H + yy) * W + xx) * ld_dx + c;
        *o = accumulate ? *o + s : s;
    }
}

// Token-layout (NHWC) fast paths: 8 channels per thread (Cin % 8 == 0 keeps the 8 inside one tap), 16-byte loads and stores.
template <typename T> __device__ __forceinline__ void ld8(const T* p, float* f) { Vec<T>::load(p, f); if constexpr (sizeof(T) == 4) Vec<T>::load(p + 4, f + 4); }
template <typename T> __device__ __forceinline__ void st8(T* p, const float* f) { Vec<T>::store(p, f); if constexpr (sizeof(T) == 4) Vec<T>::store(p + 4, f + 4); }

template <typename T>
__global__ __launch_bounds__(256) void im2col_rows8_kernel(const float* __restrict__ x, int ld_x, T* __restrict__ cols, int ldc, int B, int H, int W, int Cin, int k,
                                                           int stride, int pad, int Ho, int Wo) {
    const int cv = ldc / 8, kk = k * k * Cin;
    const long long total = (long long)B * Ho * Wo * cv;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int col = (int)(t % cv) * 8;
        const long long m = t / cv;
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (col < kk) {
            const int tap = col / Cin, c = col - tap * Cin, ky = tap / k, kx = tap - ky * k;
            const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long long)Wo * Ho));
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) ld8<float>(x + ((size_t)(b * H + iy) * W + ix) * ld_x + c, f);
        }
        st8<T>(cols + (size_t)m * ldc + col, f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void col2im_rows8_kernel(const T* __restrict__ dcols, int ldc, float* __restrict__ dx, int ld_dx, int B, int H, int W, int Cin, int k,
                                                           int stride, int pad, int Ho, int Wo, int accumulate) {
    const int cv = Cin / 8;
    const long long total = (long long)B * H * W * cv;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int c = (int)(t % cv) * 8, xx = (int)((t / cv) % W), yy = (int)((t / ((long long)cv * W)) % H), b = (int)(t / ((long long)cv * W * H));
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < k; ++ky) {                                            // same (ky, kx) order as the scalar kernel: bit-identical sums
            const int ny = yy + pad - ky;
            if (ny < 0 || ny % stride) continue;
            const int oy = ny / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int nx = xx + pad - kx;
                if (nx < 0 || nx % stride) continue;
                const int ox = nx / stride;
                if (ox >= Wo) continue;
                float f[8];
                ld8<T>(dcols + ((size_t)(b * Ho + oy) * Wo + ox) * ldc + (ky * k + kx) * Cin + c, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += f[e];
            }
        }
        float* o = dx + ((size_t)(b * H + yy) * W + xx) * ld_dx + c;
        if (accumulate) {
            float a[8];
            ld8<float>(o, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += a[e];
        }
        st8<float>(o, s);
    }
}

int grid1d(long long n) {
    long long g = (n + 1023) / 1024;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace
}  // namespace uf

static int rows_sum_parts(int M) { return M < 128 ? M : 128; }
extern "C" size_t uf_rows_sum_workspace_bytes(int M, int N) { return M <= 0 || N <= 0 ? 0 : (size_t)rows_sum_parts(M) * N * sizeof(float); }
extern "C" int uf_rows_sum(const void* X, int ld, float* out, int M, int N, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(X && out && ws, UF_ERR_NULL, "uf_rows_sum: null pointer");
    const int V = dtype_half(dtype) ? 8 : 4;
    UF_REQUIRE(M > 0 && N > 0 && N % V == 0 && ld >= N && ld % V == 0 && ((uintptr_t)X % 16) == 0, UF_ERR_SHAPE, "uf_rows_sum: M=%d N=%d ld=%d (N, ld multiples of %d)", M, N, ld, V);
    UF_REQUIRE(ws_bytes >= uf_rows_sum_workspace_bytes(M, N), UF_ERR_WORKSPACE, "uf_rows_sum: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int P = rows_sum_parts(M);
    const dim3 grid((N / V + 255) / 256, P);
    UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(rows_partial_kernel<TT>, grid, dim3(256), 0, st, (const TT*)X, ld, (float*)ws, M, N, P));
    hipLaunchKernelGGL(column_sum_kernel, dim3((N + COLSUM_COLS - 1) / COLSUM_COLS), dim3(256), 0, st, (const float*)ws, P, (size_t)N, out, N);
    return check_launch("rows_sum");
}

extern "C" int uf_rpb_table_grad(const float* dbias_dense, float* dtable, int heads, void* stream) {
    UF_REQUIRE(dbias_dense && dtable && heads > 0, UF_ERR_NULL, "uf_rpb_table_grad: null pointer / heads");
    hipLaunchKernelGGL(rpb_table_grad_kernel, dim3((225 * heads + 255) / 256), dim3(256), 0, (hipStream_t)stream, dbias_dense, dtable, heads);
    return check_launch("rpb_table_grad");
}

extern "C" int uf_im2col(const float* x, int ld_x, void* cols, int ldc, int B, int H, int W, int Cin, int k, int stride, int pad, int nchw, uf_dtype dtype, void* stream) {
    UF_REQUIRE(x && cols, UF_ERR_NULL, "uf_im2col: null pointer");
    UF_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && k > 0 && stride > 0 && pad >= 0 && ldc >= k * k * Cin && (nchw || ld_x >= Cin), UF_ERR_SHAPE, "uf_im2col: shape");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long long total = (long long)B * Ho * Wo * ldc;
    hipStream_t st = (hipStream_t)stream;
    if (!nchw && Cin % 8 == 0 && ldc % 8 == 0 && ld_x % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)cols % 16) == 0) {
        UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(im2col_rows8_kernel<TT>, dim3(grid1d(total / 8)), dim3(256), 0, st, x, ld_x, (TT*)cols, ldc, B, H, W, Cin, k, stride, pad, Ho, Wo));
        return check_launch("im2col");
    }
    UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(im2col_kernel<TT>, dim3(grid1d(total)), dim3(256), 0, st, x, ld_x, (TT*)cols, ldc, B, H, W, Cin, k, stride, pad, Ho, Wo, nchw));
    return check_launch("im2col");
}

extern "C" int uf_col2im(const void* dcols, int ldc, float* dx, int ld_dx, int B, int H, int W, int Cin, int k, int stride, int pad, int nchw, int accumulate,
                         uf_dtype dtype, void* stream) {
    UF_REQUIRE(dcols && dx, UF_ERR_NULL, "uf_col2im: null pointer");
    UF_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && k > 0 && stride > 0 && pad >= 0 && ldc >= k * k * Cin && (nchw || ld_dx >= Cin), UF_ERR_SHAPE, "uf_col2im: shape");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long long total = (long long)B * H * W * Cin;
    hipStream_t st = (hipStream_t)stream;
    if (!nchw && Cin % 8 == 0 && ldc % 8 == 0 && ld_dx % 4 == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)dcols % 16) == 0) {
        UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(col2im_rows8_kernel<TT>, dim3(grid1d(total / 8)), dim3(256), 0, st, (const TT*)dcols, ldc, dx, ld_dx, B, H, W, Cin, k, stride, pad, Ho, Wo, accumulate));
        return check_launch("col2im");
    }
    UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(col2im_kernel<TT>, dim3(grid1d(total)), dim3(256), 0, st, (const TT*)dcols, ldc, dx, ld_dx, B, H, W, Cin, k, stride, pad, Ho, Wo, nchw, accumulate));
    return check_launch("col2im");
}

// ===============================================================================================================================
// Streaming helpers of the block's recompute / backward (every one replaces 2-5 ATen elementwise passes: cast, per-sample DropPath
// scale, window_partition / window_reverse, residual add, head merge).  8 channels per thread, rows of C channels (C % 8 == 0).
// ===============================================================================================================================
namespace uf {
namespace {

template <typename T> __device__ __forceinline__ void load8(const T* p, float* f) { Vec<T>::load(p, f); }   // 2-byte types
template <> __device__ __forceinline__ void load8<float>(const float* p, float* f) { Vec<float>::load(p, f); Vec<float>::load(p + 4, f + 4); }
template <typename T> __device__ __forceinline__ void store8(T* p, const float* f) { Vec<T>::store(p, f); }   // 2-byte types
template <> __device__ __forceinline__ void store8<float>(float* p, const float* f) { Vec<float>::store(p, f); Vec<float>::store(p + 4, f + 4); }

// out[tok] = (a ? a[tok] : 0) + s(tok) * b[row],  row = the window-order index of tok when `windowed` (b is in window order: this
// is window_reverse + roll back, model.py:975-980), else tok;  s = scale[image of tok] or 1.
template <typename TB>
__global__ __launch_bounds__(256) void residual_combine_kernel(const float* a /* may alias out (in place) */, const TB* __restrict__ b, float* out, const float* __restrict__ scale,
                                                               int B, int H, int W, int C, int windowed, int shift) {
    const int cv = C / 8;
    const long long total = (long long)B * H * W * cv;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int c8 = (int)(t % cv) * 8;
        const int row = (int)(t / cv);                                             // index into b
        const int tok = windowed ? window_row_to_token(row, H, W, shift) : row;  // index into a / out
        const float s = scale ? scale[tok / (H * W)] : 1.0f;
        float fb[8], fa[8];
        load8<TB>(b + (size_t)row * C + c8, fb);
        if (a) load8<float>(a + (size_t)tok * C + c8, fa);
#pragma unroll
        for (int e = 0; e < 8; ++e) fb[e] = (a ? fa[e] : 0.f) + s * fb[e];
        store8<float>(out + (size_t)tok * C + c8, fb);
    }
}

// t = g1[tok] (+ g2[tok]);  sum_out[tok] = t (optional);  cast_out[row] = T(t * s(tok)),  row = window-order index of tok when
// `windowed` (roll + window_partition, model.py:957-963), else tok.
template <typename TO>
__global__ __launch_bounds__(256) void grad_fork_kernel(const float* g1 /* may alias sum_out (in place) */, const float* __restrict__ g2, float* sum_out, TO* __restrict__ cast_out,
                                                        const float* __restrict__ scale, int B, int H, int W, int C, int windowed, int shift) {
    const int cv = C / 8;
    const long long total = (long long)B * H * W * cv;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int c8 = (int)(t % cv) * 8;
        const int row = (int)(t / cv);
        const int tok = windowed ? window_row_to_token(row, H, W, shift) : row;
        const float s = scale ? scale[tok / (H * W)] : 1.0f;
        float f[8], f2[8];
        load8<float>(g1 + (size_t)tok * C + c8, f);
        if (g2) {
            load8<float>(g2 + (size_t)tok * C + c8, f2);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += f2[e];
        }
        if (sum_out) store8<float>(sum_out + (size_t)tok * C + c8, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= s;
        store8<TO>(cast_out + (size_t)row * C + c8, f);
    }
}

// dqkv[m][0..3C) in window-row order from the attention backward's per-(window, head) tensors:
//   [0,C)   = dq[w][h][t][d] * qscale      (dq is the gradient wrt the SCALED query: model.py:497)
//   [C,2C)  = dk[w][h][t][d]               [2C,3C) = dvt[w][h][d][t]      with m = w*64 + t, channel = h*32 + d
template <typename T>
__global__ __launch_bounds__(256) void qkv_grad_merge_kernel(const T* __restrict__ dq, const T* __restrict__ dk, const T* __restrict__ dvt, T* __restrict__ out,
                                                             int n_windows, int heads, float qscale) {
    const int C = heads * 32;
    const long long total = (long long)n_windows * 64 * heads * 12;      // 8-channel pieces: 3 parts x heads x 4 per row
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int piece = (int)(t % (heads * 12));
        const long long m = t / (heads * 12);
        const int part = piece / (heads * 4), hp = piece - part * heads * 4, h = hp >> 2, d0 = (hp & 3) * 8;
        const int w = (int)(m >> 6), tk = (int)(m & 63);
        float f[8];
        if (part < 2) {
            load8<T>((part == 0 ? dq : dk) + (((size_t)w * heads + h) * 64 + tk) * 32 + d0, f);
            if (part == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] *= qscale;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = load1(dvt + (((size_t)w * heads + h) * 32 + d0 + e) * 64 + tk);
        }
        store8<T>(out + (size_t)m * 3 * C + part * C + h * 32 + d0, f);
    }
}

}  // namespace
}  // namespace uf

extern "C" int uf_residual_combine(const float* a, const void* b, int b_is_f32, float* out, const float* scale, int B, int H, int W, int C, int windowed, int shift,
                                   uf_dtype dtype, void* stream) {
    UF_REQUIRE(b && out, UF_ERR_NULL, "uf_residual_combine: null pointer");
    UF_REQUIRE(B > 0 && H > 0 && W > 0 && C % 8 == 0 && (!windowed || (H % 8 == 0 && W % 8 == 0)), UF_ERR_SHAPE, "uf_residual_combine: B=%d H=%d W=%d C=%d", B, H, W, C);
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d((long long)B * H * W * (C / 8));
    UF_DISPATCH(b_is_f32 ? UF_F32 : dtype, TT, hipLaunchKernelGGL(residual_combine_kernel<TT>, dim3(grid), dim3(256), 0, st, a, (const TT*)b, out, scale, B, H, W, C, windowed, shift));
    return check_launch("residual_combine");
}

extern "C" int uf_grad_fork(const float* g1, const float* g2, float* sum_out, void* cast_out, const float* scale, int B, int H, int W, int C, int windowed, int shift,
                            uf_dtype dtype, void* stream) {
    UF_REQUIRE(g1 && cast_out, UF_ERR_NULL, "uf_grad_fork: null pointer");
    UF_REQUIRE(B > 0 && H > 0 && W > 0 && C % 8 == 0 && (!windowed || (H % 8 == 0 && W % 8 == 0)), UF_ERR_SHAPE, "uf_grad_fork: B=%d H=%d W=%d C=%d", B, H, W, C);
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d((long long)B * H * W * (C / 8));
    UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(grad_fork_kernel<TT>, dim3(grid), dim3(256), 0, st, g1, g2, sum_out, (TT*)cast_out, scale, B, H, W, C, windowed, shift));
    return check_launch("grad_fork");
}

extern "C" int uf_qkv_grad_merge(const void* dq, const void* dk, const void* dvt, void* dqkv, int n_windows, int heads, int head_dim, uf_dtype dtype, void* stream) {
    UF_REQUIRE(dq && dk && dvt && dqkv, UF_ERR_NULL, "uf_qkv_grad_merge: null pointer");
    UF_REQUIRE(n_windows > 0 && heads > 0 && head_dim == 32, UF_ERR_SHAPE, "uf_qkv_grad_merge: n_windows=%d heads=%d head_dim=%d (32)", n_windows, heads, head_dim);
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d((long long)n_windows * 64 * heads * 12);
    const float qs = (float)(1.0 / sqrt((double)head_dim));   // python: head_dim ** -0.5, rounded once to f32
    UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(qkv_grad_merge_kernel<TT>, dim3(grid), dim3(256), 0, st, (const TT*)dq, (const TT*)dk, (const TT*)dvt, (TT*)dqkv, n_windows, heads, qs));
    return check_launch("qkv_grad_merge");
}

// ===============================================================================================================================
// Backward of the two 3x3 stride-1 pad-1 convolutions at full resolution: InputProj (3 -> E, + LeakyReLU) and OutputProj (2E -> 3)
// (model.py:771-800, :803-827).  One side has <= 4 channels, so these are streaming kernels, not GEMMs: the patch-matrix route
// (uf_im2col + GEMMs + uf_col2im) moved ~10 GB per step for 0.7 % of the FLOPs.  All f32.
//   dyeff[p][co] = dy[p][co] * (act ? (act[p][co] > 0 ? 1 : slope) : 1)           (LeakyReLU' from the stored OUTPUT of the stem)
//   dx[q][ci]        = sum_{ky,kx,co} dyeff[q - (ky-1, kx-1)][co] * w[co][ci][ky][kx]
//   dW[co][ci][ky][kx] = sum_p dyeff[p][co] * x[p + (ky-1, kx-1)][ci]             db[co] = sum_p dyeff[p][co]
// ===============================================================================================================================
namespace uf {
namespace {

constexpr int CB_MAXW = 2048;        // Cout * Cin * 9 of the two layers: 864 and 1728

__device__ __forceinline__ float leaky_mask(const float* act, size_t i, float slope) { return act ? (act[i] > 0.f ? 1.f : slope) : 1.f; }   // torch: x > 0 ? g : g * slope; sign(out) = sign(x)

// thread = (pixel q, group of 4 input channels).  Weights in LDS as wl[tap][co][ci].
__global__ __launch_bounds__(256) void conv3x3_dx_kernel(const float* __restrict__ dy, const float* __restrict__ act, float slope, const float* __restrict__ w,
                                                         float* __restrict__ dx, int B, int H, int W, int Cin, int Cout, int nchw) {
    __shared__ float wl[CB_MAXW];
    for (int i = threadIdx.x; i < Cout * Cin * 9; i += 256) {          // w[co][ci][tap] -> wl[tap][co][ci]
        const int tap = i % 9, ci = (i / 9) % Cin, co = i / (9 * Cin);
        wl[(tap * Cout + co) * Cin + ci] = w[i];
    }
    __syncthreads();
    const int groups = (Cin + 3) / 4;
    const long long total = (long long)B * H * W * groups;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int grp = (int)(t % groups);
        const long long q = t / groups;
        const int xq = (int)(q % W), yq = (int)((q / W) % H), b = (int)(q / ((long long)W * H));
        const int c0 = grp * 4, nc = Cin - c0 < 4 ? Cin - c0 : 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yp = yq - (ky - 1);
            if (yp < 0 || yp >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xp = xq - (kx - 1);
                if (xp < 0 || xp >= W) continue;
                const size_t p = ((size_t)b * H + yp) * W + xp;
                const float* wt = wl + ((ky * 3 + kx) * Cout) * Cin + c0;
                for (int co = 0; co < Cout; ++co) {
                    const float d = dy[p * Cout + co] * leaky_mask(act, p * Cout + co, slope);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < nc) acc[e] = fmaf(d, wt[co * Cin + e], acc[e]);
                }
            }
        }
        if (nchw) {
            for (int e = 0; e < nc; ++e) dx[(((size_t)b * Cin + c0 + e) * H + yq) * W + xq] = acc[e];
        } else {
            for (int e = 0; e < nc; ++e) dx[(size_t)q * Cin + c0 + e] = acc[e];
        }
    }
}

// The same input gradient for the wide-input layer (OutputProj: token rows, Cin % 4 == 0, S = Cout <= 4, no activation mask), WALKING
// along x: a thread owns 4 input channels of SEG consecutive pixels of one row, holds its 9 x S x 4 weights in registers (they were
// 108 LDS reads per output pixel) and a 3 x 3 x S window of dy that slides one column per step (3 S loads instead of 9 S).
template <int S, int SEG>
__global__ __launch_bounds__(256) void conv3x3_dx_walk_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                              int B, int H, int W, int Cin) {
    const int groups = Cin / 4, segs = W / SEG;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * H * segs * groups) return;
    const int c0 = (int)(t % groups) * 4;
    long long rest = t / groups;
    const int x0 = (int)(rest % segs) * SEG; rest /= segs;
    const int y = (int)(rest % H), b = (int)(rest / H);
    float wr[9][S][4];                                                  // w[co][ci][ky][kx] -> wr[ky*3+kx][co][ci - c0]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int co = 0; co < S; ++co)
#pragma unroll
            for (int e = 0; e < 4; ++e) wr[tap][co][e] = w[((size_t)co * Cin + c0 + e) * 9 + tap];
    // dy rows y + 1, y, y - 1 pair with ky = 0, 1, 2  (dx[q] = sum dy[q - (ky-1, kx-1)] w[ky][kx])
    const float* rowp[3];
    float rm[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yp = y - (ky - 1);
        rm[ky] = (yp >= 0 && yp < H) ? 1.0f : 0.0f;
        rowp[ky] = dy + ((size_t)b * H + (yp < 0 ? 0 : (yp >= H ? H - 1 : yp))) * W * S;
    }
    float win[3][3][S];                                                 // [column slot][ky][co]; slots rotate: columns xx + 1, xx, xx - 1 pair with kx = 0, 1, 2
    auto load_col = [&](float (&cl)[3][S], int xp) {
        const float m = (xp >= 0 && xp < W) ? 1.0f : 0.0f;
        const int xc = xp < 0 ? 0 : (xp >= W ? W - 1 : xp);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int co = 0; co < S; ++co) cl[ky][co] = rowp[ky][(size_t)xc * S + co] * (m * rm[ky]);
    };
    auto emit = [&](const float (&cm)[3][S], const float (&cc)[3][S], const float (&cp)[3][S], int xx) {   // columns xx - 1, xx, xx + 1
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float (&cl)[3][S] = kx == 0 ? cp : (kx == 1 ? cc : cm);     // xp = xx - (kx - 1)
#pragma unroll
                for (int co = 0; co < S; ++co)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(cl[ky][co], wr[ky * 3 + kx][co][e], acc[e]);
            }
        *reinterpret_cast<f32x4*>(dx + (((size_t)b * H + y) * W + xx) * Cin + c0) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    };
    load_col(win[0], x0 - 1);
    load_col(win[1], x0);
#pragma unroll 1
    for (int xs = 0; xs < SEG; xs += 3) {
        load_col(win[2], x0 + xs + 1);
        emit(win[0], win[1], win[2], x0 + xs);
        if (xs + 1 < SEG) { load_col(win[0], x0 + xs + 2); emit(win[1], win[2], win[0], x0 + xs + 1); }
        if (xs + 2 < SEG) { load_col(win[1], x0 + xs + 3); emit(win[2], win[0], win[1], x0 + xs + 2); }
    }
}

// Weight gradient, WIDE input side (OutputProj: Cin = 64, Cout = 3).  Block = 3 waves; wave = ky, lane (+64 j) = ci.  A thread walks
// the columns u of input row y+ky-1 once: x[.][u][ci] is one coalesced load and meets dyeff[y][u+1-kx][co] for kx = 0..2 -- a
// sliding window of wave-uniform values.  Partial sums of the block's rows go to partial[block][...] in the reference's
// (Cout,Cin,3,3) order, then db (Cout values, lane 0 of wave 1); column_sum_kernel adds the blocks in order.
// (Two rewrites that read the row once for all three ky -- one wave with 27 accumulators per lane, and four waves splitting the columns
// with an LDS reduction -- measured 649 and 1692 us against 498 us for this form on the 2.1 M x 64 head input: profiles/r03_stemhead.txt.)
template <int S>   // S = Cout <= 4
__global__ __launch_bounds__(192) void conv3x3_wgrad_wide_in_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
                                                                    int B, int H, int W, int Cin, int rows_per_block) {
    const int ky = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int E = S * Cin * 9 + S;
    float* out = partial + (size_t)blockIdx.x * E;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(B * H, r0 + rows_per_block);
    float dbs[S];
#pragma unroll
    for (int s = 0; s < S; ++s) dbs[s] = 0.f;
    for (int cbase = 0; cbase < Cin; cbase += 64) {
        const int ci = cbase + lane;
        const bool live = ci < Cin;
        float acc[S][3];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s][0] = acc[s][1] = acc[s][2] = 0.f;
        for (int r = r0; r < r1; ++r) {
            const int b = r / H, y = r - b * H;
            const int yi = y + ky - 1;
            if (yi < 0 || yi >= H) continue;                                  // wave-uniform
            const float* xr = x + (((size_t)b * H + yi) * W) * Cin + (live ? ci : 0);
            const float* dr = dy + (((size_t)b * H + y) * W) * S;
            float d0[S], d1[S], d2[S];                                          // dyeff at columns u-1, u, u+1
#pragma unroll
            for (int s = 0; s < S; ++s) { d0[s] = 0.f; d1[s] = 0.f; d2[s] = dr[s]; }
            // u = -1 (only kx = 0 would pair x[-1], which is padding): start at u = 0 with the window (dy[-1] = 0, dy[0], dy[1])
#pragma unroll
            for (int s = 0; s < S; ++s) { d0[s] = d1[s]; d1[s] = d2[s]; d2[s] = W > 1 ? dr[S + s] : 0.f; }
#pragma unroll 4
            for (int u = 0; u < W; ++u) {
                const float xv = xr[(size_t)u * Cin];
#pragma unroll
                for (int s = 0; s < S; ++s) {                                   // x column u = p.x + kx - 1  ->  p.x = u + 1 - kx
                    acc[s][0] = fmaf(d2[s], xv, acc[s][0]);
                    acc[s][1] = fmaf(d1[s], xv, acc[s][1]);
                    acc[s][2] = fmaf(d0[s], xv, acc[s][2]);
                }
                if (cbase == 0 && ky == 1) {
#pragma unroll
                    for (int s = 0; s < S; ++s) dbs[s] += d1[s];
                }
#pragma unroll
                for (int s = 0; s < S; ++s) { d0[s] = d1[s]; d1[s] = d2[s]; d2[s] = (u + 2 < W) ? dr[(size_t)(u + 2) * S + s] : 0.f; }
            }
        }
        if (live) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) out[((s * Cin + ci) * 3 + ky) * 3 + kx] = acc[s][kx];
        }
    }
    if (ky == 1 && lane == 0) {
#pragma unroll
        for (int s = 0; s < S; ++s) out[S * Cin * 9 + s] = dbs[s];
    }
}

// Weight gradient, WIDE output side (InputProj: Cin = 3 NCHW image, Cout = 32, LeakyReLU' folded in).  Block = one wave;
// lane = (pixel slot, co): the 64 / Cout slots each walk a contiguous run of the row (Cout 16, 32 or 64).  dyeff[p][co] is the
// coalesced load; the image values around p are the same address for all lanes of a slot.  Partials as above.
template <int S>   // S = Cin <= 4
__global__ __launch_bounds__(64) void conv3x3_wgrad_wide_out_kernel(const float* __restrict__ img, const float* __restrict__ dy, const float* __restrict__ act, float slope,
                                                                    float* __restrict__ partial, int B, int H, int W, int Cout, int rows_per_block) {
    const int lane = threadIdx.x;
    const int ppw = 64 / Cout;                                                 // pixel slots per wave
    const int co = lane % Cout, par = lane / Cout;
    const int E = Cout * S * 9 + Cout;
    float* out = partial + ((size_t)blockIdx.x * ppw + par) * E;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(B * H, r0 + rows_per_block);
    const int seg = (W + ppw - 1) / ppw, xa = par * seg, xb = min(W, xa + seg);   // this slot's contiguous run of the row
    float acc[S][3][3], dbv = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[s][k / 3][k % 3] = 0.f;
    for (int r = r0; r < r1; ++r) {
        const int b = r / H, y = r - b * H;
        const size_t prow = ((size_t)b * H + y) * W;
        const float* ir[3];
        float rm[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yi = y + ky - 1;
            rm[ky] = (yi >= 0 && yi < H) ? 1.0f : 0.0f;
            ir[ky] = img + ((size_t)b * S * H + (yi < 0 ? 0 : (yi >= H ? H - 1 : yi))) * W;     // plane s at + s*H*W
        }
        // the 3 x 3 image window around the pixel slides along the run: S x 3 new values per step (they were 9 S per step, in each of the
        // three ky waves of the first version, each of which also read dy and the activation again)
        float win[S][3][3];
        auto col = [&](int px, int slot) {
            const float m = (px >= 0 && px < W) ? 1.0f : 0.0f;
            const int pc = px < 0 ? 0 : (px >= W ? W - 1 : px);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) win[s][ky][slot] = ir[ky][(size_t)s * H * W + pc] * (m * rm[ky]);
        };
        col(xa - 1, 1);
        col(xa, 2);
#pragma unroll 2
        for (int px = xa; px < xb; ++px) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) { win[s][ky][0] = win[s][ky][1]; win[s][ky][1] = win[s][ky][2]; }
            col(px + 1, 2);
            const size_t i = (prow + px) * Cout + co;
            const float d = dy[i] * leaky_mask(act, i, slope);
            dbv += d;
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc[s][ky][kx] = fmaf(d, win[s][ky][kx], acc[s][ky][kx]);
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) out[((co * S + s) * 3 + ky) * 3 + kx] = acc[s][ky][kx];
    out[Cout * S * 9 + co] = dbv;
}

}  // namespace
}  // namespace uf

// one-wave workgroups, each over a few image rows: enough of them to keep ~16 waves per CU streaming (1024 three-wave blocks before)
static int conv3x3_bwd_blocks(int B, int H) { const int rows = B * H; return rows < 4096 ? rows : 4096; }

extern "C" size_t uf_conv3x3_bwd_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    return (size_t)conv3x3_bwd_blocks(B, H) * 4 * ((size_t)Cin * Cout * 9 + Cout) * sizeof(float);
}

extern "C" int uf_conv3x3_bwd(const float* x, int x_nchw, const float* dy, const float* act_out, float slope, const float* w, float* dx, float* dW, float* db,
                              int B, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(x && dy && w && dW && db && ws, UF_ERR_NULL, "uf_conv3x3_bwd: null pointer");
    const bool wide_in = Cout <= 4 && !x_nchw && Cin % 4 == 0;
    const bool wide_out = Cin <= 4 && x_nchw && (Cout == 16 || Cout == 32 || Cout == 64);
    UF_REQUIRE(B > 0 && H > 0 && W > 1 && (wide_in || wide_out) && Cin * Cout * 9 <= CB_MAXW, UF_ERR_UNSUPPORTED,
               "uf_conv3x3_bwd: Cin=%d Cout=%d nchw=%d: supported are (token rows, Cin %% 4 == 0, Cout <= 4) and (NCHW, Cin <= 4, Cout 16, 32 or 64)", Cin, Cout, x_nchw);
    UF_REQUIRE(!act_out || wide_out, UF_ERR_UNSUPPORTED, "uf_conv3x3_bwd: the LeakyReLU mask is folded on the InputProj form only");
    const size_t need = uf_conv3x3_bwd_workspace_bytes(B, H, W, Cin, Cout);
    UF_REQUIRE(ws_bytes >= need, UF_ERR_WORKSPACE, "uf_conv3x3_bwd: workspace too small: %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        if (wide_in && !act_out && W % 16 == 0 && Cout == 3 && ((uintptr_t)dx % 16) == 0) {    // the OutputProj form: walking kernel
            const long long n = (long long)B * H * (W / 16) * (Cin / 4);
            hipLaunchKernelGGL((conv3x3_dx_walk_kernel<3, 16>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, w, dx, B, H, W, Cin);
        } else {
            const int grid = grid1d((long long)B * H * W * ((Cin + 3) / 4));
            hipLaunchKernelGGL(conv3x3_dx_kernel, dim3(grid), dim3(256), 0, st, dy, act_out, slope, w, dx, B, H, W, Cin, Cout, x_nchw);
        }
        if (int rc = check_launch("conv3x3_dx")) return rc;
    }
    const int blocks = conv3x3_bwd_blocks(B, H), rpb = (B * H + blocks - 1) / blocks;
    const int E = Cin * Cout * 9 + Cout;
    int P = blocks;
    float* part = (float*)ws;
    if (wide_in) {
        switch (Cout) {
            case 1: hipLaunchKernelGGL(conv3x3_wgrad_wide_in_kernel<1>, dim3(blocks), dim3(192), 0, st, x, dy, part, B, H, W, Cin, rpb); break;
            case 2: hipLaunchKernelGGL(conv3x3_wgrad_wide_in_kernel<2>, dim3(blocks), dim3(192), 0, st, x, dy, part, B, H, W, Cin, rpb); break;
            case 3: hipLaunchKernelGGL(conv3x3_wgrad_wide_in_kernel<3>, dim3(blocks), dim3(192), 0, st, x, dy, part, B, H, W, Cin, rpb); break;
            default: hipLaunchKernelGGL(conv3x3_wgrad_wide_in_kernel<4>, dim3(blocks), dim3(192), 0, st, x, dy, part, B, H, W, Cin, rpb); break;
        }
    } else {
        P = blocks * (64 / Cout);
        switch (Cin) {
            case 1: hipLaunchKernelGGL(conv3x3_wgrad_wide_out_kernel<1>, dim3(blocks), dim3(64), 0, st, x, dy, act_out, slope, part, B, H, W, Cout, rpb); break;
            case 2: hipLaunchKernelGGL(conv3x3_wgrad_wide_out_kernel<2>, dim3(blocks), dim3(64), 0, st, x, dy, act_out, slope, part, B, H, W, Cout, rpb); break;
            case 3: hipLaunchKernelGGL(conv3x3_wgrad_wide_out_kernel<3>, dim3(blocks), dim3(64), 0, st, x, dy, act_out, slope, part, B, H, W, Cout, rpb); break;
            default: hipLaunchKernelGGL(conv3x3_wgrad_wide_out_kernel<4>, dim3(blocks), dim3(64), 0, st, x, dy, act_out, slope, part, B, H, W, Cout, rpb); break;
        }
    }
    if (int rc = check_launch("conv3x3_wgrad")) return rc;
    const int nw = Cin * Cout * 9;
    launch_column_sum2(st, (const float*)part, P, (size_t)E, dW, nw, (const float*)part + nw, P, (size_t)E, db, Cout);
    return check_launch("conv3x3_wgrad_finalize");
}
