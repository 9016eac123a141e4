// Per-step operand packing of one LeWin block for training (row a15): the reference's f32 parameter tensors -> every layout the
// fused forward and the block-level backward read, in 5 launches (4 matrices + 1 for the vectors / tables) instead of ~50 ATen
// casts, transposes, cats, gathers and clones per block (the weights change every optimizer step, so this runs 40 times per step:
// it was 10 % of the GPU time of a training step, all of it 3-5 us kernels).
//   matrix W (N,K) f32 [rows from up to two tensors: to_q | to_kv]  ->  T row-major, T transposed (K,N), T fragment-major
//   bqkv = cat(to_q.bias, to_kv.bias); taps (4C,1,3,3) -> (9,4C) and the same with the tap axis reversed;
//   dense relative-position bias (heads,64,64) = table[index] (model.py:500-502) and the compact Toeplitz table (heads,15,15)
#include "uf_internal.h"

namespace uf {
namespace {

template <typename T> __device__ __forceinline__ void store8_t(T* p, const float* f) { *reinterpret_cast<u32x4*>(p) = pack8<T>(f); }   // 2-byte types
template <> __device__ __forceinline__ void store8_t<float>(float* p, const float* f) {
    *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
}

// the four matrices AND the vectors / tables of a block in ONE launch: workgroup ranges [0, e[0]), [e[0], e[1]), ... own one job each (the
// five launches were 3-7 us kernels, 200 per training step: 1.3 ms of a 76 ms step in profiles/r04_run13.txt)
struct MatJob { const float *src0, *src1; int n0, N, K; void *plain, *tr, *fm; };
template <typename T>
__device__ __forceinline__ void pack_linear_block(const MatJob& j, int blk) {
    const int kc = j.K / 8;
    const long long idx = (long long)blk * 256 + threadIdx.x;
    if (idx >= (long long)j.N * kc) return;
    const int n = (int)(idx / kc), k = (int)(idx % kc) * 8;
    const float* row = n < j.n0 ? j.src0 + (size_t)n * j.K : j.src1 + (size_t)(n - j.n0) * j.K;
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + k), b = *reinterpret_cast<const f32x4*>(row + k + 4);
    const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    T *plain = (T*)j.plain, *tr = (T*)j.tr, *fm = (T*)j.fm;
    if (plain) store8_t<T>(plain + (size_t)n * j.K + k, f);
    if (fm) {
        const int KS = j.K / 32;
        store8_t<T>(fm + ((((size_t)(n >> 4) * KS + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (n & 15)) << 3), f);
    }
    if (tr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) store1(tr + (size_t)(k + e) * j.N + n, f[e]);
    }
}

struct SmallPack {
    const float *qb, *kvb, *dw, *table;
    const long long* index;
    float *bqkv, *w9, *w9_flip, *dense, *tab;
    int C, heads;
};

// The depthwise taps are ROUNDED TO THE OPERAND TYPE here (round 6): the fused inference kernel evaluates the stencil on the matrix pipe with
// taps of the operand type (uf_leff2.hip, UF_MCONV = 1) while the op-level training kernels took the f32 taps -- eval and train forwards of the same weights
// differed in tap precision, and in the recompute form the loss came from one function and its gradient from another.  With the pack rounding them, every
// training kernel (forward stencil, input gradient with the flipped taps) and the fused forward see the same values; the tap GRADIENT stays f32 (straight through).
template <typename T> __device__ __forceinline__ float round_tap(float w) {
    if constexpr (sizeof(T) == 2) { T t; store1(&t, w); return load1(&t); }
    else return w;
}
template <typename T>
__device__ __forceinline__ void pack_small_items(const SmallPack& p, int first, int stride) {
    const int C = p.C, C4 = 4 * C;
    const int n_b = 3 * C, n_w = 9 * C4, n_d = p.heads * 4096, n_t = p.heads * 225;
    for (int i = first; i < n_b + 2 * n_w + n_d + n_t; i += stride) {
        int j = i;
        if (j < n_b) { p.bqkv[j] = j < C ? p.qb[j] : p.kvb[j - C]; continue; }
        j -= n_b;
        if (j < n_w) { const int t = j / C4, c = j - t * C4; p.w9[j] = round_tap<T>(p.dw[c * 9 + t]); continue; }
        j -= n_w;
        if (j < n_w) { const int t = j / C4, c = j - t * C4; p.w9_flip[j] = round_tap<T>(p.dw[c * 9 + 8 - t]); continue; }
        j -= n_w;
        if (j < n_d) { const int h = j >> 12, qk = j & 4095; p.dense[j] = p.table[(size_t)p.index[qk] * p.heads + h]; continue; }
        j -= n_d;
        // tab[h][dy+7][7-dx] = table[(dy+7)*15 + (dx+7)][h]   (the reference's relative_position_index, model.py:471-481)
        { const int h = j / 225, r = j - h * 225, a = r / 15, b = r - a * 15; p.tab[j] = p.table[(size_t)(a * 15 + (14 - b)) * p.heads + h]; }
    }
}
struct BlockPackJobs { MatJob m[4]; SmallPack sp; int e[4]; int n_small_blocks; };
template <typename T>
__global__ __launch_bounds__(256) void pack_block_kernel(const BlockPackJobs j) {
    const int b = blockIdx.x;                                   // workgroup-uniform dispatch
    if (b < j.e[0]) pack_linear_block<T>(j.m[0], b);
    else if (b < j.e[1]) pack_linear_block<T>(j.m[1], b - j.e[0]);
    else if (b < j.e[2]) pack_linear_block<T>(j.m[2], b - j.e[1]);
    else if (b < j.e[3]) pack_linear_block<T>(j.m[3], b - j.e[2]);
    else pack_small_items<T>(j.sp, (b - j.e[3]) * 256 + threadIdx.x, j.n_small_blocks * 256);
}

struct PackPlan {
    size_t wqkv, wqkv_t, wqkv_fm, wp, wp_t, wp_fm, w1, w1_t, w1_fm, w2, w2_t, w2_fm, bqkv, w9, w9_flip, dense, tab, total;
};

PackPlan plan_pack(int C, int heads, uf_dtype dtype) {
    const size_t sz = dtype_size(dtype), c2 = (size_t)C * C;
    size_t off = 0;
    auto take = [&](size_t bytes) { off = align_up(off, 256); const size_t o = off; off += bytes; return o; };
    PackPlan p;
    p.wqkv = take(3 * c2 * sz); p.wqkv_t = take(3 * c2 * sz); p.wqkv_fm = take(3 * c2 * sz);
    p.wp = take(c2 * sz);       p.wp_t = take(c2 * sz);       p.wp_fm = take(c2 * sz);
    p.w1 = take(4 * c2 * sz);   p.w1_t = take(4 * c2 * sz);   p.w1_fm = take(4 * c2 * sz);
    p.w2 = take(4 * c2 * sz);   p.w2_t = take(4 * c2 * sz);   p.w2_fm = take(4 * c2 * sz);
    p.bqkv = take((size_t)3 * C * 4); p.w9 = take((size_t)36 * C * 4); p.w9_flip = take((size_t)36 * C * 4);
    p.dense = take((size_t)heads * 4096 * 4); p.tab = take((size_t)heads * 225 * 4);
    p.total = align_up(off, 256);
    return p;
}

}  // namespace
}  // namespace uf

using namespace uf;

extern "C" size_t uf_pack_block_train_bytes(int C, int heads, uf_dtype dtype) {
    if (C <= 0 || heads <= 0 || !dtype_ok(dtype)) return 0;
    return plan_pack(C, heads, dtype).total;
}

extern "C" int uf_pack_block_train(const uf_block_raw_params* raw, int C, int heads, int shift, uf_dtype dtype, void* buf, size_t buf_bytes,
                                   uf_block_params* fwd, uf_block_train_params* bwd, void* stream) {
    UF_REQUIRE(raw && buf && (fwd || bwd), UF_ERR_NULL, "uf_pack_block_train: null pointer");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "uf_pack_block_train: dtype %d", (int)dtype);
    UF_REQUIRE(C >= 32 && C % 32 == 0 && heads > 0 && C % heads == 0, UF_ERR_SHAPE, "uf_pack_block_train: C=%d heads=%d (C a multiple of 32)", C, heads);
    UF_REQUIRE(raw->norm1_w && raw->norm1_b && raw->norm2_w && raw->norm2_b && raw->rpb_table && raw->rpb_index && raw->to_q_w && raw->to_q_b && raw->to_kv_w &&
                   raw->to_kv_b && raw->proj_w && raw->proj_b && raw->lin1_w && raw->lin1_b && raw->dw_w && raw->dw_b && raw->lin2_w && raw->lin2_b,
               UF_ERR_NULL, "uf_pack_block_train: a parameter pointer is NULL (only modulator may be)");
    const PackPlan pl = plan_pack(C, heads, dtype);
    UF_REQUIRE(buf_bytes >= pl.total && ((uintptr_t)buf % 256) == 0, UF_ERR_WORKSPACE, "uf_pack_block_train: buffer %zu < %zu bytes (256-byte aligned)", buf_bytes, pl.total);
    hipStream_t st = (hipStream_t)stream;
    char* b = (char*)buf;
    auto at = [&](size_t o) { return (void*)(b + o); };
    SmallPack sp{raw->to_q_b, raw->to_kv_b, raw->dw_w, raw->rpb_table, (const long long*)raw->rpb_index,
                 (float*)at(pl.bqkv), (float*)at(pl.w9), (float*)at(pl.w9_flip), (float*)at(pl.dense), (float*)at(pl.tab), C, heads};
    const int n_small = 3 * C + 72 * C + heads * (4096 + 225);
    {   // one launch for the four weight matrices and the small tensors (round 4; the five-launch form it replaced is gone)
        BlockPackJobs j;
        j.m[0] = MatJob{raw->to_q_w, raw->to_kv_w, C, 3 * C, C, at(pl.wqkv), at(pl.wqkv_t), at(pl.wqkv_fm)};
        j.m[1] = MatJob{raw->proj_w, nullptr, C, C, C, at(pl.wp), at(pl.wp_t), at(pl.wp_fm)};
        j.m[2] = MatJob{raw->lin1_w, nullptr, 4 * C, 4 * C, C, at(pl.w1), at(pl.w1_t), at(pl.w1_fm)};
        j.m[3] = MatJob{raw->lin2_w, nullptr, C, C, 4 * C, at(pl.w2), at(pl.w2_t), at(pl.w2_fm)};
        int end = 0;
        for (int q = 0; q < 4; ++q) {
            end += (int)(((long long)j.m[q].N * (j.m[q].K / 8) + 255) / 256);
            j.e[q] = end;
        }
        j.sp = sp;
        j.n_small_blocks = (n_small + 255) / 256;
        UF_DISPATCH(dtype, TT, hipLaunchKernelGGL(pack_block_kernel<TT>, dim3((unsigned)(end + j.n_small_blocks)), dim3(256), 0, st, j));
    }
    if (int rc = check_launch("pack_block_train")) return rc;
    if (fwd) {
        *fwd = uf_block_params{};
        fwd->norm1_w = raw->norm1_w; fwd->norm1_b = raw->norm1_b; fwd->modulator = raw->modulator;
        fwd->rpb_dense = (const float*)at(pl.dense); fwd->rpb_fm = nullptr; fwd->rpb_tab = raw->index_is_standard ? (const float*)at(pl.tab) : nullptr;
        fwd->wqkv_fm = at(pl.wqkv_fm); fwd->bqkv = (const float*)at(pl.bqkv);
        fwd->wproj = at(pl.wp); fwd->wproj_fm = at(pl.wp_fm); fwd->bproj = raw->proj_b;
        fwd->norm2_w = raw->norm2_w; fwd->norm2_b = raw->norm2_b;
        fwd->w1_fm = at(pl.w1_fm); fwd->b1 = raw->lin1_b; fwd->wdw9 = (const float*)at(pl.w9); fwd->bdw = raw->dw_b;
        fwd->w2_fm = at(pl.w2_fm); fwd->b2 = raw->lin2_b;
        fwd->shift = shift; fwd->heads = heads;
    }
    if (bwd) {
        *bwd = uf_block_train_params{};
        bwd->norm1_w = raw->norm1_w; bwd->norm1_b = raw->norm1_b; bwd->norm2_w = raw->norm2_w; bwd->norm2_b = raw->norm2_b;
        bwd->modulator = raw->modulator; bwd->rpb_dense = (const float*)at(pl.dense);
        bwd->wqkv = at(pl.wqkv); bwd->wqkv_t = at(pl.wqkv_t); bwd->bqkv = (const float*)at(pl.bqkv);
        bwd->wproj = at(pl.wp); bwd->wproj_t = at(pl.wp_t); bwd->bproj = raw->proj_b;
        bwd->w1 = at(pl.w1); bwd->w1_t = at(pl.w1_t); bwd->b1 = raw->lin1_b;
        bwd->wdw9 = (const float*)at(pl.w9); bwd->wdw9_flip = (const float*)at(pl.w9_flip); bwd->bdw = raw->dw_b;
        bwd->w2_t = at(pl.w2_t); bwd->w2 = at(pl.w2);
        bwd->shift = shift; bwd->heads = heads;
    }
    return UF_OK;
}
