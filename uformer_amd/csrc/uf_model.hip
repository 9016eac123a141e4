// Host-side drivers: the LeWin block (attention half + LeFF half), the samplers and the whole
// Uformer forward, expressed as launches of the kernels in uf_gemm / uf_attn / uf_elementwise.
// Mirrors LeWinTransformerBlock.forward (model.py:908-989), BasicUformerLayer.forward
// (:1054-1060) and Uformer.forward (:1269-1305) of the reference.
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "uf_internal.h"

namespace uf {

int launch_layernorm(const float* x, int ld_x, const float* gamma, const float* beta, const float* modulator, void* out,
                     int rows, int H, int W, int C, int windowed, int shift, uf_dtype dtype, hipStream_t st);

namespace {

struct BlockWs {
    char* a;   // T[M][C]   : LN output / attention output
    char* h1;  // T[M][4C]  : q,k,v^T (first 3*M*C) / LeFF hidden after fc1
    char* h2;  // T[M][4C]  : LeFF hidden after the depthwise conv
};

size_t block_ws_bytes(size_t M, size_t C, uf_dtype dtype) {
    const size_t sz = dtype_size(dtype);
    return align_up(M * C * sz, 256) + 2 * align_up(M * 4 * C * sz, 256);
}

int carve(BlockWs& w, void* ws, size_t ws_bytes, size_t M, size_t C, uf_dtype dtype) {
    UF_REQUIRE(ws, UF_ERR_NULL, "workspace is null");
    UF_REQUIRE(((uintptr_t)ws % 256) == 0, UF_ERR_ALIGN, "workspace must be 256-byte aligned");
    const size_t need = block_ws_bytes(M, C, dtype);
    UF_REQUIRE(ws_bytes >= need, UF_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, need);
    const size_t sz = dtype_size(dtype);
    w.a = (char*)ws;
    w.h1 = w.a + align_up(M * C * sz, 256);
    w.h2 = w.h1 + align_up(M * 4 * C * sz, 256);
    return UF_OK;
}

int check_block_args(const uf_block_params* p, const float* x, int ld, int B, int H, int W, int C, uf_dtype dtype) {
    UF_REQUIRE(p && x, UF_ERR_NULL, "block: null pointer");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "block: dtype %d", (int)dtype);
    UF_REQUIRE(B > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0, UF_ERR_SHAPE, "block: B=%d H=%d W=%d (H,W multiples of 8)", B, H, W);
    UF_REQUIRE(C >= 16 && C % 16 == 0 && ld >= C && ld % 4 == 0, UF_ERR_SHAPE, "block: C=%d ld=%d", C, ld);
    UF_REQUIRE(p->heads > 0 && C % p->heads == 0, UF_ERR_SHAPE, "block: C=%d heads=%d", C, p->heads);
    UF_REQUIRE(p->shift == 0 || p->shift == 4, UF_ERR_UNSUPPORTED, "block: shift=%d (0 or 4)", p->shift);
    UF_REQUIRE((long long)B * H * W * 4LL * C < 0x7fffffffLL * 4, UF_ERR_SHAPE, "block: tensor too large for 32-bit row indexing");
    return UF_OK;
}

// fc1_done (optional): set when the fused kernel also produced the LeFF hidden h1 in w.h1 (whole-block calls only)
int block_fwd(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C, const float* user_mask, int n_mask, uf_dtype dtype,
              const BlockWs& w, hipStream_t st, const float* drop_attn, const float* drop_leff);

int attn_half(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C, const float* user_mask, int n_mask,
              uf_dtype dtype, const BlockWs& w, hipStream_t st, bool* fc1_done = nullptr, const float* drop = nullptr) {
    const int M = B * H * W;
    const size_t sz = dtype_size(dtype);
    const int heads = p->heads, hd = C / heads;
    // one fused kernel per window when the shape is covered (head_dim 32): LN1, q/k/v, attention, proj,
    // window_reverse and the residual never leave the CU                       (model.py:951-986)
    // (Stages with fewer windows than CUs -- 4 ... 64 at small batches -- stay on the fused kernel too: the 3-kernel path, whose GEMMs tile over the
    // whole chip, measured slower at every batch size from 1 to 16: profiles/r03_unfuse.txt, profiles/r05_run5_unfuse.txt.)
    if (attn_block_supported(p, user_mask, dtype, C, heads)) {
        const bool with_fc1 = fc1_done && dtype_half(dtype) && C >= 32;
        if (fc1_done) *fc1_done = with_fc1;
        return launch_attn_block(p, x, ld, B, H, W, C, dtype, with_fc1 ? w.h1 : nullptr, st, drop);
    }
    UF_REQUIRE(!drop, UF_ERR_UNSUPPORTED, "DropPath scales need the fused attention kernel (bf16 / f16, or f32 with C <= 256; no caller mask)");
    // LN1 -> roll -> partition -> + modulator -> q,k,v projections, one kernel
    // (model.py:952-969, :431-442, :497)
    char* q = w.h1;
    char* k = q + (size_t)M * C * sz;
    char* vt = k + (size_t)M * C * sz;
    int rc = uf_ln_qkv_fwd(x, ld, p->norm1_w, p->norm1_b, p->modulator, p->wqkv_fm, p->bqkv, q, k, vt, B, H, W, C, heads,
                           p->shift, dtype, st);
    if (rc) return rc;
    // softmax(q k^T + bias + mask) v                    (model.py:498-519)
    rc = uf_window_attention_fwd(q, k, vt, p->rpb_dense, user_mask, n_mask, w.a, M / 64, heads, hd, H, W, p->shift, dtype, st);
    if (rc) return rc;
    // proj, window_reverse, roll back, + shortcut       (model.py:520, :975-986)
    GemmParams g{};
    g.A = w.a; g.lda = C; g.W = p->wproj; g.bias = p->bproj; g.M = M; g.N = C; g.K = C;
    g.H = H; g.W_ = W; g.shift = p->shift;
    g.out = x; g.ldo = ld; g.resid = x; g.ldr = ld;
    return launch_gemm(g, A_PLAIN, E_RES_WINREV, dtype, st);
}

int leff_half(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C, uf_dtype dtype, const BlockWs& w,
              hipStream_t st, bool fc1_done = false, const float* drop = nullptr) {
    const int M = B * H * W;
    // LN2 -> linear1 -> GELU, one kernel                 (model.py:987, :657-658, :671) unless the attention kernel did it
    if (!fc1_done) {
        int rc = uf_ln_linear_gelu_fwd(x, ld, p->norm2_w, p->norm2_b, p->w1_fm, p->b1, w.h1, M, 4 * C, C, dtype, st);
        if (rc) return rc;
    }
    // depthwise 3x3 + GELU over the whole H x W map, linear2, + residual: one kernel, the conv output
    // stays on chip                                      (model.py:659-661, :674-682, :987)
    return launch_leff2(w.h1, p->wdw9, p->bdw, p->w2_fm, p->b2, x, ld, B, H, W, C, dtype, drop, st);
}

int block_fwd(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C, const float* user_mask, int n_mask, uf_dtype dtype,
              const BlockWs& w, hipStream_t st, const float* drop_attn, const float* drop_leff) {
    bool fc1_done = false;
    int rc = attn_half(p, x, ld, B, H, W, C, user_mask, n_mask, dtype, w, st, &fc1_done, drop_attn);
    if (rc) return rc;
    return leff_half(p, x, ld, B, H, W, C, dtype, w, st, fc1_done, drop_leff);
}

}  // namespace
}  // namespace uf

using namespace uf;

extern "C" size_t uf_block_workspace_bytes(int M, int C, uf_dtype dtype) {
    if (M <= 0 || C <= 0) return 0;
    return block_ws_bytes((size_t)M, (size_t)C, dtype);
}

extern "C" int uf_lewin_attn_fwd(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C,
                                 const float* user_mask, int n_mask, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_block_args(p, x, ld, B, H, W, C, dtype);
    if (rc) return rc;
    BlockWs w;
    rc = carve(w, ws, ws_bytes, (size_t)B * H * W, C, dtype);
    if (rc) return rc;
    return attn_half(p, x, ld, B, H, W, C, user_mask, n_mask, dtype, w, (hipStream_t)stream);
}

extern "C" int uf_leff_fwd(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C, uf_dtype dtype, void* ws,
                           size_t ws_bytes, void* stream) {
    int rc = check_block_args(p, x, ld, B, H, W, C, dtype);
    if (rc) return rc;
    BlockWs w;
    rc = carve(w, ws, ws_bytes, (size_t)B * H * W, C, dtype);
    if (rc) return rc;
    return leff_half(p, x, ld, B, H, W, C, dtype, w, (hipStream_t)stream);
}

extern "C" int uf_lewin_block_fwd(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C,
                                  const float* user_mask, int n_mask, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_block_args(p, x, ld, B, H, W, C, dtype);
    if (rc) return rc;
    BlockWs w;
    rc = carve(w, ws, ws_bytes, (size_t)B * H * W, C, dtype);
    if (rc) return rc;
    return block_fwd(p, x, ld, B, H, W, C, user_mask, n_mask, dtype, w, (hipStream_t)stream, nullptr, nullptr);
}

extern "C" int uf_lewin_attn_train_fwd(const uf_block_params* p, const float* x, int ld, float* x1, int ld1, int B, int H, int W, int C,
                                       const float* drop_attn, uf_dtype dtype, void* xn, void* q, void* k, void* vt, void* o, void* z, void* a1,
                                       void* stream) {
    int rc = check_block_args(p, x, ld, B, H, W, C, dtype);
    if (rc) return rc;
    UF_REQUIRE(x1 && xn && q && k && vt && o && z && a1, UF_ERR_NULL, "uf_lewin_attn_train_fwd: null output pointer");
    UF_REQUIRE(x1 != x && ld1 >= C && ld1 % 4 == 0, UF_ERR_SHAPE, "uf_lewin_attn_train_fwd: x1 must be out of place, ld1=%d", ld1);
    UF_REQUIRE(dtype_half(dtype) && attn_block_supported(p, nullptr, dtype, C, p->heads), UF_ERR_UNSUPPORTED,
               "uf_lewin_attn_train_fwd: bf16 / f16 operands, head_dim 32, C = 32 ... 512 and the compact bias table (got dtype %d, C=%d, heads=%d)", (int)dtype, C, p->heads);
    const uintptr_t al = (uintptr_t)x1 | (uintptr_t)xn | (uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)o | (uintptr_t)z | (uintptr_t)a1;
    UF_REQUIRE((al % 16) == 0, UF_ERR_ALIGN, "uf_lewin_attn_train_fwd: outputs must be 16-byte aligned");
    return launch_attn_block_train(p, x, ld, x1, ld1, B, H, W, C, dtype, drop_attn, xn, q, k, vt, o, z, a1, (hipStream_t)stream);
}

extern "C" int uf_lewin_block_train_fwd(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C, const float* drop_attn,
                                        const float* drop_leff, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_block_args(p, x, ld, B, H, W, C, dtype);
    if (rc) return rc;
    BlockWs w;
    rc = carve(w, ws, ws_bytes, (size_t)B * H * W, C, dtype);
    if (rc) return rc;
    return block_fwd(p, x, ld, B, H, W, C, nullptr, 0, dtype, w, (hipStream_t)stream, drop_attn, drop_leff);
}

extern "C" int uf_downsample_fwd(const float* x, int ld_x, const void* w, const float* bias, float* out, int ld_o, int B,
                                 int H, int W, int C, uf_dtype dtype, void* stream) {
    UF_REQUIRE(x && w && bias && out, UF_ERR_NULL, "uf_downsample_fwd: null pointer");
    UF_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0 && H > 0 && W > 0, UF_ERR_SHAPE, "uf_downsample_fwd: H=%d W=%d", H, W);
    UF_REQUIRE(ld_x >= C && ld_o >= 2 * C && ld_o % 4 == 0, UF_ERR_SHAPE, "uf_downsample_fwd: ld_x=%d ld_o=%d C=%d", ld_x, ld_o, C);
    GemmParams g{};
    g.A = x; g.lda = ld_x; g.W = w; g.bias = bias; g.M = B * (H / 2) * (W / 2); g.N = 2 * C; g.K = 16 * C;
    g.H = H; g.W_ = W; g.C = C; g.out = out; g.ldo = ld_o;
    return launch_gemm(g, A_CONV_DOWN, E_STORE_R, dtype, (hipStream_t)stream);
}

extern "C" int uf_downsample_fm_fwd(const float* x, int ld_x, const void* w, const void* w_fm, const float* bias, float* out, int ld_o, int B,
                                    int H, int W, int C, uf_dtype dtype, void* stream) {
    UF_REQUIRE(x && w && bias && out, UF_ERR_NULL, "uf_downsample_fm_fwd: null pointer");
    UF_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0 && H > 0 && W > 0, UF_ERR_SHAPE, "uf_downsample_fm_fwd: H=%d W=%d", H, W);
    UF_REQUIRE(ld_x >= C && ld_o >= 2 * C && ld_o % 4 == 0, UF_ERR_SHAPE, "uf_downsample_fm_fwd: ld_x=%d ld_o=%d C=%d", ld_x, ld_o, C);
    UF_REQUIRE(!w_fm || ((uintptr_t)w_fm % 16) == 0, UF_ERR_ALIGN, "uf_downsample_fm_fwd: w_fm must be 16-byte aligned");
    GemmParams g{};
    g.A = x; g.lda = ld_x; g.W = w; g.W_fm = dtype_half(dtype) ? w_fm : nullptr; g.bias = bias; g.M = B * (H / 2) * (W / 2); g.N = 2 * C; g.K = 16 * C;
    g.H = H; g.W_ = W; g.C = C; g.out = out; g.ldo = ld_o;
    return launch_gemm(g, A_CONV_DOWN, E_STORE_R, dtype, (hipStream_t)stream);
}

extern "C" int uf_upsample_fwd(const float* x, int ld_x, const void* w, const float* bias, float* out, int ld_o, int B, int H,
                               int W, int Cin, int Cout, uf_dtype dtype, void* stream) {
    UF_REQUIRE(x && w && bias && out, UF_ERR_NULL, "uf_upsample_fwd: null pointer");
    UF_REQUIRE(B > 0 && H > 0 && W > 0 && Cout % 4 == 0, UF_ERR_SHAPE, "uf_upsample_fwd: bad shape");
    UF_REQUIRE(ld_x >= Cin && ld_o >= Cout && ld_o % 4 == 0, UF_ERR_SHAPE, "uf_upsample_fwd: ld_x=%d ld_o=%d", ld_x, ld_o);
    GemmParams g{};
    g.A = x; g.lda = ld_x; g.W = w; g.bias = bias; g.M = B * H * W; g.N = 4 * Cout; g.K = Cin;
    g.H = H; g.W_ = W; g.Cout = Cout; g.out = out; g.ldo = ld_o;
    return launch_gemm(g, A_FROM_R, E_UPSAMPLE, dtype, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// whole model
// ------------------------------------------------------------------------------------------
namespace {
struct Plan {
    int C[9];        // block width per stage
    int res[9];      // H (=W) per stage
    size_t M[9];     // tokens per stage
    size_t off_D[4]; // byte offsets of the decoder concat buffers D0..D3 (f32 [M][2*Cskip])
    size_t off_P;    // bottleneck stream
    size_t off_blk;  // block scratch
    size_t blk_bytes;
    size_t total;
};

int make_plan(Plan& pl, const uf_model_desc* d, int B, int H, int W, uf_dtype dtype) {
    UF_REQUIRE(d, UF_ERR_NULL, "model desc is null");
    UF_REQUIRE(H == W, UF_ERR_SHAPE, "Uformer needs square inputs (reference takes sqrt(L), model.py:910-911): H=%d W=%d", H, W);
    UF_REQUIRE(H % 128 == 0 && H > 0, UF_ERR_SHAPE, "H=W=%d must be a multiple of 128 (4 downsamplings x window 8)", H);
    UF_REQUIRE(B > 0, UF_ERR_SHAPE, "B=%d", B);
    UF_REQUIRE(d->embed_dim >= 16 && d->embed_dim % 16 == 0, UF_ERR_SHAPE, "embed_dim=%d must be a multiple of 16", d->embed_dim);
    const int e = d->embed_dim;
    const int mult[9] = {1, 2, 4, 8, 16, 16, 8, 4, 2};
    const int div[9] = {1, 2, 4, 8, 16, 8, 4, 2, 1};
    size_t blk = 0;
    for (int s = 0; s < 9; ++s) {
        pl.C[s] = e * mult[s];
        pl.res[s] = H / div[s];
        pl.M[s] = (size_t)B * pl.res[s] * pl.res[s];
        const size_t b = block_ws_bytes(pl.M[s], pl.C[s], dtype);
        if (b > blk) blk = b;
    }
    size_t off = 0;
    for (int k = 0; k < 4; ++k) {  // D_k lives at decoder stage 5+k: rows M[5+k], width C[5+k]
        pl.off_D[k] = off;
        off += align_up(pl.M[5 + k] * pl.C[5 + k] * sizeof(float), 256);
    }
    pl.off_P = off;
    off += align_up(pl.M[4] * pl.C[4] * sizeof(float), 256);
    pl.off_blk = off;
    pl.blk_bytes = blk;
    pl.total = off + blk;
    return UF_OK;
}
}  // namespace

// slack so that the per-part plans of the multi-stream mode (every buffer 256-byte aligned) fit the workspace of the whole batch
constexpr size_t WS_SPLIT_SLACK = 256 * 256;

extern "C" size_t uf_uformer_workspace_bytes(const uf_model_desc* d, int B, int H, int W, uf_dtype dtype) {
    Plan pl;
    if (make_plan(pl, d, B, H, W, dtype) != UF_OK) return 0;
    return pl.total + WS_SPLIT_SLACK;
}

namespace {
int forward_one_stream(const uf_model_desc* d, const float* img, float* out, int B, int H, int W, uf_dtype dtype,
                       void* ws, size_t ws_bytes, void* stream) {
    Plan pl;
    int rc = make_plan(pl, d, B, H, W, dtype);
    if (rc) return rc;
    UF_REQUIRE(img && out && ws && d->blocks, UF_ERR_NULL, "uf_uformer_fwd: null pointer");
    UF_REQUIRE(((uintptr_t)ws % 256) == 0, UF_ERR_ALIGN, "uf_uformer_fwd: workspace must be 256-byte aligned");
    UF_REQUIRE(ws_bytes >= pl.total, UF_ERR_WORKSPACE, "uf_uformer_fwd: workspace too small: %zu < %zu", ws_bytes, pl.total);
    UF_REQUIRE(d->in_chans == 3, UF_ERR_UNSUPPORTED, "uf_uformer_fwd: in_chans=%d (3 supported)", d->in_chans);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    float* D[4];
    for (int k = 0; k < 4; ++k) D[k] = (float*)(base + pl.off_D[k]);
    float* P = (float*)(base + pl.off_P);
    void* bws = base + pl.off_blk;

    // stream of encoder stage s lives in the SECOND half of the concat buffer of decoder stage
    // 8-s (torch.cat([up, skip], -1), model.py:1288): no concat copy is ever made.
    auto enc_view = [&](int s, float*& ptr, int& ld) {
        ptr = D[3 - s] + pl.C[s];
        ld = 2 * pl.C[s];
    };
    const uf_block_params* blk = d->blocks;
    // (Running a block's two kernels image chunk by image chunk, so that h1 stays inside the 256 MiB Infinity Cache between them, measured neutral
    // in round 3 -- profiles/r03_chunk_ab.txt -- and is gone.)
    auto run_stage = [&](int s, float* x, int ld) -> int {
        for (int i = 0; i < d->depths[s]; ++i, ++blk) {
            int r = uf_lewin_block_fwd(blk, x, ld, B, pl.res[s], pl.res[s], pl.C[s], nullptr, 0, dtype, bws, pl.blk_bytes, st);
            if (r) return r;
        }
        return UF_OK;
    };

    float* x; int ld;
    enc_view(0, x, ld);
    rc = uf_input_proj_fwd(img, d->in_w27, d->in_b, x, ld, B, d->dd_in, H, W, d->embed_dim, st);  // model.py:1271
    if (rc) return rc;
    for (int s = 0; s < 4; ++s) {  // encoder, model.py:1274-1281
        rc = run_stage(s, x, ld);
        if (rc) return rc;
        float* nx; int nld;
        if (s < 3) enc_view(s + 1, nx, nld); else { nx = P; nld = pl.C[4]; }
        rc = uf_downsample_fm_fwd(x, ld, d->down_w[s], d->down_w_fm[s], d->down_b[s], nx, nld, B, pl.res[s], pl.res[s], pl.C[s], dtype, st);
        if (rc) return rc;
        x = nx; ld = nld;
    }
    rc = run_stage(4, x, ld);  // bottleneck, model.py:1284
    if (rc) return rc;
    for (int k = 0; k < 4; ++k) {  // decoder, model.py:1287-1301
        const int s = 5 + k;
        const int cin = pl.C[s - 1], cout = pl.C[s] / 2;
        rc = uf_upsample_fwd(x, ld, d->up_w[k], d->up_b[k], D[k], pl.C[s], B, pl.res[s - 1], pl.res[s - 1], cin, cout, dtype, st);
        if (rc) return rc;
        x = D[k]; ld = pl.C[s];
        rc = run_stage(s, x, ld);
        if (rc) return rc;
    }
    // output projection + global residual, model.py:1304-1305
    return uf_output_proj_fwd(x, ld, d->out_w, d->out_b, img, out, B, H, W, pl.C[8], d->dd_in == 3 ? 1 : 0, st);
}

}  // namespace

// Images never interact (LN per token, attention per window, convolutions per image), so the batch can be cut into
// parts enqueued on separate HIP streams: the ramp-up and tail of every kernel of one part (CUs idle while the last
// workgroups finish) overlap with kernels of the others.  Default: 2 parts for B >= 8, else 1; UF_STREAMS=1..8 overrides.
// Measured on one box, Uformer-B 256x256 B=16: 1 stream 2085 img/s, 2: 2135, 3: 2145, 4: 2100, 8: 1235 (host launch
// bound).  Results are bit-identical to the one-stream run (every kernel is batch-size invariant,
// tests/test_gpu_model.py); the caller's stream is ordered after all parts before the call returns.
extern "C" int uf_uformer_fwd(const uf_model_desc* d, const float* img, float* out, int B, int H, int W, uf_dtype dtype,
                              void* ws, size_t ws_bytes, void* stream) {
    static const int env_streams = getenv("UF_STREAMS") ? atoi(getenv("UF_STREAMS")) : 0;
    int n = env_streams > 0 ? (env_streams > MAX_SIDE + 1 ? MAX_SIDE + 1 : env_streams) : (B >= 8 ? 2 : 1);
    if (n > B) n = B;
    if (n < 2 || timing_enabled()) return forward_one_stream(d, img, out, B, H, W, dtype, ws, ws_bytes, stream);
    UF_REQUIRE(d && img && out && ws, UF_ERR_NULL, "uf_uformer_fwd: null pointer");
    int dev = 0;
    Lane* ss = acquire_lane(n - 1, &dev);
    UF_REQUIRE(ss, UF_ERR_LAUNCH, "uf_uformer_fwd: could not create the side streams");
    struct LaneGuard { Lane* l; int d; ~LaneGuard() { release_lane(l, d); } } guard{ss, dev};
    hipStream_t st = (hipStream_t)stream;
    UF_REQUIRE(hipEventRecord(ss->fork, st) == hipSuccess, UF_ERR_LAUNCH, "uf_uformer_fwd: hipEventRecord failed");
    int rc_all = UF_OK, b0 = 0;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {   // part i: images [b0, b0 + Bi); part 0 runs on the caller's stream
        const int Bi = B / n + (i < B % n ? 1 : 0);
        Plan pl;
        int rc = make_plan(pl, d, Bi, H, W, dtype);
        if (rc) return rc;
        UF_REQUIRE(ws_bytes >= off + pl.total, UF_ERR_WORKSPACE, "uf_uformer_fwd: workspace too small for %d streams", n);
        hipStream_t si = i == 0 ? st : ss->s[i - 1];
        if (i > 0) hipStreamWaitEvent(si, ss->fork, 0);
        rc = forward_one_stream(d, img + (size_t)b0 * d->dd_in * H * W, out + (size_t)b0 * 3 * H * W, Bi, H, W, dtype, (char*)ws + off,
                                pl.total, si);
        if (rc && !rc_all) rc_all = rc;
        if (i > 0) {
            hipEventRecord(ss->join[i - 1], si);
            hipStreamWaitEvent(st, ss->join[i - 1], 0);
        }
        off += align_up(pl.total, 256);
        b0 += Bi;
    }
    return rc_all;
}
