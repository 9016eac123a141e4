// Block-level backward of the path (SURVEY section 8 row a15 / 8(b) export list): host-side composition, in C++, of the
// op-level kernels of this library into
//   uf_lewin_attn_bwd     gradient of x1 = x + DropPath(proj(attn(LN1(x) ...)))         (model.py:951-986)
//   uf_leff_bwd           gradient of y  = x1 + DropPath(LeFF(LN2(x1)))                  (model.py:987, :666-685)
//   uf_lewin_block_bwd    both, sharing one recomputation
//   uf_downsample_bwd     Conv2d k4 s2 p1 on the token layout                            (model.py:728-746)
//   uf_upsample_cat_bwd   ConvTranspose2d k2 s2 writing the first half of the concat     (model.py:749-771, :1288)
// A block keeps ONLY its f32 input during the forward; everything else is recomputed here from that input with the op-level
// forward kernels (the training forms that write pre-activation AND activation in one pass), then differentiated.  All
// intermediates live in the caller's workspace (bump-allocated, buffers reused as soon as their last reader has run); nothing is
// allocated, nothing syncs, every launch goes to the caller's stream.  Parameter gradients are OVERWRITTEN, in the layouts of
// the reference's parameters (uf_block_grads).  Sums over tokens are the two-stage fixed-order reductions of the op kernels, so a
// block's gradients are bit-reproducible run to run.
#include <stdlib.h>

#include "uf_internal.h"

namespace uf {
namespace {

struct Bump {
    char* base; size_t off;
    template <typename U> U* take(size_t n_bytes) {
        off = align_up(off, 256);
        U* p = base ? reinterpret_cast<U*>(base + off) : nullptr;
        off += n_bytes;
        return p;
    }
};

struct BlockPlan {
    // recomputed forward tensors and backward operands (T = operand type; f32 where noted).  Nothing the weight-gradient jobs read
    // is overwritten later in the block, so they may run on a side stream while the data-gradient chain goes on.
    void *xn, *q, *k, *vt, *o;                // T[M][C] (q, k, vt per head)
    void *tA, *dyw, *dxn, *z, *tE;            // T[M][C]: tA: yw -> dyT;  tE: dz -> dO
    void *a1, *h1, *c, *g2, *dc, *da1;        // T[M][4C]
    void* dqkv;                               // T[M][3C]
    float *x1, *fB;                           // f32[M][C]: x1;  fB: dx1ln -> dx1 (in place)
    float *dbias, *dw9, *zero;                // f32[heads*4096], f32[9][4C] tap-major dwconv gradient, f32[4C] zeros (bias of the input-gradient GEMMs)
    void *scratch, *scratch_w; size_t scratch_bytes, scratch_w_bytes;   // workspaces of the op kernels: data-gradient chain / weight-gradient jobs
    size_t total;
};

void op_scratch_bytes(int B, int H, int W, int C, int heads, uf_dtype dtype, size_t* main_bytes, size_t* side_bytes) {
    const int M = B * H * W;
    size_t s = 0;
    auto mx = [&](size_t v) { if (v > s) s = v; };
    mx(uf_layernorm_bwd_workspace_bytes(M, C));
    mx(uf_window_attention_bwd_workspace_bytes(M / 64, heads));
    mx(uf_dwconv3x3_bwd_workspace_bytes(4 * C, dtype));
    *main_bytes = s;
    s = 0;
    mx(uf_linear_wgrad_workspace_bytes(M, C, 4 * C));
    mx(uf_linear_wgrad_workspace_bytes(M, 4 * C, C));
    mx(uf_linear_wgrad_workspace_bytes(M, C, C));
    mx(uf_linear_wgrad_workspace_bytes(M, 3 * C, C));
    mx(uf_dwconv3x3_wgrad_workspace_bytes(4 * C, dtype));
    mx(uf_rows_sum_workspace_bytes(M / 64, 64 * C));
    *side_bytes = s;
}

BlockPlan plan_block(void* ws, int B, int H, int W, int C, int heads, uf_dtype dtype) {
    const size_t M = (size_t)B * H * W, sz = dtype_size(dtype);
    Bump b{(char*)ws, 0};
    BlockPlan p{};
    p.xn = b.take<void>(M * C * sz);  p.q = b.take<void>(M * C * sz);  p.k = b.take<void>(M * C * sz);  p.vt = b.take<void>(M * C * sz);
    p.o = b.take<void>(M * C * sz);
    p.tA = b.take<void>(M * C * sz);  p.dyw = b.take<void>(M * C * sz); p.dxn = b.take<void>(M * C * sz); p.z = b.take<void>(M * C * sz); p.tE = b.take<void>(M * C * sz);
    p.a1 = b.take<void>(M * 4 * C * sz); p.h1 = b.take<void>(M * 4 * C * sz); p.c = b.take<void>(M * 4 * C * sz);
    p.g2 = b.take<void>(M * 4 * C * sz); p.dc = b.take<void>(M * 4 * C * sz); p.da1 = b.take<void>(M * 4 * C * sz);
    p.dqkv = b.take<void>(M * 3 * C * sz);
    p.x1 = b.take<float>(M * C * 4); p.fB = b.take<float>(M * C * 4);
    p.dbias = b.take<float>((size_t)heads * 4096 * 4);
    p.dw9 = b.take<float>((size_t)9 * 4 * C * 4);
    p.zero = b.take<float>((size_t)4 * C * 4);
    op_scratch_bytes(B, H, W, C, heads, dtype, &p.scratch_bytes, &p.scratch_w_bytes);
    p.scratch = b.take<void>(p.scratch_bytes);
    p.scratch_w = b.take<void>(p.scratch_w_bytes);
    p.total = align_up(b.off, 256);
    return p;
}

// Two in-order queues of one block backward.  `main` = the caller's stream: recomputation and the data-gradient chain (every kernel
// the next one waits for).  `side` = a stream of the device's lane pool for the WEIGHT-gradient jobs (token-split GEMMs and their
// second stages, depthwise-tap gradient, modulator / bias-table sums): nothing downstream reads their results, so they fill the
// gaps of the chain instead of lengthening it.  A job is forked after the kernel that produces its last operand (event recorded on
// main, waited on by side); the block joins side back into main before it returns (the next block reuses the workspace).
// side == main (UF_BWD_STREAMS=1, the two half entry points, or no lane available): plain in-order execution, same results.
struct Queues {
    void* main; void* side; Lane* ln; int next_ev;
    int fork() {                                     // side waits for everything enqueued on main so far
        if (side == main) return UF_OK;
        hipEvent_t e = ln->ev[next_ev++ % MAX_LANE_EVENTS];
        if (hipEventRecord(e, (hipStream_t)main) != hipSuccess || hipStreamWaitEvent((hipStream_t)side, e, 0) != hipSuccess) {
            set_error("block backward: event record / wait failed");
            return UF_ERR_LAUNCH;
        }
        return UF_OK;
    }
    int join() {                                     // main waits for everything enqueued on side so far
        if (side == main) return UF_OK;
        if (hipEventRecord(ln->join[0], (hipStream_t)side) != hipSuccess || hipStreamWaitEvent((hipStream_t)main, ln->join[0], 0) != hipSuccess) {
            set_error("block backward: event record / wait failed");
            return UF_ERR_LAUNCH;
        }
        return UF_OK;
    }
};

// (9, N) tap-major -> (N, 9): the layout of mlp.dwconv.0.weight (4C,1,3,3)
__global__ __launch_bounds__(256) void taps_to_param_kernel(const float* __restrict__ dw9, float* __restrict__ out, int N) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 9 * N) out[i] = dw9[(size_t)(i % 9) * N + i / 9];
}

#define UF_TRY(call)            \
    do {                        \
        const int rc_ = (call); \
        if (rc_) return rc_;    \
    } while (0)

int check_common(const char* fn, const uf_block_train_params* p, const uf_block_grads* g, int B, int H, int W, int C, uf_dtype dtype, const void* ws, size_t ws_bytes) {
    UF_REQUIRE(p && g && ws, UF_ERR_NULL, "%s: null pointer", fn);
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "%s: dtype %d", fn, (int)dtype);
    UF_REQUIRE(B > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0 && C >= 32 && C % 32 == 0 && p->heads * 32 == C, UF_ERR_SHAPE,
               "%s: B=%d H=%d W=%d C=%d heads=%d (H, W multiples of 8; head_dim 32)", fn, B, H, W, C, p->heads);
    UF_REQUIRE(p->shift == 0 || p->shift == 4, UF_ERR_UNSUPPORTED, "%s: shift %d", fn, p->shift);
    UF_REQUIRE((long long)B * H * W * 4 * C < 0x7fffffffLL, UF_ERR_SHAPE, "%s: B*H*W*4C exceeds 32-bit indexing: split the batch", fn);
    const size_t need = uf_lewin_block_bwd_workspace_bytes(B, H, W, C, p->heads, dtype);
    UF_REQUIRE(ws_bytes >= need, UF_ERR_WORKSPACE, "%s: workspace too small: %zu < %zu", fn, ws_bytes, need);
    UF_REQUIRE(((uintptr_t)ws % 256) == 0, UF_ERR_ALIGN, "%s: workspace must be 256-byte aligned", fn);
    return UF_OK;
}

// ---- recomputation ---------------------------------------------------------------------------------------------------------
// attention half: xn, q, k, vt, o (window order) and, when x1 != NULL, x1 = x + drop * window_reverse(proj(o))
int recompute_attn(const uf_block_train_params* p, const BlockPlan& pl, const float* x, const float* drop_attn, bool want_x1, int B, int H, int W, int C,
                   uf_dtype dtype, void* st) {
    const int M = B * H * W;
    UF_TRY(uf_layernorm_fwd(x, C, p->norm1_w, p->norm1_b, p->modulator, pl.xn, B, H, W, C, 1, p->shift, dtype, st));
    UF_TRY(uf_qkv_fwd(pl.xn, p->wqkv, p->bqkv, pl.q, pl.k, pl.vt, M, C, p->heads, dtype, st));
    UF_TRY(uf_window_attention_fwd(pl.q, pl.k, pl.vt, p->rpb_dense, nullptr, 0, pl.o, M / 64, p->heads, 32, H, W, p->shift, dtype, st));
    if (want_x1)     // x1 = x + drop * window_reverse(proj(o)): residual, DropPath and un-partition in the projection GEMM's store
        UF_TRY(uf_linear_residual_fwd(pl.o, p->wproj, p->bproj, x, pl.x1, drop_attn, B, H, W, C, C, 1, p->shift, dtype, st));
    return UF_OK;
}

// LeFF half up to the second GELU (linear2's output is not needed by the backward): z, a1, h1, c, g2
int recompute_leff(const uf_block_train_params* p, const BlockPlan& pl, const float* x1, int B, int H, int W, int C, uf_dtype dtype, void* st) {
    const int M = B * H * W;
    UF_TRY(uf_layernorm_fwd(x1, C, p->norm2_w, p->norm2_b, nullptr, pl.z, B, H, W, C, 0, 0, dtype, st));
    // linear1 keeps only its pre-activation; the stencil activates it as it loads it (the fused depthwise backward recomputes the
    // activation from it)
    UF_TRY(uf_linear_fwd(pl.z, p->w1, p->b1, pl.a1, M, 4 * C, C, 0, dtype, st));
    UF_TRY(uf_dwconv3x3_gelu_in_pre_gelu_fwd(pl.a1, p->wdw9, p->bdw, pl.c, pl.g2, B, H, W, 4 * C, dtype, st));
    return UF_OK;
}

// ---- backward --------------------------------------------------------------------------------------------------------------
// LeFF half: dy -> fB = LN2-path gradient wrt x1 (WITHOUT the residual dy), parameter gradients.
// fork_out != NULL: fB = that gradient PLUS dy (the residual path) and fork_out = T(fB * fork_scale) in window order, both written by the LN2 backward
// kernel (uf_layernorm_bwd_cast) -- the block backward's next step, which uf_grad_fork made in a pass of its own.
int backward_leff(const uf_block_train_params* p, const BlockPlan& pl, const float* x1, const float* dy, const float* drop_leff, const uf_block_grads* g,
                  int B, int H, int W, int C, uf_dtype dtype, Queues& qs, void* fork_out = nullptr, const float* fork_scale = nullptr, float* sum_out = nullptr) {
    const int M = B * H * W, C4 = 4 * C;
    void *st = qs.main, *sw = qs.side;
    UF_TRY(uf_grad_fork(dy, nullptr, nullptr, pl.tA, drop_leff, B, H, W, C, 0, 0, dtype, st));                          // dyT = T(dy * drop)
    UF_TRY(qs.fork());
    UF_TRY(uf_linear_wgrad(pl.tA, C, pl.g2, C4, g->w2, g->b2, M, C, C4, dtype, pl.scratch_w, pl.scratch_w_bytes, sw));
    UF_TRY(uf_linear_mul_dgelu(pl.tA, p->w2_t, pl.zero, pl.c, pl.dc, M, C4, C, dtype, st));                               // dc = (dyT W2) GELU'(c)
    // da1 and the tap / bias gradients in one pass over dc (h1 recomputed from a1); the taps come out on the main queue
    UF_TRY(uf_dwconv3x3_bwd(pl.dc, p->wdw9_flip, pl.a1, pl.da1, pl.dw9, g->bdw, B, H, W, C4, dtype, pl.scratch, pl.scratch_bytes, st));
    hipLaunchKernelGGL(taps_to_param_kernel, dim3((9 * C4 + 255) / 256), dim3(256), 0, (hipStream_t)st, (const float*)pl.dw9, g->wdw, C4);
    UF_TRY(check_launch("taps_to_param"));
    UF_TRY(qs.fork());
    UF_TRY(uf_linear_wgrad(pl.da1, C4, pl.z, C, g->w1, g->b1, M, C4, C, dtype, pl.scratch_w, pl.scratch_w_bytes, sw));
    UF_TRY(uf_linear_fwd(pl.da1, p->w1_t, pl.zero, pl.tE, M, C, C4, 0, dtype, st));                                        // dz
    if (fork_out)
        UF_TRY(uf_layernorm_bwd_cast(x1, C, p->norm2_w, pl.tE, C, 0, dy, pl.fB, C, g->norm2_w, g->norm2_b, B, H, W, C, 0, 0, dtype, fork_out, fork_scale, 1, p->shift, pl.scratch,
                                     pl.scratch_bytes, st));
    else if (sum_out)      // gradient + dy (the residual path) straight into the caller's tensor: the same fused add as the forked form, bit for bit
        UF_TRY(uf_layernorm_bwd_fused(x1, C, p->norm2_w, pl.tE, C, 0, dy, sum_out, C, g->norm2_w, g->norm2_b, B, H, W, C, 0, 0, dtype, pl.scratch, pl.scratch_bytes, st));
    else
        UF_TRY(uf_layernorm_bwd_fused(x1, C, p->norm2_w, pl.tE, C, 0, nullptr, pl.fB, C, g->norm2_w, g->norm2_b, B, H, W, C, 0, 0, dtype, pl.scratch, pl.scratch_bytes, st));
    return UF_OK;
}

// attention half: dyw = T(dx1 * drop) in window order (already there), recomputed xn, q, k, vt, o -> dx = LN1-path gradient + dx1,
// parameter gradients.
int backward_attn(const uf_block_train_params* p, const BlockPlan& pl, const float* x, const float* dx1, float* dx, const uf_block_grads* g,
                  int B, int H, int W, int C, uf_dtype dtype, Queues& qs) {
    const int M = B * H * W;
    void *st = qs.main, *sw = qs.side;
    UF_TRY(qs.fork());                                                                                                     // dyw is ready
    UF_TRY(uf_linear_wgrad(pl.dyw, C, pl.o, C, g->wproj, g->bproj, M, C, C, dtype, pl.scratch_w, pl.scratch_w_bytes, sw));
    UF_TRY(uf_linear_fwd(pl.dyw, p->wproj_t, pl.zero, pl.tE, M, C, C, 0, dtype, st));                                      // dO
    UF_TRY(uf_window_attention_bwd_qkv(pl.q, pl.k, pl.vt, p->rpb_dense, nullptr, 0, pl.tE, C, pl.dqkv, pl.dbias, M / 64, p->heads, 32, H, W, p->shift, dtype,
                                       pl.scratch, pl.scratch_bytes, st));
    UF_TRY(qs.fork());
    UF_TRY(uf_rpb_table_grad(pl.dbias, g->rpb_table, p->heads, sw));
    UF_TRY(uf_linear_wgrad(pl.dqkv, 3 * C, pl.xn, C, g->wqkv, g->bqkv, M, 3 * C, C, dtype, pl.scratch_w, pl.scratch_w_bytes, sw));
    UF_TRY(uf_linear_fwd(pl.dqkv, p->wqkv_t, pl.zero, pl.dxn, M, C, 3 * C, 0, dtype, st));                                 // dxn (window order)
    if (p->modulator) {
        UF_REQUIRE(g->modulator, UF_ERR_NULL, "block backward: the block has a modulator but grads->modulator is NULL");
        UF_TRY(qs.fork());
        UF_TRY(uf_rows_sum(pl.dxn, 64 * C, g->modulator, M / 64, 64 * C, dtype, pl.scratch_w, pl.scratch_w_bytes, sw));
    }
    // LN1 backward reads dxn in window order (window_reverse + roll back folded in) and adds the residual path's gradient
    UF_TRY(uf_layernorm_bwd_fused(x, C, p->norm1_w, pl.dxn, C, 0, dx1, dx, C, g->norm1_w, g->norm1_b, B, H, W, C, 1, p->shift, dtype, pl.scratch, pl.scratch_bytes, st));
    return qs.join();
}

int zero_bias(const BlockPlan& pl, int C, void* st) {
    const hipError_t e = hipMemsetAsync(pl.zero, 0, (size_t)4 * C * sizeof(float), (hipStream_t)st);
    if (e != hipSuccess) { set_error("block backward: hipMemsetAsync: %s", hipGetErrorString(e)); return UF_ERR_LAUNCH; }
    return UF_OK;
}

}  // namespace
}  // namespace uf

using namespace uf;

extern "C" size_t uf_lewin_block_bwd_workspace_bytes(int B, int H, int W, int C, int heads, uf_dtype dtype) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || !dtype_ok(dtype)) return 0;
    return plan_block(nullptr, B, H, W, C, heads, dtype).total;
}

static int bwd_streams() { static const int v = getenv("UF_BWD_STREAMS") ? atoi(getenv("UF_BWD_STREAMS")) : 2; return v; }

extern "C" int uf_lewin_block_bwd(const uf_block_train_params* p, const float* x, const float* dy, float* dx, const float* drop_attn, const float* drop_leff,
                                  const uf_block_grads* g, int B, int H, int W, int C, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_TRY(check_common("uf_lewin_block_bwd", p, g, B, H, W, C, dtype, ws, ws_bytes));
    UF_REQUIRE(x && dy && dx, UF_ERR_NULL, "uf_lewin_block_bwd: null pointer");
    const BlockPlan pl = plan_block(ws, B, H, W, C, p->heads, dtype);
    Queues qs{stream, stream, nullptr, 0};
    int dev = 0;
    Lane* ln = bwd_streams() >= 2 ? acquire_lane(1, &dev) : nullptr;                 // no lane: everything on the caller's stream
    struct Guard { Lane* l; int d; ~Guard() { if (l) release_lane(l, d); } } guard{ln, dev};
    if (ln && lane_events(ln, MAX_LANE_EVENTS)) { qs.side = ln->s[0]; qs.ln = ln; }
    auto run = [&]() -> int {
        UF_TRY(zero_bias(pl, C, stream));
        UF_TRY(recompute_attn(p, pl, x, drop_attn, true, B, H, W, C, dtype, stream));
        UF_TRY(recompute_leff(p, pl, pl.x1, B, H, W, C, dtype, stream));
        constexpr bool fuse_fork = true;
        if (fuse_fork) {   // dx1 = LN2-path gradient + dy -> fB and T(dx1 * drop_attn) in window order -> dyw, from the LN2 backward kernel itself
            UF_TRY(backward_leff(p, pl, pl.x1, dy, drop_leff, g, B, H, W, C, dtype, qs, pl.dyw, drop_attn));
        } else {
            UF_TRY(backward_leff(p, pl, pl.x1, dy, drop_leff, g, B, H, W, C, dtype, qs));
            UF_TRY(uf_grad_fork(pl.fB, dy, pl.fB, pl.dyw, drop_attn, B, H, W, C, 1, p->shift, dtype, stream));
        }
        return backward_attn(p, pl, x, pl.fB, dx, g, B, H, W, C, dtype, qs);
    };
    const int rc = run();
    if (rc != UF_OK) (void)qs.join();      // an error half-way: whatever was already forked must still be ordered before the caller reuses the workspace
    return rc;
}

extern "C" int uf_leff_bwd(const uf_block_train_params* p, const float* x1, const float* dy, float* dx1, const float* drop_leff, const uf_block_grads* g,
                           int B, int H, int W, int C, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_TRY(check_common("uf_leff_bwd", p, g, B, H, W, C, dtype, ws, ws_bytes));
    UF_REQUIRE(x1 && dy && dx1, UF_ERR_NULL, "uf_leff_bwd: null pointer");
    const BlockPlan pl = plan_block(ws, B, H, W, C, p->heads, dtype);
    Queues qs{stream, stream, nullptr, 0};
    UF_TRY(zero_bias(pl, C, stream));
    UF_TRY(recompute_leff(p, pl, x1, B, H, W, C, dtype, stream));
    if (dx1 == dy) {       // in place: the sum goes through the workspace
        UF_TRY(backward_leff(p, pl, x1, dy, drop_leff, g, B, H, W, C, dtype, qs, nullptr, nullptr, pl.fB));
        const hipError_t e = hipMemcpyAsync(dx1, pl.fB, (size_t)B * H * W * C * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e != hipSuccess) { set_error("uf_leff_bwd: hipMemcpyAsync: %s", hipGetErrorString(e)); return UF_ERR_LAUNCH; }
        return UF_OK;
    }
    return backward_leff(p, pl, x1, dy, drop_leff, g, B, H, W, C, dtype, qs, nullptr, nullptr, dx1);                       // + the residual path, in the LN2 backward's store
}

extern "C" int uf_lewin_attn_bwd(const uf_block_train_params* p, const float* x, const float* dx1, float* dx, const float* drop_attn, const uf_block_grads* g,
                                 int B, int H, int W, int C, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_TRY(check_common("uf_lewin_attn_bwd", p, g, B, H, W, C, dtype, ws, ws_bytes));
    UF_REQUIRE(x && dx1 && dx, UF_ERR_NULL, "uf_lewin_attn_bwd: null pointer");
    const BlockPlan pl = plan_block(ws, B, H, W, C, p->heads, dtype);
    Queues qs{stream, stream, nullptr, 0};
    UF_TRY(zero_bias(pl, C, stream));
    UF_TRY(recompute_attn(p, pl, x, drop_attn, false, B, H, W, C, dtype, stream));
    UF_TRY(uf_grad_fork(dx1, nullptr, nullptr, pl.dyw, drop_attn, B, H, W, C, 1, p->shift, dtype, stream));
    return backward_attn(p, pl, x, dx1, dx, g, B, H, W, C, dtype, qs);
}

// ---- samplers --------------------------------------------------------------------------------------------------------------
// Downsample = Conv2d(Cin -> Cout, k4 s2 p1) on token rows.  x f32[B*H*W][ld_x]; dy f32[B*(H/2)*(W/2)][Cout]; w_pk T[Cout][16 Cin] with
// k = (ky*4+kx)*Cin + c (the forward's packing) and w_pk_t its transpose T[16 Cin][Cout].  dx f32 rows of stride ld_dx (accumulate = 1
// adds: the skip connection's gradient is already there); dW_pk f32[Cout][16 Cin] in the PACKED order; db f32[Cout].
extern "C" size_t uf_downsample_bwd_workspace_bytes(int B, int H, int W, int Cin, int Cout, uf_dtype dtype) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const size_t Mo = (size_t)B * (H / 2) * (W / 2), sz = dtype_size(dtype), K = 16 * (size_t)Cin;
    Bump b{nullptr, 0};
    b.take<void>(Mo * K * sz); b.take<void>(Mo * K * sz); b.take<void>(Mo * Cout * sz); b.take<float>(K * 4);
    b.take<void>(uf_linear_wgrad_workspace_bytes((int)Mo, Cout, (int)K));
    return align_up(b.off, 256);
}

extern "C" int uf_downsample_bwd(const float* x, int ld_x, const float* dy, const void* w_pk_t, float* dx, int ld_dx, int accumulate, float* dW_pk, float* db,
                                 int B, int H, int W, int Cin, int Cout, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(x && dy && w_pk_t && dx && dW_pk && db && ws, UF_ERR_NULL, "uf_downsample_bwd: null pointer");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "uf_downsample_bwd: dtype %d", (int)dtype);
    UF_REQUIRE(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && Cin % 8 == 0 && Cout % 8 == 0 && ld_x >= Cin && ld_dx >= Cin, UF_ERR_SHAPE,
               "uf_downsample_bwd: B=%d H=%d W=%d Cin=%d Cout=%d", B, H, W, Cin, Cout);
    const size_t need = uf_downsample_bwd_workspace_bytes(B, H, W, Cin, Cout, dtype);
    UF_REQUIRE(ws_bytes >= need && ((uintptr_t)ws % 256) == 0, UF_ERR_WORKSPACE, "uf_downsample_bwd: workspace %zu < %zu (256-byte aligned)", ws_bytes, need);
    const int Mo = B * (H / 2) * (W / 2), K = 16 * Cin;
    const size_t sz = dtype_size(dtype);
    Bump b{(char*)ws, 0};
    void* cols = b.take<void>((size_t)Mo * K * sz);
    void* dcols = b.take<void>((size_t)Mo * K * sz);
    void* dyT = b.take<void>((size_t)Mo * Cout * sz);
    float* zero = b.take<float>((size_t)K * 4);
    const size_t wg_bytes = uf_linear_wgrad_workspace_bytes(Mo, Cout, K);
    void* wg = b.take<void>(wg_bytes);
    if (hipMemsetAsync(zero, 0, (size_t)K * 4, (hipStream_t)stream) != hipSuccess) { set_error("uf_downsample_bwd: hipMemsetAsync failed"); return UF_ERR_LAUNCH; }
    UF_TRY(uf_im2col(x, ld_x, cols, K, B, H, W, Cin, 4, 2, 1, 0, dtype, stream));
    UF_TRY(uf_grad_fork(dy, nullptr, nullptr, dyT, nullptr, 1, 1, Mo, Cout, 0, 0, dtype, stream));                          // cast to the operand type
    UF_TRY(uf_linear_wgrad(dyT, Cout, cols, K, dW_pk, db, Mo, Cout, K, dtype, wg, wg_bytes, stream));
    // input gradient: from an LDS patch of dy where that form is built (round 6: the 16 Cin-wide patch matrix never exists; the four class weights are packed into the
    // space it would take), else the patch-matrix route
    if ((size_t)Mo * K >= (size_t)16 * Cin * Cout) {
        bool done = false;
        UF_TRY(launch_down_dx(dyT, Cout, w_pk_t, dcols, dx, ld_dx, B, H, W, Cin, Cout, accumulate, dtype, (hipStream_t)stream, &done));
        if (done) return UF_OK;
    }
    UF_TRY(uf_linear_fwd(dyT, w_pk_t, zero, dcols, Mo, K, Cout, 0, dtype, stream));
    return uf_col2im(dcols, K, dx, ld_dx, B, H, W, Cin, 4, 2, 1, 0, accumulate, dtype, stream);
}

// Upsample = ConvTranspose2d(Cin -> Cout, k2 s2) whose output is the first Cout channels of the decoder's concat buffer.
// d f32[B*2H*2W][ld_d]: gradient of the concat buffer (its first Cout columns are read); x f32[B*H*W][ld_x]: the layer input;
// w_pk_t T[Cin][4 Cout]: transpose of the forward's packing n = (dy*2+dx)*Cout + co.  dx f32[B*H*W][Cin] OVERWRITTEN;
// dW_pk f32[4 Cout][Cin] in the packed order; db f32[Cout].
extern "C" size_t uf_upsample_cat_bwd_workspace_bytes(int B, int H, int W, int Cin, int Cout, uf_dtype dtype) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const size_t M = (size_t)B * H * W, sz = dtype_size(dtype);
    Bump b{nullptr, 0};
    b.take<void>(M * 4 * Cout * sz); b.take<void>(M * Cin * sz); b.take<void>(M * Cin * sz); b.take<float>((size_t)Cin * 4); b.take<float>((size_t)4 * Cout * 4);
    size_t s = uf_linear_wgrad_workspace_bytes((int)M, 4 * Cout, Cin);
    const size_t r = uf_rows_sum_workspace_bytes(4, Cout);
    b.take<void>(s > r ? s : r);
    return align_up(b.off, 256);
}

extern "C" int uf_upsample_cat_bwd(const float* d, int ld_d, const float* x, int ld_x, const void* w_pk_t, float* dx, float* dW_pk, float* db,
                                   int B, int H, int W, int Cin, int Cout, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(d && x && w_pk_t && dx && dW_pk && db && ws, UF_ERR_NULL, "uf_upsample_cat_bwd: null pointer");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "uf_upsample_cat_bwd: dtype %d", (int)dtype);
    UF_REQUIRE(B > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout % 8 == 0 && ld_d >= Cout && ld_x == Cin, UF_ERR_SHAPE,
               "uf_upsample_cat_bwd: B=%d H=%d W=%d Cin=%d Cout=%d ld_d=%d ld_x=%d (x rows must be dense)", B, H, W, Cin, Cout, ld_d, ld_x);
    const size_t need = uf_upsample_cat_bwd_workspace_bytes(B, H, W, Cin, Cout, dtype);
    UF_REQUIRE(ws_bytes >= need && ((uintptr_t)ws % 256) == 0, UF_ERR_WORKSPACE, "uf_upsample_cat_bwd: workspace %zu < %zu (256-byte aligned)", ws_bytes, need);
    const int M = B * H * W, N4 = 4 * Cout;
    const size_t sz = dtype_size(dtype);
    Bump b{(char*)ws, 0};
    void* d4 = b.take<void>((size_t)M * N4 * sz);
    void* xT = b.take<void>((size_t)M * Cin * sz);
    void* dxT = b.take<void>((size_t)M * Cin * sz);
    float* zero = b.take<float>((size_t)Cin * 4);
    float* db4 = b.take<float>((size_t)N4 * 4);
    size_t s = uf_linear_wgrad_workspace_bytes(M, N4, Cin);
    const size_t r = uf_rows_sum_workspace_bytes(4, Cout);
    if (r > s) s = r;
    void* wg = b.take<void>(s);
    if (hipMemsetAsync(zero, 0, (size_t)Cin * 4, (hipStream_t)stream) != hipSuccess) { set_error("uf_upsample_cat_bwd: hipMemsetAsync failed"); return UF_ERR_LAUNCH; }
    // the 2x2 output pixels of every input pixel gathered into one row: the k2 s2 patch matrix of the output gradient
    UF_TRY(uf_im2col(d, ld_d, d4, N4, B, 2 * H, 2 * W, Cout, 2, 2, 0, 0, dtype, stream));
    UF_TRY(uf_grad_fork(x, nullptr, nullptr, xT, nullptr, 1, 1, M, Cin, 0, 0, dtype, stream));
    UF_TRY(uf_linear_wgrad(d4, N4, xT, Cin, dW_pk, db4, M, N4, Cin, dtype, wg, s, stream));
    UF_TRY(uf_rows_sum(db4, Cout, db, 4, Cout, UF_F32, wg, s, stream));                                                    // bias: sum over the 4 sub-pixels
    UF_TRY(uf_linear_fwd(d4, w_pk_t, zero, dxT, M, Cin, N4, 0, dtype, stream));
    return uf_residual_combine(nullptr, dxT, 0, dx, nullptr, 1, 1, M, Cin, 0, 0, dtype, stream);
}
