/* libuformer_hip -- C ABI of the MI355X (gfx950) Uformer LeWin-block hot path.
 *
 * The reference (ZhendongWang6/Uformer) has no FFI / operator API: its de-facto boundary is
 * the nn.Module surface of model.py.  This header is what a binding for that path would bind:
 * one entry point per reference function on the hot path, plus the
 * whole-block and whole-model drivers.  Each declaration cites the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates,
 *     frees or retains memory; `stream` is a hipStream_t passed as void* (NULL = default
 *     stream); all work is enqueued on it, nothing synchronises.
 *   - return value: 0 = UF_OK, negative = error; uf_last_error() gives the text (thread local).
 *     Nothing throws or aborts across this boundary.
 *   - tokens are channel-last:  x[b][h][w][c]  ==  the reference's (B, L=H*W, C) layout.
 *   - `dtype` (uf_dtype) is the type T of GEMM operands and of intermediate activations
 *     (UF_F32: exact-f32 MFMA; UF_BF16: bf16 operands, f32 accumulate; UF_F16: IEEE half operands, f32
 *     accumulate -- the reference's own reduced-precision mode (torch.cuda.amp autocast is fp16,
 *     train/train_denoise.py:164,180-184): the full-speed mode that meets the 1e-3 output tolerance).
 *     The residual stream, LayerNorm/softmax/GELU math, biases and all small tables are f32.
 *   - residual-stream tensors carry a row stride `ld` (elements) so that an encoder stage can
 *     live inside the second half of a decoder concat buffer (model.py:1288 torch.cat).
 *   - window size is 8 (every shipped arch, utils/model_utils.py:65-78); H, W multiples of 8.
 */
#ifndef UFORMER_HIP_H
#define UFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UF_ABI_VERSION 1

typedef enum { UF_F32 = 0, UF_BF16 = 1, UF_F16 = 2 } uf_dtype;

#define UF_OK 0
#define UF_ERR_SHAPE (-1)
#define UF_ERR_UNSUPPORTED (-2)
#define UF_ERR_ALIGN (-3)
#define UF_ERR_LAUNCH (-4)
#define UF_ERR_WORKSPACE (-5)
#define UF_ERR_NULL (-6)

int uf_version(void);
/* copies the calling thread's last error text (NUL terminated) into buf; returns its length */
int uf_last_error(char* buf, size_t n);

/* ---- diagnostics (not part of the reference surface) --------------------------------------
 * uf_timing_enable(1): every kernel launch is bracketed by HIP events on its own stream and its
 * algorithmic flops/bytes are booked per kernel class.  uf_timing_report() waits for the events,
 * writes a JSON array [{"kernel","launches","ms","flops","bytes"},...] and clears the log;
 * returns the length the full text needs.  bench.py uses it for the live roofline figure. */
int uf_timing_enable(int on);
int uf_timing_report(char* json, size_t n);
/* development aid: device buffer of u64 that instrumented kernels fill -- sampled phase stamps in entries [0, 65536),
 * then one 8-entry census record per workgroup {s_memtime start, end, 100 MHz clock start, end, HW_ID|XCC_ID<<32}: the
 * buffer must hold 65536 + 8 * (largest grid) entries.  NULL = off (the default). */
int uf_debug_set_tbuf(void* p);

/* ---- fragment-major weights ------------------------------------------------------------------
 * The fused kernels stream nn.Linear weights W[N][K] straight from L2 into MFMA operand registers.
 * They take W re-laid out so that one wave-level load is 1 KiB contiguous:
 *     out[((n/16 * KS + k/32) * 64 + ((k%32)/8)*16 + n%16) * 8 + k%8] = W[n][k],   KS = ceil(K/32),
 * zero padded in k.  Arguments named *_fm below are in this layout (uf_weight_fm_elems elements). */
size_t uf_weight_fm_elems(int N, int K);
int uf_pack_weight_fm(const void* w_rowmajor, void* out_fm, int N, int K, uf_dtype dtype, void* stream);

/* ---- a1-a4: index-only ops (bit exact) -------------------------------------------------- */
/* torch.roll(x,(-shift,-shift)) + window_partition: model.py:957, :704-715.
 * x (B,H,W,C) -> out (B*nW, 8, 8, C); elem_bytes in {2,4}; pure copy. */
int uf_window_partition(const void* x, void* out, int B, int H, int W, int C, int shift,
                        int elem_bytes, void* stream);
/* window_reverse + torch.roll(+shift): model.py:717-726, :980. */
int uf_window_reverse(const void* windows, void* out, int B, int H, int W, int C, int shift,
                      int elem_bytes, void* stream);
/* SW-MSA mask (nW,64,64) f32 in {0,-100}: model.py:924-942.  The attention kernel evaluates
 * the same predicate in registers; this entry materialises it for the bit-exact test. */
int uf_shift_mask(float* out, int H, int W, int shift, void* stream);

/* ---- a5/a6: LayerNorm (+ roll + partition + modulator) ---------------------------------- */
/* out[m] = LN(x[src(m)]) * gamma + beta (+ modulator[m % 64]);  eps 1e-5, biased variance.
 * windowed != 0: src(m) = roll/partition index (norm1 path, model.py:952-969);
 * windowed == 0: src(m) = m (norm2 path, model.py:987).  x f32 [rows][ld_x]; out T [rows][C]. */
int uf_layernorm_fwd(const float* x, int ld_x, const float* gamma, const float* beta,
                     const float* modulator /* (64,C) or NULL */, void* out, int B, int H, int W,
                     int C, int windowed, int shift, uf_dtype dtype, void* stream);

/* ---- a7/a9/a10: nn.Linear  out = act(A @ W^T + bias)  (model.py:426-427,489,657,661) ------
 * A T[M][K], W T[N][K] (nn.Linear layout), bias f32[N], out T[M][N]; act 0=none 1=erf-GELU. */
int uf_linear_fwd(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                  int act, uf_dtype dtype, void* stream);

/* ---- a7: fused Q/K/V projection (LinearProjection.forward, model.py:431-442) --------------
 * A T[M][C] window-order tokens; Wqkv T[3C][C] = cat(to_q.weight, to_kv.weight); bias f32[3C].
 * Writes q T[M/64][heads][64][hd] (already multiplied by hd^-0.5, model.py:497),
 *        k T[M/64][heads][64][hd],  vt T[M/64][heads][hd][64] (V transposed per window/head). */
int uf_qkv_fwd(const void* A, const void* Wqkv, const float* bqkv, void* q, void* k, void* vt,
               int M, int C, int heads, uf_dtype dtype, void* stream);

/* ---- a5+a6+a7 fused: LN1 -> roll -> window_partition -> +modulator -> Q/K/V projection -------
 * (model.py:952-969 then :431-442, :497).  x f32 rows (stride ld) of the (B,H,W,C) stream; outputs as
 * uf_qkv_fwd.  The normalised activations stay in LDS; x is read once. */
int uf_ln_qkv_fwd(const float* x, int ld, const float* gamma, const float* beta,
                  const float* modulator /* (64,C) or NULL */, const void* Wqkv_fm, const float* bqkv,
                  void* q, void* k, void* vt, int B, int H, int W, int C, int heads, int shift,
                  uf_dtype dtype, void* stream);
/* ---- a5+a10 fused: LN2 -> linear1 -> GELU (model.py:987, :657-658).  x f32 [M] rows (stride ld),
 * W1_fm = fragment-major T[N][C], b1 f32[N], out T[M][N]. */
int uf_ln_linear_gelu_fwd(const float* x, int ld, const float* gamma, const float* beta,
                          const void* W1_fm, const float* b1, void* out, int M, int N, int C,
                          uf_dtype dtype, void* stream);

/* ---- a8: window attention core (WindowAttention.forward model.py:494-519, without proj) ---
 * q,k,vt as produced by uf_qkv_fwd.  bias_dense f32[heads][64][64] =
 * relative_position_bias_table[relative_position_index] permuted (model.py:500-502).
 * shift>0 adds the SW-MSA mask analytically (window position from H,W);  mask (optional,
 * f32 [n_mask][64][64], row = window index % n_mask, model.py:508-512) is added on top.
 * out T[M][C], channel = head*hd + d (model.py:519). */
int uf_window_attention_fwd(const void* q, const void* k, const void* vt, const float* bias_dense,
                            const float* mask, int n_mask, void* out, int n_windows, int heads,
                            int head_dim, int H, int W, int shift, uf_dtype dtype, void* stream);

/* ---- a10: depthwise 3x3 + bias + GELU on token layout (LeFF dwconv, model.py:659-660) -----
 * x,out T[B][H][W][C]; w9 f32[9][C] (tap-major repack of (C,1,3,3)); bias f32[C]. */
int uf_dwconv3x3_gelu_fwd(const void* x, const float* w9, const float* bias, void* out, int B,
                          int H, int W, int C, uf_dtype dtype, void* stream);
/* the same stencil with the activation optional (gelu = 0: bias may be NULL).  With the taps flipped (w9[8 - t]) and
 * gelu = 0 it is the INPUT gradient of the depthwise conv: dh[y,x] = sum w[ky,kx] dc[y-ky+1, x-kx+1]. */
int uf_dwconv3x3_fwd(const void* x, const float* w9, const float* bias, void* out, int B, int H,
                     int W, int C, int gelu, uf_dtype dtype, void* stream);

/* training forms of the stencil (uformer_amd/train.py; model.py:657-660 and its backward):
 *   uf_dwconv3x3_pre_gelu_fwd: pre_out = stencil + bias AND act_out = GELU(pre_out as stored), one pass (the backward needs both);
 *   uf_dwconv3x3_mul_dgelu: out = stencil(dy; flipped taps, no bias) as stored * GELU'(pre): the gradient through the depthwise
 *     conv AND the GELU in front of it (uf_dwconv3x3_fwd + uf_gelu_bwd in one pass). */
int uf_dwconv3x3_pre_gelu_fwd(const void* x, const float* w9, const float* bias, void* pre_out, void* act_out, int B, int H,
                              int W, int C, uf_dtype dtype, void* stream);
/* uf_dwconv3x3_pre_gelu_fwd on x = GELU(pre_in): pre_in T[B][H][W][C] is the PRE-activation of the GELU in front of the convolution (linear1's
 * output, model.py:657-658); the kernel activates it as it loads it, rounded to T as a stored activation is -- bit-identical to
 * uf_dwconv3x3_pre_gelu_fwd on the activation uf_linear_pre_gelu_fwd writes, which the training forward then need not store. */
int uf_dwconv3x3_gelu_in_pre_gelu_fwd(const void* pre_in, const float* w9, const float* bias, void* pre_out, void* act_out, int B, int H,
                                      int W, int C, uf_dtype dtype, void* stream);
int uf_dwconv3x3_mul_dgelu(const void* dy, const float* w9_flipped, const void* pre, void* out, int B, int H, int W, int C,
                           uf_dtype dtype, void* stream);

/* ---- a10 fused: x += linear2(GELU(dwconv3x3(h1)))  (LeFF second half, model.py:674-683, :987) ----
 * h1 T[B][H][W][4C] = GELU(linear1(LN2(x))); w9 f32[9][4C]; bdw f32[4C]; W2_fm fragment-major T[C][4C]; b2 f32[C];
 * x f32 rows of C (stride ld), updated in place.  The conv output never reaches HBM. */
int uf_dwconv_linear2_fwd(const void* h1, const float* w9, const float* bdw, const void* W2_fm,
                          const float* b2, float* x, int ld, int B, int H, int W, int C,
                          uf_dtype dtype, void* stream);

/* ---- per-block parameters (packed; SURVEY.md Appendix C names in comments) ----------------- */
typedef struct uf_block_params {
    const float* norm1_w;   /* norm1.weight (C) */
    const float* norm1_b;   /* norm1.bias */
    const float* modulator; /* modulator.weight (64,C) or NULL */
    const float* rpb_dense; /* (heads,64,64) gathered from attn.relative_position_bias_table */
    const float* rpb_fm;    /* reserved (was a fragment-major copy of rpb_dense); ignored, may be NULL */
    const float* rpb_tab;   /* (heads,15,15) compact table [dy+7][7-dx] when the bias is Toeplitz (always, for the
                               reference's relative_position_index), else NULL -> the 3-kernel path with rpb_dense */
    const void* wqkv_fm;    /* T (3C,C) fragment-major: attn.qkv.to_q.weight ; attn.qkv.to_kv.weight */
    const float* bqkv;      /* (3C) */
    const void* wproj;      /* T (C,C) attn.proj.weight, row-major (3-kernel fallback path) */
    const void* wproj_fm;   /* T (C,C) fragment-major */
    const float* bproj;     /* (C) */
    const float* norm2_w;
    const float* norm2_b;
    const void* w1_fm;      /* T (4C,C) fragment-major mlp.linear1.0.weight */
    const float* b1;        /* (4C) */
    const float* wdw9;      /* (9,4C) tap-major repack of mlp.dwconv.0.weight (4C,1,3,3) */
    const float* bdw;       /* (4C) */
    const void* w2_fm;      /* T (C,4C) fragment-major mlp.linear2.0.weight */
    const float* b2;        /* (C) */
    int32_t shift;          /* 0 or 4, decided at construction (model.py:1030, :863-866) */
    int32_t heads;
} uf_block_params;

/* bytes of scratch a block needs for M = B*H*W tokens of width C */
size_t uf_block_workspace_bytes(int M, int C, uf_dtype dtype);

/* ---- a11: attention half of LeWinTransformerBlock.forward (model.py:951-986) --------------
 * x = x + proj(attn(partition(roll(LN1(x))) + modulator)) ; in place on the f32 stream. */
int uf_lewin_attn_fwd(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C,
                      const float* user_mask, int n_mask, uf_dtype dtype, void* ws,
                      size_t ws_bytes, void* stream);
/* ---- a10/a11: FFN half (model.py:987): x = x + LeFF(LN2(x)) ; in place. */
int uf_leff_fwd(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C,
                uf_dtype dtype, void* ws, size_t ws_bytes, void* stream);
/* ---- a15, training forward of the attention half + linear1 (round 6; 2-byte operands, head_dim 32): the fused window kernel of
 * uf_lewin_attn_fwd with SIDE STORES of every operand the backward reads, so that the kept-intermediates training forward
 * (train/train_denoise.py:180-184 over model.py:951-987, :657-658) is one launch instead of six:
 *   x1 [M][ld1] f32 = x + drop_attn[b] * proj(attention(...))   (out of place: x stays for the LayerNorm backward)
 *   xn, o  T[M][C] in window-row order; q (times head_dim^-0.5), k T[nW][heads][64][32]; vt T[nW][heads][32][64]   (= uf_ln_qkv_fwd / uf_window_attention_fwd)
 *   z = LN2(x1) T[M][C] and a1 = linear1(z) + b1 T[M][4C] (PRE-activation; the stencil applies GELU as it loads it) in token order.
 * drop_attn: per-image DropPath scales (B) or NULL. */
int uf_lewin_attn_train_fwd(const uf_block_params* p, const float* x, int ld, float* x1, int ld1, int B, int H, int W, int C,
                            const float* drop_attn, uf_dtype dtype, void* xn, void* q, void* k, void* vt, void* o, void* z, void* a1,
                            void* stream);
/* whole block = the two halves */
int uf_lewin_block_fwd(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C,
                       const float* user_mask, int n_mask, uf_dtype dtype, void* ws,
                       size_t ws_bytes, void* stream);

/* ---- a11 in train() mode: the same two fused kernels with timm's DropPath applied to each residual branch:
 * x = x + drop_attn[b] * attention_branch;  x = x + drop_leff[b] * LeFF_branch  (model.py:986-987; scales = bernoulli(keep)/keep
 * per image, f32[B] on the device, NULL = 1).  The caller keeps a copy of x before the call: it is all the backward needs
 * (everything else is recomputed, uformer_amd/train.py). */
int uf_lewin_block_train_fwd(const uf_block_params* p, float* x, int ld, int B, int H, int W, int C,
                             const float* drop_attn, const float* drop_leff, uf_dtype dtype, void* ws,
                             size_t ws_bytes, void* stream);

/* ---- a12: Downsample.forward (Conv2d k4 s2 p1 on tokens, model.py:739-746) -----------------
 * x f32[B][H][W] rows of C (stride ld_x); w T[2C][16C] with k = (ky*4+kx)*C + c; out f32
 * [B][H/2][W/2] rows of 2C (stride ld_o). */
int uf_downsample_fwd(const float* x, int ld_x, const void* w, const float* bias, float* out,
                      int ld_o, int B, int H, int W, int C, uf_dtype dtype, void* stream);
/* The same with the weight ALSO in the fragment-major layout of uf_pack_weight_fm(w, w_fm, 2C, 16C) (round 6): where the second form of the kernel is built
 * (2-byte operand types, C = 32 / 64 / 128 / 256, output maps that are whole tiles) it streams w_fm -- contiguous KiB per wave-instruction instead of 16 rows x 64
 * bytes -- and is bit-identical to uf_downsample_fwd; everywhere else, and with w_fm = NULL, it IS uf_downsample_fwd. */
int uf_downsample_fm_fwd(const float* x, int ld_x, const void* w, const void* w_fm, const float* bias, float* out,
                         int ld_o, int B, int H, int W, int C, uf_dtype dtype, void* stream);
/* ---- a13: Upsample.forward (ConvTranspose2d k2 s2, model.py:765-771) written straight into
 * the concat buffer: out rows (stride ld_o) at [B][2H][2W], channels [0,Cout).
 * w T[4*Cout][Cin], n = (dy*2+dx)*Cout + co. */
int uf_upsample_fwd(const float* x, int ld_x, const void* w, const float* bias, float* out,
                    int ld_o, int B, int H, int W, int Cin, int Cout, uf_dtype dtype, void* stream);
/* ---- a14: InputProj (conv3x3 + LeakyReLU(0.01), model.py:795-800): img f32 NCHW (B,Cin,H,W)
 * -> tokens f32 rows of E (stride ld_o); w27 f32[Cin*9][E] ((ci,ky,kx)-major repack). */
int uf_input_proj_fwd(const float* img, const float* w27, const float* bias, float* out, int ld_o,
                      int B, int Cin, int H, int W, int E, void* stream);
/* ---- a14: OutputProj + global residual (model.py:828-836, :1305): tokens f32 rows of C2
 * (stride ld_x) -> out f32 NCHW (B,3,H,W) = conv3x3(tokens) + bias (+ img if add_img).
 * w f32[3][9][C2] ((co,ky,kx)-major repack of (3,C2,3,3)). */
int uf_output_proj_fwd(const float* x, int ld_x, const float* w, const float* bias,
                       const float* img, float* out, int B, int H, int W, int C2, int add_img,
                       void* stream);

/* ---- a15: backward building blocks (autograd of a5 / a10 in the reference; closed forms in
 * oracle/uformer_oracle_bwd.py, which is pinned to the reference's autograd).  First members of the family; the
 * rest of the backward path is not built yet.  Token reductions are two-stage through the caller's workspace
 * (bit-reproducible, no atomics). -------------------------------------------------------------------------- */
/* dx = dy * GELU'(a), erf form (nn.GELU, model.py:657-660).  a, dy, dx: T[n], n multiple of 16 bytes / sizeof(T) */
int uf_gelu_bwd(const void* a, const void* dy, void* dx, long long n, uf_dtype dtype, void* stream);
/* y = GELU(a) as a separate pass, for a training forward that keeps the pre-activation (same flavour as the fused epilogues) */
int uf_gelu_fwd(const void* a, void* y, long long n, uf_dtype dtype, void* stream);
/* training forms of uf_linear_fwd (uformer_amd/train.py):
 *   uf_linear_pre_gelu_fwd: out = a = A W^T + bias AND act_out = GELU(a as stored): linear1 of the LeFF in one pass (model.py:657-658);
 *   uf_linear_mul_dgelu: out = T(A W^T + bias) * GELU'(pre[m][n]): the input gradient of a Linear whose input came out of a GELU
 *     (A = dy, W = the layer's weight transposed, bias = zeros): uf_linear_fwd + uf_gelu_bwd in one pass. */
int uf_linear_pre_gelu_fwd(const void* A, const void* W, const float* bias, void* out, void* act_out, int M, int N, int K,
                           uf_dtype dtype, void* stream);
int uf_linear_mul_dgelu(const void* A, const void* W, const float* bias, const void* pre, void* out, int M, int N, int K,
                        uf_dtype dtype, void* stream);
/* out f32[tok][N] = resid[tok][N] + scale[image of tok] * (A W^T + bias); tok = the token of window row m when `windowed` (window_reverse +
 * roll back by `shift`), else m.  The attention projection / linear2 of a block with `x + drop_path(...)` (model.py:975-987) in the GEMM's
 * store: scale f32[B] = bernoulli(keep) / keep per sample, or NULL.  A T[B*H*W][K], W T[N][K]; resid may alias out. */
int uf_linear_residual_fwd(const void* A, const void* W, const float* bias, const float* resid, float* out, const float* scale,
                           int B, int H, int Wd, int N, int K, int windowed, int shift, uf_dtype dtype, void* stream);
/* nn.LayerNorm(C) backward over rows of the f32 stream: dx f32[rows][ld_dx]; dgamma, dbeta f32[C] are OVERWRITTEN
 * with the sums over all rows.  C in {16,32,64,128,256,512,1024}. */
size_t uf_layernorm_bwd_workspace_bytes(int rows, int C);
int uf_layernorm_bwd(const float* x, int ld_x, const float* gamma, const float* dy, int ld_dy, float* dx, int ld_dx,
                     float* dgamma, float* dbeta, int rows, int C, void* ws, size_t ws_bytes, void* stream);
/* the same backward reading the output gradient where the block backward has it: dy of the operand type (dy_is_f32 = 1: f32), in WINDOW
 * order when `windowed` (x, add and dx rows are then the tokens of those window rows: window_reverse + roll back folded in), plus an
 * optional second gradient `add` f32 (stride ld_dx) summed into dx (the residual path).  Same workspace. */
int uf_layernorm_bwd_fused(const float* x, int ld_x, const float* gamma, const void* dy, int ld_dy, int dy_is_f32,
                           const float* add, float* dx, int ld_dx, float* dgamma, float* dbeta, int B, int H, int W, int C,
                           int windowed, int shift, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream);
/* uf_layernorm_bwd_fused with a second output: cast_out T[B*H*W][C] = T(dx * cast_scale[image]) (cast_scale f32[B] or NULL) written at the
 * token's own row, or at its window-order row when cast_windowed (roll by -cast_shift + window_partition, model.py:957-963) -- the GEMM
 * operand the next step of the backward reads (the block's attention branch after LN2's backward, the preceding block's LeFF branch after
 * LN1's), which uf_grad_fork otherwise makes in a pass of its own.  dy of the operand type (dy_is_f32 = 0, or dtype f32).  dx, dgamma,
 * dbeta are bit-identical to uf_layernorm_bwd_fused; cast_out is bit-identical to uf_grad_fork of the dx this call wrote.  Same workspace. */
int uf_layernorm_bwd_cast(const float* x, int ld_x, const float* gamma, const void* dy, int ld_dy, int dy_is_f32,
                          const float* add, float* dx, int ld_dx, float* dgamma, float* dbeta, int B, int H, int W, int C,
                          int windowed, int shift, uf_dtype dtype, void* cast_out, const float* cast_scale, int cast_windowed, int cast_shift,
                          void* ws, size_t ws_bytes, void* stream);
/* nn.Linear weight / bias gradients: dW f32[N][K] = sum_m dY[m][n] X[m][k], db f32[N] = sum_m dY[m][n] (db may be NULL),
 * OVERWRITTEN.  dY T[M][ldy] (N columns), X T[M][ldx] (K columns); N, K, ldy, ldx multiples of 16 bytes / sizeof(T).
 * (The input gradient dX = dY W is uf_linear_fwd with the transposed weight.) */
size_t uf_linear_wgrad_workspace_bytes(int M, int N, int K);
int uf_linear_wgrad(const void* dY, int ldy, const void* X, int ldx, float* dW, float* db, int M, int N, int K,
                    uf_dtype dtype, void* ws, size_t ws_bytes, void* stream);
/* Window attention backward (a8, model.py:494-519 without the projections), everything recomputed from the forward
 * operands of uf_window_attention_fwd: q (scaled), k T[n_windows*heads][64][hd], vt T[..][hd][64], bias_dense, mask.
 * dO T[n_windows*64][ldo] is the gradient of the merged-head output.  dq, dk, dvt: gradients wrt q (as stored, i.e.
 * scaled), k, vt, same layouts; dbias f32[heads][64][64] = dS summed over all windows (OVERWRITTEN; scatter-add it
 * through relative_position_index for the table gradient).  head_dim 32. */
size_t uf_window_attention_bwd_workspace_bytes(int n_windows, int heads);
int uf_window_attention_bwd(const void* q, const void* k, const void* vt, const float* bias_dense, const float* mask,
                            int n_mask, const void* dO, int ldo, void* dq, void* dk, void* dvt, float* dbias,
                            int n_windows, int heads, int head_dim, int H, int W, int shift, uf_dtype dtype,
                            void* ws, size_t ws_bytes, void* stream);
/* the same backward with ONE output: dqkv T[n_windows*64][3C], the gradient of the fused q|k|v projection output in window-row
 * order (heads merged; the q third already multiplied by head_dim^-0.5) -- the operand of the projection's weight / input gradients. */
int uf_window_attention_bwd_qkv(const void* q, const void* k, const void* vt, const float* bias_dense, const float* mask,
                                int n_mask, const void* dO, int ldo, void* dqkv, float* dbias, int n_windows, int heads,
                                int head_dim, int H, int W, int shift, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream);
/* depthwise 3x3 tap / bias gradients: dw9 f32[9][C] (tap-major, like w9), dbias f32[C], OVERWRITTEN;
 * h (the conv input) and dc (gradient of the conv output, before the bias): T[B][H][W][C], H multiple of 4 */
size_t uf_dwconv3x3_wgrad_workspace_bytes(int C, uf_dtype dtype);
int uf_dwconv3x3_wgrad(const void* h, const void* dc, float* dw9, float* dbias, int B, int H, int W, int C,
                       uf_dtype dtype, void* ws, size_t ws_bytes, void* stream);
/* the whole backward of the LeFF's depthwise conv behind its GELU in ONE pass over dc (model.py:657-660, autograd of
 * `self.dwconv(x)` after `self.linear1(x)`): da T[B][H][W][C] = what uf_dwconv3x3_mul_dgelu(dc, w9_flipped, pre) writes (bit-identical; f32: to an ulp),
 * dw9 / dbias = what uf_dwconv3x3_wgrad(h = T(GELU(pre)), dc) computes (another summation order: equal to rounding) -- the conv input
 * h is recomputed from the pre-activation instead of being read, dc is read once.  Tensors under 4 GiB, H multiple of 4. */
size_t uf_dwconv3x3_bwd_workspace_bytes(int C, uf_dtype dtype);
int uf_dwconv3x3_bwd(const void* dc, const float* w9_flipped, const void* pre, void* da, float* dw9, float* dbias,
                     int B, int H, int W, int C, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream);

/* ---- a15: reductions and patch gathers of the backward (no ATen glue) -------------------------------------------------------
 * out f32[N] = sum over the M rows of X T[M][ld]: the modulator gradient (model.py:966-969: the (64,C) table is added to every
 * window, so its gradient is the sum of d(xn) over windows: M = n_windows, N = 64*C) and bias-like column sums.  Fixed order. */
size_t uf_rows_sum_workspace_bytes(int M, int N);
int uf_rows_sum(const void* X, int ld, float* out, int M, int N, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream);
/* relative_position_bias_table gradient f32[225][heads] from the dense bias gradient f32[heads][64][64] that
 * uf_window_attention_bwd returns: the transpose of the gather table[index] (model.py:500-502), as a gather over the pairs
 * of each table entry instead of a scatter-add with atomics. */
int uf_rpb_table_grad(const float* dbias_dense, float* dtable, int heads, void* stream);
/* patch matrices of the strided convolutions (Downsample k4 s2 p1, stem / head k3 s1 p1; model.py:734,785,817) for their
 * weight and input gradients through uf_linear_wgrad / uf_linear_fwd:
 *   cols T[B*Ho*Wo][ldc], column (ky*k + kx)*Cin + c = x[b][oy*s + ky - pad][ox*s + kx - pad][c], zero outside and in the
 *   padding columns; x f32 token rows (stride ld_x) or, nchw = 1, (B,Cin,H,W) planes.
 *   uf_col2im is the transpose (gather form, fixed order): dx (+)= sum of the matching dcols entries; accumulate = 1 adds to dx. */
int uf_im2col(const float* x, int ld_x, void* cols, int ldc, int B, int H, int W, int Cin, int k, int stride, int pad,
              int nchw, uf_dtype dtype, void* stream);
int uf_col2im(const void* dcols, int ldc, float* dx, int ld_dx, int B, int H, int W, int Cin, int k, int stride, int pad,
              int nchw, int accumulate, uf_dtype dtype, void* stream);

/* backward of the two full-resolution 3x3 stride-1 pad-1 convolutions, all f32 (InputProj 3 -> E with its LeakyReLU, model.py:771-800;
 * OutputProj 2E -> 3, model.py:803-827).  Two forms: (token rows x, Cin % 4 == 0, Cout <= 4) and (x_nchw = 1: NCHW image, Cin <= 4,
 * Cout 16, 32 or 64).  dy f32[B*H*W][Cout] token rows; act_out (InputProj form only, or NULL): the layer's stored OUTPUT rows, the
 * gradient is first multiplied by LeakyReLU'(slope) taken from its sign; w (Cout,Cin,3,3) as in the state_dict.
 * dx (NULL to skip) in the layout of x; dW (Cout,Cin,3,3); db (Cout).  Sums over pixels: per-block partials added in block order. */
size_t uf_conv3x3_bwd_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int uf_conv3x3_bwd(const float* x, int x_nchw, const float* dy, const float* act_out, float slope, const float* w, float* dx,
                   float* dW, float* db, int B, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* streaming helpers of the block's recompute / backward; each replaces several elementwise passes (cast, DropPath scale, window
 * permutation, residual add, head merge).  Rows of C channels, C % 8 == 0; scale = f32[B] per-image DropPath scale or NULL.
 *   uf_residual_combine: out[tok] = (a ? a[tok] : 0) + scale[img] * b[row]; b is T (or f32 when b_is_f32) in WINDOW order when
 *     `windowed` (window_reverse + roll back, model.py:975-980), else raster; a, out f32 raster.  (x1 = x + DropPath(attn branch).)
 *   uf_grad_fork: t = g1[tok] (+ g2[tok]); sum_out[tok] = t (optional); cast_out[row] = T(t * scale[img]), row = window order when
 *     `windowed` (roll + window_partition, model.py:957-963).  (gradient entering a branch, cast to the GEMM operand type.)
 *   uf_qkv_grad_merge: dqkv T[n_windows*64][3C] from uf_window_attention_bwd's dq (x head_dim^-0.5), dk, dvt (head merge). */
int uf_residual_combine(const float* a, const void* b, int b_is_f32, float* out, const float* scale, int B, int H, int W,
                        int C, int windowed, int shift, uf_dtype dtype, void* stream);
int uf_grad_fork(const float* g1, const float* g2, float* sum_out, void* cast_out, const float* scale, int B, int H, int W,
                 int C, int windowed, int shift, uf_dtype dtype, void* stream);
int uf_qkv_grad_merge(const void* dq, const void* dk, const void* dvt, void* dqkv, int n_windows, int heads, int head_dim,
                      uf_dtype dtype, void* stream);

/* ---- a15: block-level backward = recomputation + gradients (SURVEY 8(b) export list) --------------------------------------------
 * The training forward (uf_lewin_block_train_fwd) keeps only a block's f32 INPUT; these entry points rebuild the intermediates
 * from it with the op-level forward kernels and differentiate them (csrc/uf_trainblk.hip: host composition, every launch on the
 * caller's stream, all temporaries in the caller's workspace).  Operands as a training step packs them per step: */
typedef struct uf_block_train_params {
    const float* norm1_w;  const float* norm1_b;  const float* norm2_w;  const float* norm2_b;   /* (C) each */
    const float* modulator;      /* (64,C) or NULL */
    const float* rpb_dense;      /* (heads,64,64) = relative_position_bias_table[relative_position_index] */
    const void* wqkv;            /* T (3C,C) row-major: cat(attn.qkv.to_q.weight, attn.qkv.to_kv.weight) */
    const void* wqkv_t;          /* T (C,3C): its transpose (input-gradient GEMM) */
    const float* bqkv;           /* (3C) */
    const void* wproj;  const void* wproj_t;  const float* bproj;      /* T (C,C), transpose, (C) */
    const void* w1;     const void* w1_t;     const float* b1;         /* T (4C,C), T (C,4C), (4C) */
    const float* wdw9;  const float* wdw9_flip;  const float* bdw;     /* (9,4C) taps, the same with the tap axis reversed, (4C) */
    const void* w2_t;            /* T (4C,C): mlp.linear2.0.weight transposed */
    int32_t shift, heads;
    const void* w2;              /* T (C,4C): mlp.linear2.0.weight as is -- read only by callers that run the op-by-op forward with this pack
                                  * (uformer_amd/train.py, stored-intermediates form); the block backward entry points do not touch it */
} uf_block_train_params;
/* f32 outputs, OVERWRITTEN, in the layouts of the reference's parameters */
typedef struct uf_block_grads {
    float* norm1_w;  float* norm1_b;  float* norm2_w;  float* norm2_b;
    float* modulator;            /* (64,C); NULL iff the block has no modulator */
    float* rpb_table;            /* (225, heads) attn.relative_position_bias_table */
    float* wqkv;  float* bqkv;   /* (3C,C), (3C): rows [0,C) = to_q, [C,3C) = to_kv */
    float* wproj; float* bproj;
    float* w1;    float* b1;
    float* wdw;   float* bdw;    /* (4C,1,3,3), (4C) */
    float* w2;    float* b2;     /* (C,4C), (C) */
} uf_block_grads;
/* Per-step packing of one block for training: the reference's f32 parameter tensors (device pointers; they must stay unchanged
 * until the block's backward has run) -> every operand layout the fused forward (uf_block_params) and the block-level backward
 * (uf_block_train_params) read, written into ONE caller buffer in 5 launches.  The small f32 tensors (LayerNorm, biases, modulator)
 * are referenced in place.  index_is_standard: the caller has verified that relative_position_index is the reference's
 * (model.py:471-481) -- the compact Toeplitz bias table the fused attention kernel wants is only valid then (else rpb_tab = NULL
 * and the forward takes the 3-kernel path with the dense table). */
typedef struct uf_block_raw_params {
    const float* norm1_w;  const float* norm1_b;  const float* norm2_w;  const float* norm2_b;
    const float* modulator;                 /* (64,C) or NULL */
    const float* rpb_table;                 /* (225, heads) */
    const int64_t* rpb_index;               /* (64, 64) */
    const float* to_q_w;   const float* to_q_b;   const float* to_kv_w;  const float* to_kv_b;    /* (C,C) (C) (2C,C) (2C) */
    const float* proj_w;   const float* proj_b;
    const float* lin1_w;   const float* lin1_b;   /* (4C,C) (4C) */
    const float* dw_w;     const float* dw_b;     /* (4C,1,3,3) (4C) */
    const float* lin2_w;   const float* lin2_b;   /* (C,4C) (C) */
    int32_t index_is_standard;
} uf_block_raw_params;
size_t uf_pack_block_train_bytes(int C, int heads, uf_dtype dtype);
int uf_pack_block_train(const uf_block_raw_params* raw, int C, int heads, int shift, uf_dtype dtype, void* buf, size_t buf_bytes,
                        uf_block_params* fwd /* may be NULL */, uf_block_train_params* bwd /* may be NULL */, void* stream);
size_t uf_lewin_block_bwd_workspace_bytes(int B, int H, int W, int C, int heads, uf_dtype dtype);   /* serves all three below */
/* x: the block input f32[B*H*W][C]; dy: gradient of the block output; dx: gradient of the input (may NOT alias); drop_*: the
 * per-image DropPath scales the forward used (f32[B]) or NULL.  model.py:951-987. */
int uf_lewin_block_bwd(const uf_block_train_params* p, const float* x, const float* dy, float* dx, const float* drop_attn,
                       const float* drop_leff, const uf_block_grads* g, int B, int H, int W, int C, uf_dtype dtype,
                       void* ws, size_t ws_bytes, void* stream);
/* the halves: y = x1 + DropPath(LeFF(LN2(x1))) given x1 and dy -> dx1 (g: norm2, w1, b1, wdw, bdw, w2, b2 written);
 * x1 = x + DropPath(attention branch) given x and dx1 -> dx (g: norm1, modulator, rpb_table, wqkv, bqkv, wproj, bproj written). */
int uf_leff_bwd(const uf_block_train_params* p, const float* x1, const float* dy, float* dx1, const float* drop_leff,
                const uf_block_grads* g, int B, int H, int W, int C, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream);
int uf_lewin_attn_bwd(const uf_block_train_params* p, const float* x, const float* dx1, float* dx, const float* drop_attn,
                      const uf_block_grads* g, int B, int H, int W, int C, uf_dtype dtype, void* ws, size_t ws_bytes, void* stream);
/* Downsample backward (Conv2d k4 s2 p1 on token rows, model.py:728-746): x f32[B*H*W][ld_x] the layer input, dy f32[B*H/2*W/2][Cout],
 * w_pk_t T[16 Cin][Cout] = transpose of the forward's packed weight (k = (ky*4+kx)*Cin + c).  dx f32 rows of stride ld_dx
 * (accumulate = 1: added to what is there, e.g. the skip connection's gradient); dW_pk f32[Cout][16 Cin] in the PACKED order
 * (reference layout = reshape (Cout,4,4,Cin) -> permute (0,3,1,2)); db f32[Cout]. */
size_t uf_downsample_bwd_workspace_bytes(int B, int H, int W, int Cin, int Cout, uf_dtype dtype);
int uf_downsample_bwd(const float* x, int ld_x, const float* dy, const void* w_pk_t, float* dx, int ld_dx, int accumulate,
                      float* dW_pk, float* db, int B, int H, int W, int Cin, int Cout, uf_dtype dtype, void* ws,
                      size_t ws_bytes, void* stream);
/* Upsample backward (ConvTranspose2d k2 s2 into the first Cout channels of the concat buffer, model.py:749-771, :1288):
 * d f32[B*2H*2W][ld_d] gradient of the concat buffer (columns [0,Cout) are read; the skip half is columns [Cout, ld_d));
 * x f32[B*H*W][Cin] the layer input (dense rows); w_pk_t T[Cin][4 Cout] = transpose of the forward's packed weight
 * (n = (dy*2+dx)*Cout + co).  dx f32[B*H*W][Cin]; dW_pk f32[4 Cout][Cin] packed order (reference layout (Cin,Cout,2,2) =
 * reshape (2,2,Cout,Cin) -> permute (3,2,0,1)); db f32[Cout]. */
size_t uf_upsample_cat_bwd_workspace_bytes(int B, int H, int W, int Cin, int Cout, uf_dtype dtype);
int uf_upsample_cat_bwd(const float* d, int ld_d, const float* x, int ld_x, const void* w_pk_t, float* dx, float* dW_pk,
                        float* db, int B, int H, int W, int Cin, int Cout, uf_dtype dtype, void* ws, size_t ws_bytes,
                        void* stream);

/* ---- f-2: training-step tail ------------------------------------------------------------------------------
 * CharbonnierLoss.forward + its gradient in one pass (losses.py:41-52; criterion of train/train_denoise.py:164,181):
 *   loss[0] = mean(sqrt((y - target)^2 + eps^2));   dy[i] = grad_scale * (y - target)[i] / sqrt(.) / n   (dy may be NULL)
 * y, target, dy: f32[n] (any shape flattened), 16-byte aligned.  Two-stage fixed-order reduction in double through `ws`
 * (uf_charbonnier_workspace_bytes): bit-reproducible.  The restored image y already contains the global residual
 * (uf_output_proj_fwd adds it, model.py:1305). */
size_t uf_charbonnier_workspace_bytes(long long n);
int uf_charbonnier_fwd_bwd(const float* y, const float* target, float* dy, float* loss, long long n, float eps,
                           float grad_scale, void* ws, size_t ws_bytes, void* stream);
/* torch.optim.AdamW(lr, betas, eps, weight_decay) (train/train_denoise.py:77: 2e-4, (0.9,0.999), 1e-8, 0.02), one step for
 * n_tensors parameters in a handful of launches (40 tensors per launch, arguments by value: nothing is allocated or retained).
 * params / grads / exp_avg / exp_avg_sq / numel are HOST arrays of device pointers / element counts; state is f32.  `step`
 * counts from 1 (bias correction); gradients are multiplied by grad_scale first (1 / world_size folds the all-reduce average).
 *   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * (hyper-parameters are doubles and the derived scalars are rounded to f32 once, exactly as torch rounds its Python scalars) */
int uf_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  const long long* numel, int n_tensors, double lr, double beta1, double beta2, double eps, double weight_decay,
                  int step, double grad_scale, void* stream);

/* f16 training under a DYNAMIC loss scale, entirely on the device: torch.cuda.amp.GradScaler semantics as the reference uses them through timm's
 * NativeScaler (train/train_denoise.py:42, :180-184: scale the loss, unscale + inf check, skip the optimizer step on overflow, grow / back off).
 * scaler_state: device float[8] = {scale, 1 / scale, found_inf, growth tracker, optimizer steps taken, 0, 0, 0}; the caller initialises
 * {S, 1 / S, 0, 0, 0} (GradScaler: S = 65536) and multiplies the loss by S (reading state[0] on the device, or knowing S on the host).
 *   uf_grad_scaler_check   state[2] = 1 if any element of the gradients is inf / nan (never cleared here)
 *   uf_adamw_step_scaled   uf_adamw_step with gradients x grad_scale x state[1], NOTHING written when state[2] != 0, and the bias corrections taken
 *                          at step state[4] + 1 -- a skipped step does not advance the optimizer, as GradScaler.step never calls optimizer.step then
 *   uf_grad_scaler_update  GradScaler.update(): overflow -> scale *= backoff_factor, tracker = 0; else state[4] += 1 and after growth_interval clean
 *                          steps in a row scale *= growth_factor; refreshes 1 / scale and clears found_inf.  No host synchronisation anywhere. */
int uf_grad_scaler_check(const float* const* grads, const long long* numel, int n_tensors, float* scaler_state, void* stream);
int uf_adamw_step_scaled(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                         const long long* numel, int n_tensors, double lr, double beta1, double beta2, double eps, double weight_decay,
                         double grad_scale, const float* scaler_state, void* stream);
int uf_grad_scaler_update(float* scaler_state, double growth_factor, double backoff_factor, int growth_interval, void* stream);

/* ---- f-3: evaluation metrics on the device ------------------------------------------------------------------------
 * per-image mean squared difference of (optionally [0,1]-clamped) images: myPSNR = 20 log10(1 / sqrt(mse)) and batch_PSNR
 * (utils/image_utils.py:40-51) follow from it without copying an image to the host.  a, b: f32 (n_images, C, H, W). */
size_t uf_image_metric_workspace_bytes(int n_images, int C, int H, int W);
int uf_batch_mse(const float* a, const float* b, float* mse_per_image, int n_images, int C, int H, int W, int clamp01,
                 void* ws, size_t ws_bytes, void* stream);
/* calculate_ssim (utils/caculate_psnr_ssim.py:35-81): images quantised to uint8 levels (x255, round, clamp), 11x11 Gaussian
 * sigma 1.5, "valid" region, mean over the map and over channels; one value per image.  H, W > 10. */
int uf_batch_ssim(const float* a, const float* b, float* ssim_per_image, int n_images, int C, int H, int W, void* ws,
                  size_t ws_bytes, void* stream);

/* ---- f-1: arbitrary-resolution wrapper of the evaluation scripts ----------------------------------------------------
 * expand2square (test/test_sidd.py:79-92): canvas f32 (B,C,X,X) = 0 with the (B,C,h,w) image at ((X-h)/2, (X-w)/2); mask
 * (B,1,X,X) = 1 over the image (may be NULL).  One pass, no memset.  uf_crop_clamp is the way back (masked_select + clamp,
 * test/test_sidd.py:108-109). */
int uf_expand2square(const float* img, float* canvas, float* mask, int B, int C, int h, int w, int X, void* stream);
int uf_crop_clamp(const float* canvas, float* out, int B, int C, int h, int w, int X, int clamp01, void* stream);

/* ---- f-4: training input pipeline on the device ---------------------------------------------------------------------
 * DataLoaderTrain.__getitem__ (dataset/dataset_denoise.py:42-73) for a whole batch: out[b] = T_k(frame[idx][:, r0:r0+ps,
 * c0:c0+ps]) with T_k = Augment_RGB_torch.transform<k> (utils/dataset_utils.py:5-33), k in 0..7.  src: N frames, uint8
 * (divided by 255 like load_img) or f32, layout (N,H,W,3) (src_hwc = 1) or (N,3,H,W); meta: int32 [B][4] = {idx, r0, c0, k}
 * on the device; out f32 (B,3,ps,ps).  Call it twice with the same meta for the clean / noisy pair. */
int uf_crop_augment(const void* src, int src_is_u8, int src_hwc, float* out, const int* meta, int B, int N, int H, int W,
                    int ps, void* stream);
/* MixUp_AUG.aug (utils/dataset_utils.py:37-53): out[b] = lam[b] x[b] + (1 - lam[b]) x[perm[b]]; out must not alias x. */
int uf_mixup(const float* x, float* out, const float* lam, const int* perm, int B, long long per_sample, void* stream);

/* ---- a9 (boundary): Uformer.forward (model.py:1269-1305) ------------------------------------ */
typedef struct uf_model_desc {
    int32_t embed_dim, dd_in, in_chans;
    int32_t depths[9];
    const uf_block_params* blocks; /* sum(depths) entries, stage-major (enc0..3, conv, dec0..3) */
    const float* in_w27;   /* input_proj.proj.0.weight repacked */
    const float* in_b;
    const float* out_w;    /* output_proj.proj.0.weight repacked */
    const float* out_b;
    const void* down_w[4]; /* dowsample_i.conv.0.weight repacked T[2C][16C] */
    const float* down_b[4];
    const void* up_w[4];   /* upsample_i.deconv.0.weight repacked T[4Cout][Cin] */
    const float* up_b[4];
    const void* down_w_fm[4]; /* (round 6, appended) down_w in the fragment-major layout of uf_pack_weight_fm(down_w, ., 2C, 16C), or NULL: see uf_downsample_fm_fwd */
} uf_model_desc;

size_t uf_uformer_workspace_bytes(const uf_model_desc* d, int B, int H, int W, uf_dtype dtype);
/* img, out: f32 NCHW (B,dd_in,H,W) / (B,in_chans,H,W).  H == W, multiple of 128.
 * All work is ordered on `stream`: for B >= 8 the library cuts the batch in two and runs the second half on an
 * internal side stream forked from / joined back into `stream` with events (kernel ramps and tails of the halves
 * overlap; results are bit-identical); UF_STREAMS=1 in the environment turns that off, UF_STREAMS=n (<= 8) forces n. */
int uf_uformer_fwd(const uf_model_desc* d, const float* img, float* out, int B, int H, int W,
                   uf_dtype dtype, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UFORMER_HIP_H */
