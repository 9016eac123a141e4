// Internal (non-ABI) launch functions shared between the translation units of libuformer_hip.
#pragma once
#include "uf_common.h"

namespace uf {

enum ALoad { A_PLAIN = 0, A_FROM_R = 1, A_CONV_DOWN = 2 };
enum Epi {
    E_STORE_T = 0,      // out T[m][n] = acc + bias
    E_STORE_T_GELU = 1, // out T[m][n] = gelu(acc + bias)
    E_QKV = 2,          // split into q (scaled) / k / v^T per (window, head)
    E_RES_WINREV = 3,   // out f32[tok(m)][n] = resid[tok(m)][n] + acc + bias  (window_reverse + unroll)
    E_RES = 4,          // out f32[m][n] = resid[m][n] + acc + bias
    E_STORE_R = 5,      // out f32[m][n] = acc + bias
    E_UPSAMPLE = 6,     // ConvTranspose2d k2 s2 scatter into (2H,2W)
    E_STORE_T_PRE_GELU = 7,  // out T[m][n] = a = acc + bias AND aux T[m][n] = gelu(a as stored)  (training: linear1 keeps both)
    E_STORE_T_MUL_DGELU = 8  // out T[m][n] = T(acc + bias) * GELU'(aux[m][n])                   (input gradient through a GELU)
};

struct GemmParams {
    const void* A; int lda;     // A_PLAIN: T[M][lda]; A_FROM_R / A_CONV_DOWN: f32 rows of stride lda
    const void* W;              // T[N][K]
    const void* W_fm = nullptr; // Downsample second form: the same weight in the fragment-major layout of uf_pack_weight_fm (optional)
    const float* bias;
    int M, N, K;
    int H, W_, C;               // geometry: conv-down input (H,W,C); winrev (H,W); upsample input (H,W)
    int shift;
    void* out; int ldo;
    const float* resid; int ldr;
    void* q; void* k; void* vt; int heads, hd; float qscale;
    int Cout;
    void* aux;                  // E_STORE_T_PRE_GELU: second output; E_STORE_T_MUL_DGELU: the pre-activation (read); ld = ldo
    const float* scale; int hw; // E_RES / E_RES_WINREV: optional per-image factor of the GEMM branch (DropPath, model.py:986-987), hw = tokens per image
};

// dtype-dispatching launcher; AL/EP are the enums above.  Returns UF_* status.
int launch_gemm(const GemmParams& p, int aload, int epi, uf_dtype dtype, hipStream_t stream);
// input gradient of Downsample from an LDS patch of dy (uf_gemm.hip, round 6); *done = false where the form is not built
int launch_down_dx(const void* dyT, int ld_dy, const void* w_pk_t, void* wc, float* dx, int ld_dx, int B, int H, int W, int Cin, int Cout, int accumulate, uf_dtype dtype,
                   hipStream_t st, bool* done);

// Side streams.  A LANE = {side streams, fork / join events, general-purpose events}; a call owns one lane exclusively between
// acquire_lane and release_lane, so concurrent callers on one device never record or wait on each other's events (uf_core.hip).
constexpr int MAX_SIDE = 7, MAX_LANE_EVENTS = 8;
struct Lane { hipStream_t s[MAX_SIDE] = {}; hipEvent_t fork = nullptr, join[MAX_SIDE] = {}; int n = 0; hipEvent_t ev[MAX_LANE_EVENTS] = {}; int n_ev = 0; };
Lane* acquire_lane(int want, int* dev_out);
void release_lane(Lane* ln, int dev);
bool lane_events(Lane* ln, int want);

// fused attention half (uf_attnblk.hip)
void debug_set_tbuf(void* p);
bool attn_block_supported(const uf_block_params* bp, const float* user_mask, uf_dtype dtype, int C, int heads);
// h1_out != NULL (and dtype bf16): the kernel also writes h1 = GELU(linear1(LN2(x_new))), T[B*H*W][4C]
// drop: per-image DropPath scale of the attention branch (training forward) or NULL
// xo / ldo (optional): write the new rows THERE instead of in place
int launch_attn_block(const uf_block_params* bp, float* x, int ld, int B, int H, int W, int C, uf_dtype dtype, void* h1_out, hipStream_t st,
                      const float* drop = nullptr, float* xo = nullptr, int ldo = 0);
// second half of LeFF with an optional per-image DropPath scale of the branch (uf_leff2.hip)
int launch_leff2(const void* h1, const float* w9, const float* bdw, const void* W2, const float* b2, float* x, int ld, int B, int H, int W, int C,
                 uf_dtype dtype, const float* drop, hipStream_t st);

// training forward of the attention half + linear1: the fused kernel with side stores of what the backward reads (uf_attnblk.hip)
int launch_attn_block_train(const uf_block_params* bp, const float* x, int ld, float* x1, int ld1, int B, int H, int W, int C, uf_dtype dtype, const float* drop,
                            void* xn, void* q, void* k, void* vt, void* o, void* z, void* a1, hipStream_t st);

}  // namespace uf
