// Window attention core (reference WindowAttention.forward, model.py:494-519, without proj).
//
// One wave64 owns one (window, head): S^T = K Q^T (64x64), + relative-position bias,
// + SW-MSA mask evaluated analytically, softmax over keys, O^T = V^T P^T.  Everything stays in
// registers: no LDS, no barriers, scores never reach memory.
//
// Operand trick (MFMA 16x16, D: col = lane&15, row = 4*(lane>>4)+reg):
//   * S^T tile [key-tile kt][query-tile qt] = mma(A = K rows, B = Q rows)  -> a lane holds, for
//     ONE query (lane&15), keys 16*kt + 4*g + r.  The softmax reductions over keys are 16
//     in-lane values + two cross-lane steps (xor 16, 32).
//   * those registers ARE the B operand (col = query, 8 key slots) of O^T = V^T P^T, with the
//     key-slot order {32s+4g+j, 32s+16+4g+j}; the V^T A operand is read in the same slot order
//     (two contiguous 4-key pieces of a V^T row), which is why uf_qkv_fwd stores V transposed.
#include "uf_internal.h"

namespace uf {
namespace {

template <typename T> struct PFrag {  // build the P operand from exp'ed scores (primary: the 2-byte operand types)
    static __device__ __forceinline__ void make(Frag<T>& f, f32x4 a, f32x4 b) {
        f.v = u32x4{pack2<T>(a[0], a[1]), pack2<T>(a[2], a[3]), pack2<T>(b[0], b[1]), pack2<T>(b[2], b[3])};
    }
};
template <> struct PFrag<float> {
    static __device__ __forceinline__ void make(Frag<float>& f, f32x4 a, f32x4 b) { f.lo = a; f.hi = b; }
};

// V^T operand: 4 keys at p0 and 4 keys at p1
template <typename T> __device__ __forceinline__ void load_vt(Frag<T>& f, const T* p0, const T* p1) {
    static_assert(sizeof(T) == 2, "2-byte operand type");
    const u32x2 a = *reinterpret_cast<const u32x2*>(p0);
    const u32x2 b = *reinterpret_cast<const u32x2*>(p1);
    f.v = u32x4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ void load_vt(Frag<float>& f, const float* p0, const float* p1) {
    f.lo = *reinterpret_cast<const f32x4*>(p0);
    f.hi = *reinterpret_cast<const f32x4*>(p1);
}

template <typename T, int HD>
__global__ __launch_bounds__(256, 4) void window_attn_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ vt,
                                                          const float* __restrict__ bias_dense,
                                                          const float* __restrict__ mask, int n_mask, T* __restrict__ out,
                                                          int n_pairs, int heads, int H, int W, int shift) {
    constexpr int DT = HD / 16;  // 16-wide d tiles of the output
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);  // (window, head) pair of this wave
    if (pair >= n_pairs) return;
    const int bw = pair / heads, h = pair - bw * heads;
    const int fr = lane & 15, fg = lane >> 4;
    const size_t base = (size_t)pair * (64 * HD);
    const T* qp = q + base;
    const T* kp = k + base;
    const T* vp = vt + base;

    // ---- S^T = K Q^T ------------------------------------------------------------------
    Frag<T> qf[4], kf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (fg * 8 < HD) {
            load_frag(qf[i], qp + (i * 16 + fr) * HD + fg * 8);
            load_frag(kf[i], kp + (i * 16 + fr) * HD + fg * 8);
        } else {  // head_dim 16: k-slots 16..31 are zero padding
            qf[i].zero();
            kf[i].zero();
        }
    }
    f32x4 s[4][4];  // [kt][qt]
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16(s[kt][qt], kf[kt], qf[qt]);
        }

    // ---- + bias (+ masks) ---------------------------------------------------------------
    const int nWc = W >> 3, nW = (H >> 3) * nWc;
    const int wi = bw % nW;
    const bool last_r = shift > 0 && (wi / nWc) == (H >> 3) - 1;
    const bool last_c = shift > 0 && (wi % nWc) == nWc - 1;
    const float* bh = bias_dense + (size_t)h * 4096;
    const float* mk = mask ? mask + (size_t)(bw % n_mask) * 4096 : nullptr;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int qi = qt * 16 + fr;
        const bool q_lo_y = (qi >> 3) >= 4, q_lo_x = (qi & 7) >= 4;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int k0 = kt * 16 + fg * 4;
            f32x4 v = s[kt][qt] + *reinterpret_cast<const f32x4*>(bh + qi * 64 + k0);
            if (mk) v += *reinterpret_cast<const f32x4*>(mk + qi * 64 + k0);
            // SW-MSA mask, model.py:924-942: -100 where the 9-region ids of query and key differ.
            // Inside one window the id can only differ in the last window row (y>=4 vs y<4) or
            // the last window column (x>=4 vs x<4).  k0..k0+3 share y; x = (k0&7)+j.
            const bool k_lo_y = (k0 >> 3) >= 4;
            const bool dy = last_r && (k_lo_y != q_lo_y);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool k_lo_x = ((k0 & 7) + j) >= 4;
                if (dy || (last_c && (k_lo_x != q_lo_x))) v[j] += -100.0f;
            }
            s[kt][qt] = v;
        }
    }

    // ---- softmax over keys (per query = per lane column) ------------------------------------
    float inv[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[kt][qt][j]);
        mx = red_xor32<RedMax>(red_xor16<RedMax>(mx));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = __expf(s[kt][qt][j] - mx);
                s[kt][qt][j] = e;
                sum += e;
            }
        sum = red_xor32<RedSum>(red_xor16<RedSum>(sum));
        inv[qt] = 1.0f / sum;
    }

    // ---- O^T = V^T P^T -----------------------------------------------------------------------
    f32x4 o[DT][4];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {  // 32 keys per step
        Frag<T> vf[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const T* row = vp + (dt * 16 + fr) * 64 + sk * 32 + fg * 4;
            load_vt(vf[dt], row, row + 16);
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            Frag<T> pf;
            PFrag<T>::make(pf, s[2 * sk][qt], s[2 * sk + 1][qt]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) mma16(o[dt][qt], vf[dt], pf);
        }
    }

    // ---- normalise and store: out[(bw*64 + query)][h*HD + d] (model.py:519 head merge) --------
    const int C = heads * HD;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        T* orow = out + ((size_t)bw * 64 + qt * 16 + fr) * C + h * HD + fg * 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) store4(orow + dt * 16, o[dt][qt] * inv[qt]);
    }
}

__global__ void shift_mask_kernel(float* out, int H, int W, int shift) {
    // out[(wi*64 + qi)*64 + ki] ; region ids as in model.py:927-936
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nWc = W >> 3, nW = (H >> 3) * nWc;
    if (idx >= nW * 4096) return;
    const int ki = idx & 63, qi = (idx >> 6) & 63, wi = idx >> 12;
    auto region = [&](int tkn) {
        const int hh = (wi / nWc) * 8 + (tkn >> 3), ww = (wi % nWc) * 8 + (tkn & 7);
        const int rh = hh < H - 8 ? 0 : (hh < H - shift ? 1 : 2);
        const int rw = ww < W - 8 ? 0 : (ww < W - shift ? 1 : 2);
        return rh * 3 + rw;
    };
    out[idx] = (shift > 0 && region(qi) != region(ki)) ? -100.0f : 0.0f;
}

}  // namespace
}  // namespace uf

extern "C" int uf_window_attention_fwd(const void* q, const void* k, const void* vt, const float* bias_dense,
                                       const float* mask, int n_mask, void* out, int n_windows, int heads,
                                       int head_dim, int H, int W, int shift, uf_dtype dtype, void* stream) {
    using namespace uf;
    UF_REQUIRE(q && k && vt && bias_dense && out, UF_ERR_NULL, "uf_window_attention_fwd: null pointer");
    UF_REQUIRE(n_windows > 0 && heads > 0, UF_ERR_SHAPE, "uf_window_attention_fwd: n_windows=%d heads=%d", n_windows, heads);
    UF_REQUIRE(head_dim == 16 || head_dim == 32, UF_ERR_UNSUPPORTED, "uf_window_attention_fwd: head_dim %d (16 or 32)", head_dim);
    UF_REQUIRE(H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8, UF_ERR_SHAPE, "uf_window_attention_fwd: H=%d W=%d", H, W);
    UF_REQUIRE(shift == 0 || shift == 4, UF_ERR_UNSUPPORTED, "uf_window_attention_fwd: shift %d (0 or 4)", shift);
    UF_REQUIRE(n_windows % ((H / 8) * (W / 8)) == 0, UF_ERR_SHAPE, "uf_window_attention_fwd: n_windows=%d not a multiple of nW", n_windows);
    UF_REQUIRE(!mask || n_mask > 0, UF_ERR_SHAPE, "uf_window_attention_fwd: mask given with n_mask=%d", n_mask);
    const int n_pairs = n_windows * heads;
    dim3 grid((n_pairs + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    const double el = (double)n_pairs * 64 * head_dim;
    char tname[64] = "";
    if (timing_enabled()) snprintf(tname, sizeof(tname), "window_attn_%s %dx%dx%d", dtype_name(dtype), n_windows, heads, head_dim);
    ScopedTimer tm(tname, 4.0 * 64 * el,
                   4.0 * el * dtype_size(dtype), st);
#define UF_ATTN_LAUNCH(TT, HDV)                                                                                  \
    hipLaunchKernelGGL((window_attn_kernel<TT, HDV>), grid, block, 0, st, (const TT*)q, (const TT*)k,            \
                       (const TT*)vt, bias_dense, mask, n_mask, (TT*)out, n_pairs, heads, H, W, shift)
    if (dtype == UF_BF16) {
        if (head_dim == 32) UF_ATTN_LAUNCH(bf16, 32); else UF_ATTN_LAUNCH(bf16, 16);
    } else if (dtype == UF_F16) {
        if (head_dim == 32) UF_ATTN_LAUNCH(f16, 32); else UF_ATTN_LAUNCH(f16, 16);
    } else if (dtype == UF_F32) {
        if (head_dim == 32) UF_ATTN_LAUNCH(float, 32); else UF_ATTN_LAUNCH(float, 16);
    } else {
        set_error("uf_window_attention_fwd: unknown dtype %d", (int)dtype);
        return UF_ERR_UNSUPPORTED;
    }
#undef UF_ATTN_LAUNCH
    return check_launch("window_attention");
}

extern "C" int uf_shift_mask(float* out, int H, int W, int shift, void* stream) {
    using namespace uf;
    UF_REQUIRE(out, UF_ERR_NULL, "uf_shift_mask: null out");
    UF_REQUIRE(H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8, UF_ERR_SHAPE, "uf_shift_mask: H=%d W=%d", H, W);
    const int n = (H / 8) * (W / 8) * 4096;
    hipLaunchKernelGGL(shift_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, H, W, shift);
    return check_launch("shift_mask");
}
