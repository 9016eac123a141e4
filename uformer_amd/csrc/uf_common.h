// Shared device/host helpers for the gfx950 (MI355X, CDNA4) Uformer kernels.
// wave = 64 lanes, MFMA 16x16 tiles, fp32 accumulate everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include "../../include/uformer_hip.h"

namespace uf {

// ------------------------------------------------------------------------------------
// element types.  T = operand/activation type (bf16, f16 or f32), R = residual stream = f32.
// f16 is the reference's own reduced-precision mode (torch.cuda.amp autocast, train/train_denoise.py:180-184): same MFMA
// rate and fragment layout as bf16 on gfx950, three more mantissa bits.
// ------------------------------------------------------------------------------------
struct bf16 { uint16_t v; };
struct f16 { uint16_t v; };
typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 mfma_f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// float -> bf16 through the compiler's own conversion: on gfx950 it lowers to ONE v_cvt_pk_bf16_f32
// per two values (round-to-nearest-even), instead of ~8 VALU ops per value of integer rounding.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f)); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const bf16x2_t v = {static_cast<__bf16>(lo), static_cast<__bf16>(hi)};
    return __builtin_bit_cast(uint32_t, v);
}

// float <-> f16: round-to-nearest-even conversions through the compiler (v_cvt_f16_f32 / v_cvt_pk_f16_f32, v_cvt_f32_f16)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float h2f(uint16_t h) { return static_cast<float>(__builtin_bit_cast(_Float16, h)); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, static_cast<_Float16>(f)); }
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
    const f16x2_t v = {static_cast<_Float16>(lo), static_cast<_Float16>(hi)};
    return __builtin_bit_cast(uint32_t, v);
}
// the 2-byte operand types behind one interface: pack2<T>(lo, hi) -> one dword, unpack2<T>(w, lo, hi), cvt1<T> / ld1<T>
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16>(float lo, float hi) { return pack2bf(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16>(float lo, float hi) { return pack2h(lo, hi); }
template <typename T> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16>(uint32_t w, float& lo, float& hi) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
template <> __device__ __forceinline__ void unpack2<f16>(uint32_t w, float& lo, float& hi) {
    const f16x2_t v = __builtin_bit_cast(f16x2_t, w);
    lo = static_cast<float>(v[0]); hi = static_cast<float>(v[1]);
}
template <typename T> __device__ __forceinline__ u32x4 pack8(const float* f) {
    return u32x4{pack2<T>(f[0], f[1]), pack2<T>(f[2], f[3]), pack2<T>(f[4], f[5]), pack2<T>(f[6], f[7])};
}
template <typename T> __device__ __forceinline__ void unpack8(u32x4 r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2<T>(r[i], f[2 * i], f[2 * i + 1]);
}
template <typename T> struct TypeName;
template <> struct TypeName<bf16> { static constexpr const char* s = "bf16"; };
template <> struct TypeName<f16> { static constexpr const char* s = "f16"; };
template <> struct TypeName<float> { static constexpr const char* s = "f32"; };

// exact erf GELU (nn.GELU default; reference model.py:657-660)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GELU for results that are STORED AS bf16: x * sigmoid(2*sqrt(2/pi)*(x + 0.044715 x^3)) = x / (1 + 2^(x*(A + B*x^2))),
// two values per call so the plain ops become v_pk_mul/v_pk_fma_f32 (6 packed-pair VALU + 2x(v_exp,v_rcp) instead of
// 2x(14 VALU + 2 transcendental) for an Abramowitz-Stegun erf, ~35 with branches for erff).  |tanh form - erf form| <= 4.8e-4 (at |x| ~ 2.7, where bf16's
// half-ulp is 7.8e-3): 16x under the storage rounding; the whole Uformer-B output moves by 5.8e-5 (95.9 dB), measured
// on the oracle, against 2e-3 / 67 dB for bf16 operands themselves.  The f32 (parity) mode never uses it.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#ifndef UF_GELU_POLY
#define UF_GELU_POLY 0
#endif
__device__ __forceinline__ f32x2_t gelu_bf2(f32x2_t x) {
#if UF_GELU_POLY
    // FMA-only form: Phi(x) ~ 0.5 + t P(t^2), t = clamp(x, -4, 4), P of degree 6 (weighted minimax fit of (Phi - 0.5) / x for the
    // error of x Phi): |GELU error| <= 1.9e-4 evaluated in f32 (the sigmoid form: 4.7e-4), and no transcendental -- v_exp / v_rcp
    // issue at a quarter of the FMA rate and cannot be packed, the 11 instructions here are 2 v_med3 + 9 packed ops per PAIR.
    const f32x2_t t = f32x2_t{__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f)};
    const f32x2_t u = t * t;
    f32x2_t q = u * 2.2783789077607253e-08f + (-1.5987216102075763e-06f);
    q = q * u + 4.79578593512997e-05f;
    q = q * u + (-0.0008140378049574792f);
    q = q * u + 0.00877249427139759f;
    q = q * u + (-0.06457333266735077f);
    q = q * u + 0.39788350462913513f;
    return x * (t * q + 0.5f);
#else
    constexpr float A = -2.3022081985f;    // -2*sqrt(2/pi)*log2(e)
    constexpr float B = -0.10294324f;      // A * 0.044715
    const f32x2_t u = x * (x * x * B + A);
    const f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + 1.0f;
    return x * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
#endif
}
// Which GELU the operand type T gets.  bf16: the packed sigmoid form above.  f16: the same form by default -- the whole Uformer-B output
// moves by 5.8e-5 with it (oracle/bf16_budget.py row "sigmoid-form GELU"), a twentieth of the 1e-3 output tolerance, while an
// erf-accurate form doubles the VALU work of every activation (two GELUs per hidden channel and token); UF_F16_GELU_ERF=1 builds the
// f16 kernels with the Abramowitz-Stegun erf form below instead (|GELU error| < 5e-7) for A/B runs.  f32: erff.
#ifndef UF_F16_GELU_ERF
#define UF_F16_GELU_ERF 0
#endif
template <typename T> struct GeluKind { static constexpr int v = 0; };                          // 0: erff (f32 parity mode)
template <> struct GeluKind<bf16> { static constexpr int v = 1; };                              // 1: packed sigmoid form
template <> struct GeluKind<f16> { static constexpr int v = UF_F16_GELU_ERF ? 2 : 1; };        // 2: A-S 7.1.26 erf
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): GELU(x) = max(x, 0) - 0.5 |x| P(t) exp(-z^2), z = |x| / sqrt(2), t = 1 / (1 + p z)
__device__ __forceinline__ float gelu_as(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f) * t;
    const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    return fmaf(-0.70710678118654752440f * z, q * e, fmaxf(x, 0.0f));
}
// in-place GELU of N (even) values in the flavour the operand type T calls for
template <typename T, int N> __device__ __forceinline__ void gelu_n(float* v) {
    if constexpr (GeluKind<T>::v == 2) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = gelu_as(v[i]);
    } else if constexpr (GeluKind<T>::v == 1) {
#pragma unroll
        for (int i = 0; i < N; i += 2) {
            const f32x2_t g = gelu_bf2(f32x2_t{v[i], v[i + 1]});
            v[i] = g[0]; v[i + 1] = g[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = gelu_erf(v[i]);
    }
}
// GELU'(a).  f32 operands: Phi(a) + a phi(a) of nn.GELU's erf form.  bf16 operands: the exact derivative of gelu_bf2 (the function the
// bf16 forward evaluates), s + a e s^2 (2k + 6k 0.044715 a^2) with e = exp(-2u), s = 1 / (1 + e): one exp2 and one rcp instead of
// erff + expf.  It differs from the erf form by < 8.7e-4 absolute, under half a bf16 ulp of the O(1) result it is rounded to.
template <typename T> __device__ __forceinline__ float gelu_grad_t(float a) {
    if constexpr (GeluKind<T>::v == 1) {
        constexpr float A = -2.3022081985f, B = -0.10294324f;                 // as gelu_bf2
        const float u = fminf(a * (a * a * B + A), 80.0f);                   // e s^2 stays finite for very negative a
        const float e = __builtin_amdgcn_exp2f(u);
        const float sg = __builtin_amdgcn_rcpf(e + 1.0f);
        return sg + a * e * sg * sg * (1.5957691216f + 0.2140610f * a * a);
    } else {
        return 0.5f * (1.0f + erff(a * 0.70710678118654752440f)) + a * __expf(-0.5f * a * a) * 0.39894228040143267794f;
    }
}

// GELU(a) and GELU'(a) together (round 6: the fused depthwise backward needs both for every element -- h1 = GELU(a1) for the tap gradients, GELU'(a1) for
// the input gradient).  2-byte operand types: ONE exp2 / rcp pair serves both (the two separate calls evaluated the same e = 2^u and 1 / (1 + e) twice: 4
// quarter-rate instructions and 3 multiplies per element more).  dg is gelu_grad_t<T>(a) bit for bit; g is gelu_n's value for every a > -8.4 (below that the
// clamp of u that keeps dg finite makes g = a 2^-80 instead of a / (1 + 2^u): both are zero to 23 decimal places).  Other types: the two calls.
#ifndef UF_GELU_PAIR
#define UF_GELU_PAIR 1      // 0: the two separate evaluations (A/B builds: scripts/build_variant.sh)
#endif
template <typename T> __device__ __forceinline__ void gelu_and_grad_t(float a, float& g, float& dg) {
    if constexpr (GeluKind<T>::v == 1 && UF_GELU_PAIR) {
        constexpr float A = -2.3022081985f, B = -0.10294324f;                 // as gelu_bf2
        const float u = fminf(a * (a * a * B + A), 80.0f);
        const float e = __builtin_amdgcn_exp2f(u);
        const float sg = __builtin_amdgcn_rcpf(e + 1.0f);
        g = a * sg;
        dg = sg + a * e * sg * sg * (1.5957691216f + 0.2140610f * a * a);
    } else {
        float t[2] = {a, a};
        gelu_n<T, 2>(t);
        g = t[0];
        dg = gelu_grad_t<T>(a);
    }
}

template <typename T> __device__ __forceinline__ void gelu4(f32x4& v) {
    float t[4] = {v[0], v[1], v[2], v[3]};
    gelu_n<T, 4>(t);
    v = f32x4{t[0], t[1], t[2], t[3]};
}

// N channels of the operand type as one load / store: 16 bytes (8 x 2-byte, 4 x f32) or 8 bytes (4 x 2-byte)
template <typename T, int N> struct Chunk;
template <typename T> struct Chunk<T, 8> {
    static_assert(sizeof(T) == 2, "8 channels of a 2-byte type");
    using Raw = u32x4;
    static __device__ __forceinline__ void unpack(const Raw& r, float* f) { unpack8<T>(r, f); }
    static __device__ __forceinline__ Raw pack(const float* f) { return pack8<T>(f); }
};
template <typename T> struct Chunk<T, 4> {
    using Raw = typename std::conditional<sizeof(T) == 2, u32x2, u32x4>::type;
    static __device__ __forceinline__ void unpack(const Raw& r, float* f) {
        if constexpr (sizeof(T) == 2) { unpack2<T>(r[0], f[0], f[1]); unpack2<T>(r[1], f[2], f[3]); }
        else { f[0] = __uint_as_float(r[0]); f[1] = __uint_as_float(r[1]); f[2] = __uint_as_float(r[2]); f[3] = __uint_as_float(r[3]); }
    }
    static __device__ __forceinline__ Raw pack(const float* f) {
        if constexpr (sizeof(T) == 2) return Raw{pack2<T>(f[0], f[1]), pack2<T>(f[2], f[3])};
        else return Raw{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
    }
};

// ------------------------------------------------------------------------------------
// debug census (uf_debug_set_tbuf): every workgroup records where and when it ran, so the host can count how many
// workgroups a CU really held at once and the effective shader clock.  Entry b at tbuf[65536 + 8 b]:
// {s_memtime start, end, wall_clock64 (100 MHz) start, end, HW_ID | XCC_ID << 32}.
// ------------------------------------------------------------------------------------
struct Census {
    unsigned long long t0, r0;
    __device__ __forceinline__ void begin() { t0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
    __device__ __forceinline__ void end(unsigned long long* tbuf, unsigned block) const {
        if (!tbuf || threadIdx.x != 0) return;
        unsigned long long* o = tbuf + 65536 + (size_t)block * 8;
        o[0] = t0; o[1] = __builtin_readcyclecounter(); o[2] = r0; o[3] = wall_clock64();
        o[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
};

// ------------------------------------------------------------------------------------
// cross-lane all-reduce over W consecutive lanes (W = 2..64), entirely on the VALU: DPP quad
// permutes / row mirrors inside a 16-lane row, v_permlane16_swap / v_permlane32_swap (gfx950)
// across rows.  `__shfl_xor` lowers to ds_bpermute = one LDS round trip per step, which made the
// softmax and LayerNorm reductions the slowest part of the fused kernels.
// ------------------------------------------------------------------------------------
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xF, 0xF, true));
}
struct RedSum { static __device__ __forceinline__ float f(float a, float b) { return a + b; } };
struct RedMax { static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); } };
template <class Op> __device__ __forceinline__ float red_xor16(float x) {   // combine with lane ^ 16
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return Op::f(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <class Op> __device__ __forceinline__ float red_xor32(float x) {   // combine with lane ^ 32
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return Op::f(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <class Op, int W> __device__ __forceinline__ float allreduce(float x) {
    if constexpr (W >= 2) x = Op::f(x, dpp_mov<0xB1>(x));     // quad_perm [1,0,3,2]
    if constexpr (W >= 4) x = Op::f(x, dpp_mov<0x4E>(x));     // quad_perm [2,3,0,1]
    if constexpr (W >= 8) x = Op::f(x, dpp_mov<0x141>(x));    // row_half_mirror
    if constexpr (W >= 16) x = Op::f(x, dpp_mov<0x140>(x));   // row_mirror
    if constexpr (W >= 32) x = red_xor16<Op>(x);
    if constexpr (W >= 64) x = red_xor32<Op>(x);
    return x;
}

// Workgroup barrier for data exchanged through LDS only.  `__syncthreads()` also waits for every
// outstanding GLOBAL load (s_waitcnt vmcnt(0)), which kills software prefetch across the barrier; this
// form waits for LDS traffic only and leaves global loads in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- LDS-DMA helpers (gfx950 `buffer_load_dwordx4 ... lds` / `global_load_lds_dwordx4`) ---------------------------------
// The halo tile and the tap table of a chunk go HBM/L2 -> LDS without passing through registers.  They are written as
// inline asm on purpose: hipcc treats a builtin LDS-DMA as a pending LDS write and drains it (s_waitcnt vmcnt(0)) in front
// of the next ds_read, which would serialise the stencil behind its own prefetch; an asm statement is invisible to that
// pass, the waits are counted by hand below (one `s_waitcnt vmcnt(N)` per iteration, then the workgroup barrier).
// Destination: LDS byte address M0 + lane * 16 (wave-uniform base, lane-linear image); source: per-lane.  M0 is saved and
// restored inside the statement (the compiler owns it).  The leading s_nop covers the SGPR-written-by-VALU
// (v_readfirstlane) -> VMEM-descriptor hazard, the one after s_mov the M0 -> LDS-DMA hazard.
__device__ __forceinline__ void dma_buffer_to_lds(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);   // wave-uniform by contract; makes it provably scalar for the "s" constraint
    soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_global_to_lds(const void* src, unsigned lds_addr) {
    unsigned keep;
    lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(src) : "memory");
}
template <int N> __device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// XCD-aware workgroup -> tile order: consecutive workgroup ids go to different XCDs (id % 8, MI355X_MICROARCH.md), so give
// every XCD one contiguous run of tiles (whole image rows / images): the halo pixels that neighbouring tiles share are then
// served by that XCD's L2 instead of being fetched from HBM once per XCD.  Bijective for any grid size.
__device__ __forceinline__ int xcd_tile(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// 8 operand elements of type T (one MFMA k-slot group per lane)
template <typename T> struct Frag;
template <> struct Frag<bf16> {
    u32x4 v;  // 8 x bf16
    __device__ __forceinline__ void zero() { v = u32x4{0, 0, 0, 0}; }
};
template <> struct Frag<f16> {
    u32x4 v;  // 8 x f16
    __device__ __forceinline__ void zero() { v = u32x4{0, 0, 0, 0}; }
};
template <> struct Frag<float> {
    f32x4 lo, hi;  // 8 x f32
    __device__ __forceinline__ void zero() { lo = f32x4{0, 0, 0, 0}; hi = lo; }
};

// D(16x16) += A(16 x 32) * B(32 x 16).  A-frag: lane holds row (lane&15), 8 k-slots of group
// g = lane>>4; B-frag: lane holds col (lane&15), the SAME 8 k-slots.  Only the pairing of A and
// B slots matters, so both operand types use one loading pattern.  D: col = lane&15,
// row = 4*(lane>>4) + reg.
__device__ __forceinline__ void mma16(f32x4& d, const Frag<bf16>& a, const Frag<bf16>& b) {
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, a.v),
                                                __builtin_bit_cast(mfma_bf16x8, b.v), d, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& d, const Frag<f16>& a, const Frag<f16>& b) {
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_f16x8, a.v), __builtin_bit_cast(mfma_f16x8, b.v), d, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& d, const Frag<float>& a, const Frag<float>& b) {
    // exact-f32 MFMA (v_mfma_f32_16x16x4_f32): 8 steps, step j pairs slot j of every lane group.
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[0], b.lo[0], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[1], b.lo[1], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[2], b.lo[2], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[3], b.lo[3], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[0], b.hi[0], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[1], b.hi[1], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[2], b.hi[2], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[3], b.hi[3], d, 0, 0, 0);
}

// load 8 consecutive T elements (16 B for bf16, 32 B for f32) into a fragment
__device__ __forceinline__ void load_frag(Frag<bf16>& f, const bf16* p) { f.v = *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void load_frag(Frag<f16>& f, const f16* p) { f.v = *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void load_frag(Frag<float>& f, const float* p) {
    f.lo = *reinterpret_cast<const f32x4*>(p);
    f.hi = *reinterpret_cast<const f32x4*>(p + 4);
}

// store 4 consecutive values as T
__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
    u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    *reinterpret_cast<u32x2*>(p) = o;
}
__device__ __forceinline__ void store4(f16* p, f32x4 v) {
    u32x2 o = {pack2h(v[0], v[1]), pack2h(v[2], v[3])};
    *reinterpret_cast<u32x2*>(p) = o;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store1(bf16* p, float v) { p->v = f2bf(v); }
__device__ __forceinline__ void store1(f16* p, float v) { p->v = f2h(v); }
__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ float load1(const bf16* p) { return bf2f(p->v); }
__device__ __forceinline__ float load1(const f16* p) { return h2f(p->v); }
__device__ __forceinline__ float load1(const float* p) { return *p; }

// window-order row m -> flat token index in (B,H,W) for cyclic shift `shift`
// (reference model.py:957 roll(-shift) then window_partition :713-714; the inverse mapping
//  is window_reverse :720,725 then roll(+shift) :980 -- the same index pairs).
__device__ __forceinline__ int window_row_to_token(int m, int H, int W, int shift) {
    const int nWc = W >> 3, nW = (H >> 3) * nWc;
    const int bw = m >> 6, t = m & 63;
    const int b = bw / nW, wi = bw - b * nW;
    const int wr = wi / nWc, wc = wi - wr * nWc;
    int h = (wr << 3) + (t >> 3) + shift; if (h >= H) h -= H;
    int w = (wc << 3) + (t & 7) + shift;  if (w >= W) w -= W;
    return (b * H + h) * W + w;
}

// The same mapping for the 64 rows of ONE window, with everything that depends only on the window hoisted: a workgroup
// that walks a window calls window_geom once (its argument is wave-uniform -> the divisions run on the scalar unit) and
// token() per row (two adds, two wrap-arounds, one multiply-add).  window_row_to_token above costs three integer divisions
// per call (~100 VALU instructions), which the per-row callers paid 16 times per thread in front of their loads.
struct WinGeom { int base, h0, w0, H, W, img; };
__device__ __forceinline__ WinGeom window_geom(int bw, int H, int W, int shift) {
    const int nWc = W >> 3, nW = (H >> 3) * nWc;
    const int b = bw / nW, wi = bw - b * nW;
    const int wr = wi / nWc, wc = wi - wr * nWc;
    return WinGeom{b * H * W, (wr << 3) + shift, (wc << 3) + shift, H, W, b};
}
__device__ __forceinline__ int window_token(const WinGeom& g, int t) {   // t = row within the window, 0..63
    int h = g.h0 + (t >> 3); if (h >= g.H) h -= g.H;
    int w = g.w0 + (t & 7);  if (w >= g.W) w -= g.W;
    return g.base + h * g.W + w;
}

// ------------------------------------------------------------------------------------
// host side: error reporting
// ------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int variant(const char* key, int dflt);   // UF_VARIANT="key=value,...": launch-variant selector for A/B runs and bit-identity tests (uf_core.hip)
int check_launch(const char* what);

// Opt-in timing (uf_timing_enable): brackets one kernel launch with HIP events on its stream and
// books its algorithmic flops / bytes under `name`.  A no-op (one branch) when disabled.
// Opt a kernel into its dynamic LDS size, once per (kernel instantiation, device): the attribute belongs to the device's
// copy of the function, so a process that drives several GPUs has to set it on each (one process per GPU is the normal
// mode, but the library must not depend on it).  `done` is the call site's static flag array.  Returns UF_OK or sets the
// thread's error text.
inline int ensure_dynamic_lds(const void* kernel, int bytes, bool (&done)[64], const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (done[dev]) return UF_OK;   // benign race: the attribute call is idempotent
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute(%d B of LDS) failed: %s", what, bytes, hipGetErrorString(e));
        return UF_ERR_LAUNCH;
    }
    done[dev] = true;
    return UF_OK;
}

bool timing_enabled();
struct ScopedTimer {
    ScopedTimer(const char* name, double flops, double bytes, hipStream_t st);
    ~ScopedTimer();
    hipStream_t st_;
    int idx_;
};

#define UF_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            ::uf::set_error(__VA_ARGS__);    \
            return (code);                   \
        }                                    \
    } while (0)

// run a statement with TT bound to the operand type of `dtype` (checked with dtype_ok beforehand; anything else runs as f32)
#define UF_DISPATCH(dtype, TT, ...)                                                  \
    do {                                                                             \
        if ((dtype) == UF_BF16) { using TT = ::uf::bf16; __VA_ARGS__; }              \
        else if ((dtype) == UF_F16) { using TT = ::uf::f16; __VA_ARGS__; }           \
        else { using TT = float; __VA_ARGS__; }                                      \
    } while (0)

inline size_t dtype_size(int dt) { return (dt == UF_BF16 || dt == UF_F16) ? 2 : 4; }
inline bool dtype_ok(int dt) { return dt == UF_F32 || dt == UF_BF16 || dt == UF_F16; }
inline bool dtype_half(int dt) { return dt == UF_BF16 || dt == UF_F16; }
inline const char* dtype_name(int dt) { return dt == UF_BF16 ? "bf16" : (dt == UF_F16 ? "f16" : "f32"); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace uf
