// The depthwise 3x3 of LeFF on the matrix pipe (round 4), shared by leff2 (halo tile DMA-staged from h1) and leff3 (halo tile computed in place):
// one job = pixel tiles [pt0, pt0 + NPT) of a 16-channel group of a 64-channel halo tile -> the operand tile of linear2.
#pragma once
#include "uf_common.h"

namespace uf {

template <typename T> __device__ __forceinline__ unsigned cvt16(float f);
template <> __device__ __forceinline__ unsigned cvt16<bf16>(float f) { return f2bf(f); }
template <> __device__ __forceinline__ unsigned cvt16<f16>(float f) { return f2h(f); }
template <> __device__ __forceinline__ unsigned cvt16<float>(float) { return 0; }
template <typename T> __device__ __forceinline__ float back16(unsigned h);
template <> __device__ __forceinline__ float back16<bf16>(unsigned h) { return bf2f((uint16_t)h); }
template <> __device__ __forceinline__ float back16<f16>(unsigned h) { return h2f((uint16_t)h); }
template <> __device__ __forceinline__ float back16<float>(unsigned) { return 0.f; }

// One job of the MFMA stencil (UF_MCONV 1 / 2): pixel tiles [pt0, pt0 + NPT) of the 16-channel group gq of an interval -> operand tile.
// Hs: the group's halo tile in the ring slot, Wl: its tap table [10][KC] f32, At: the operand tile [64][KC] (row stride SAT).
template <typename T, int MCV, int NPT, int KC, int SAT, int HROW>
__device__ __forceinline__ void mconv_job(const char* Hs, const float* Wl, char* At, int gq, int pt0, const int* boff, const unsigned* msk, unsigned hshift, int fr, int fg) {
    constexpr int NKS = MCV == 2 ? 9 : 5;
    const float* wc = Wl + gq * 16 + fr;                  // this lane's channel in the tap table
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(Wl + 9 * KC + gq * 16 + fg * 4);
    float wt[NKS];                                        // the lane's tap of every k-step, requested up front (one LDS round trip)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) wt[ks] = wc[(MCV == 2 ? ks : ((2 * ks + (fg >> 1)) < 9 ? 2 * ks + (fg >> 1) : 8)) * KC];
    f32x4 cacc[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) cacc[pt] = bias4;
    const char* Hp = Hs + pt0 * HROW;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        unsigned v16;
        if constexpr (MCV == 2) {                         // k-step = tap ks; lane groups 0,1: hi part, 2,3: lo part
            const unsigned hi = cvt16<T>(wt[ks]);
            v16 = (fg >> 1) ? cvt16<T>(wt[ks] - back16<T>(hi)) : hi;
        } else {                                          // k-step = taps 2ks (lane groups 0,1) and 2ks+1 (2,3); tap 9 = padding
            v16 = (2 * ks + (fg >> 1)) < 9 ? cvt16<T>(wt[ks]) : 0u;
        }
        const unsigned sh = v16 << hshift;
        Frag<T> af;                                       // block-diagonal weight fragment of this k-step: one non-zero 16-bit slot per lane
        af.v = u32x4{sh & msk[0], sh & msk[1], sh & msk[2], sh & msk[3]};
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            Frag<T> bf;
            bf.v = *reinterpret_cast<const u32x4*>(Hp + boff[ks] + pt * HROW);
            mma16(cacc[pt], af, bf);                      // weights as A: lane = pixel fr, channels 4 fg .. 4 fg + 3
        }
    }
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        gelu4<T>(cacc[pt]);
        store4(reinterpret_cast<T*>(At + ((pt0 + pt) * 16 + fr) * SAT) + gq * 16 + fg * 4, cacc[pt]);
    }
}

}  // namespace uf
