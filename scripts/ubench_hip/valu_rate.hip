// Micro-benchmark: issue cost of the VALU / transcendental / MFMA instruction classes the fused kernels spend their time on, per SIMD,
// with 1 / 2 / 3 / 4 waves per SIMD, and side by side (one wave of MFMAs + one wave of VALU on the same SIMD).  Round 4: the PMC passes
// show SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.4-4.6 cycles per VALU instruction in every fused kernel, and an erf-form GELU (+12 VALU per
// value) cost exactly 4.6 cycles per added instruction per SIMD -- this measures what the pipe itself can do.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int ITER = 2048, UNR = 8;

enum Op { FMA, PKFMA, EXP, RCP, CVTPK, AND, PERM, MFMA16, PKMUL, LDSR, FMA_DEP, MIX_GELU, PKFMA16, CVTPKH, GELU_H, GELU_HB };
static const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_cvt_pk_bf16_f32", "v_and_b32", "v_perm_b32", "mfma16x16x32bf16",
                              "v_pk_mul_f32", "ds_read_b128", "v_fma_f32 dependent", "gelu mix (5pk+4trans)", "v_pk_fma_f16", "v_cvt_pk_f16_f32",
                              "gelu pk-f16 poly (1cvt+11pk16) per pair", "gelu pk-f16 poly -> bf16 (+2cvt32+1cvtpk) per pair"};

template <int OP> __device__ __forceinline__ void body(float (&a)[UNR], f32x2 (&p)[UNR], f32x4 (&acc)[UNR], unsigned (&u)[UNR], const char* lds) {
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
        if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[k]));
        else if constexpr (OP == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[0]));
        else if constexpr (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[k]));
        else if constexpr (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[k]));
        else if constexpr (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
        else if constexpr (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
        else if constexpr (OP == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u[k]) : "v"(a[k]));
        else if constexpr (OP == AND) asm volatile("v_and_b32 %0, %0, %0" : "+v"(u[k]));
        else if constexpr (OP == PERM) asm volatile("v_perm_b32 %0, %0, %0, %0" : "+v"(u[k]));
        else if constexpr (OP == MFMA16) {
            bf16x8 z = __builtin_bit_cast(bf16x8, f32x4{a[k], a[k], a[k], a[k]});
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z, z, acc[k], 0, 0, 0);
        } else if constexpr (OP == LDSR) {
            f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 16 + k * 1024) & 16383));
            asm volatile("" :: "v"(v));
        } else if constexpr (OP == PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(u[k]));
        else if constexpr (OP == CVTPKH) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(u[k]) : "v"(a[k]));
        else if constexpr (OP == GELU_H || OP == GELU_HB) {   // round 6: FMA-only GELU of a PAIR in packed f16 (clamp, t^2, degree-6 Horner, t q + 0.5, x *)
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_max_f16 %3, %0, %0\n\tv_pk_min_f16 %3, %3, %3\n\tv_pk_mul_f16 %0, %3, %3\n\t"
                         "v_pk_fma_f16 %3, %0, %3, %3\n\tv_pk_fma_f16 %3, %0, %3, %3\n\tv_pk_fma_f16 %3, %0, %3, %3\n\tv_pk_fma_f16 %3, %0, %3, %3\n\t"
                         "v_pk_fma_f16 %3, %0, %3, %3\n\tv_pk_fma_f16 %3, %0, %3, %3\n\tv_pk_fma_f16 %3, %0, %3, %3\n\tv_pk_mul_f16 %0, %0, %3"
                         : "+v"(u[k]), "+v"(a[k]), "+v"(a[(k + 1) % UNR]), "+v"(u[(k + 3) % UNR]));
            if constexpr (OP == GELU_HB)
                asm volatile("v_cvt_f32_f16 %1, %0\n\tv_cvt_f32_f16 %2, %0\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(u[k]), "+v"(a[k]), "+v"(a[(k + 1) % UNR]));
        } else if constexpr (OP == MIX_GELU) {      // the packed sigmoid-form GELU of the kernels on a pair: 5 packed + 2 exp + 2 rcp
            asm volatile("v_pk_mul_f32 %0, %0, %0\n\tv_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_mul_f32 %0, %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\t"
                         "v_pk_add_f32 %0, %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_pk_mul_f32 %0, %0, %0" : "+v"(p[k]), "+v"(a[k]), "+v"(a[(k + 1) % UNR]));
        }
    }
}

// role A on waves [0, WA), role B on the rest; wave w and w + 4 share a SIMD (waves go to SIMDs in a fixed cyclic order)
template <int OPA, int OPB>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int wa) {
    __shared__ char lds[16384];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = i;
    __syncthreads();
    float a[UNR]; f32x2 p[UNR]; f32x4 acc[UNR]; unsigned u[UNR];
#pragma unroll
    for (int i = 0; i < UNR; ++i) { a[i] = 1.0f + threadIdx.x * 1e-6f + i; p[i] = f32x2{a[i], a[i]}; acc[i] = f32x4{0, 0, 0, 0}; u[i] = threadIdx.x + i; }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < wa) {
#pragma unroll 4
        for (int it = 0; it < ITER; ++it) body<OPA>(a, p, acc, u, lds); }
    else {
#pragma unroll 4
        for (int it = 0; it < ITER; ++it) body<OPB>(a, p, acc, u, lds); }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < UNR; ++i) s += a[i] + p[i][0] + acc[i][0] + u[i];
    if (s == 12345.678f) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int OPA, int OPB> void run(const char* what, int waves, int wa, unsigned long long* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OPA, OPB>), dim3(256), dim3(64 * waves), 0, 0, d, wa);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OPA, OPB>), dim3(256), dim3(64 * waves), 0, 0, d, wa);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // cycles per instruction of ONE wave of each role (s_memtime), and aggregate instructions per cycle per SIMD for role A / role B
    auto per = [](int op) { return op == MIX_GELU ? 9 : (op == GELU_H ? 12 : (op == GELU_HB ? 15 : 1)); };
    const double n = (double)ITER * UNR * per(OPA), nb = (double)ITER * UNR * per(OPB);
    const double ca = (double)h[0] / n, cb = wa < waves ? (double)h[wa] / nb : 0;
    const int per_simd_a = (wa + 3) / 4, per_simd_b = (waves - wa + 3) / 4;
    printf("%-44s waves/CU %2d (A %d/SIMD, B %d/SIMD)  A: %6.2f cyc/instr/wave -> %5.2f cyc/instr/SIMD", what, waves, per_simd_a, per_simd_b, ca, ca / per_simd_a);
    if (wa < waves) printf("   B: %6.2f cyc/instr/wave -> %5.2f /SIMD", cb, cb / per_simd_b);
    printf("   [%.1f us, clock %.2f GHz]\n", ms * 1e3, (double)h[0] / (ms * 1e6));
}

#define SOLO(OP) for (int w : {4, 8, 12, 16}) run<OP, OP>(names[OP], w, w, d);
int main() {
    unsigned long long* d; hipMalloc(&d, 8 * 4096); hipMemset(d, 0, 8 * 4096);
    SOLO(FMA) SOLO(FMA_DEP) SOLO(PKFMA) SOLO(PKMUL) SOLO(EXP) SOLO(RCP) SOLO(CVTPK) SOLO(AND) SOLO(PERM) SOLO(MFMA16) SOLO(LDSR) SOLO(MIX_GELU) SOLO(PKFMA16) SOLO(CVTPKH) SOLO(GELU_H) SOLO(GELU_HB)
    printf("--- side by side: role A on waves 0-3, role B on waves 4-7 (one of each per SIMD)\n");
    run<MFMA16, FMA>("A mfma | B v_fma_f32", 8, 4, d);
    run<MFMA16, PKFMA>("A mfma | B v_pk_fma_f32", 8, 4, d);
    run<MFMA16, EXP>("A mfma | B v_exp_f32", 8, 4, d);
    run<MFMA16, MIX_GELU>("A mfma | B gelu mix", 8, 4, d);
    run<MFMA16, LDSR>("A mfma | B ds_read_b128", 8, 4, d);
    run<FMA, EXP>("A v_fma_f32 | B v_exp_f32", 8, 4, d);
    run<FMA, LDSR>("A v_fma_f32 | B ds_read_b128", 8, 4, d);
    run<MFMA16, MIX_GELU>("A mfma (4) | B gelu mix (8)", 12, 4, d);
    run<MFMA16, GELU_H>("A mfma | B gelu pk-f16 poly", 8, 4, d);
    return 0;
}
