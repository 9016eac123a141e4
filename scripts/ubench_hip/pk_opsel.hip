// Reproducer (round 4): packed-f32 VALU instructions whose op_sel makes the LOW result read the HIGH half of a 64-bit operand pair, executed on a SIMD
// that another wave keeps busy with MFMAs (or LDS reads).  Found through the LDS-staged stem (input_proj2_kernel): its `v_pk_fma_f32 ... op_sel:[0,1,0]`
// gave wrong low results in lanes 48-63 whenever a kernel of another HIP stream with MFMA waves shared the CU (profiles/r04_run17.txt, r04_run18.txt);
// with every operand in a register of its own (op_sel_hi:[1,0,1] forms only) it never did.  This program checks each op_sel form in isolation:
// waves 4..7 of a workgroup run the packed instruction on per-lane inputs next to a scalar v_fma_f32 / v_add_f32 / v_mul_f32 reference and count
// mismatching results per lane quarter; waves 0..3 (one per SIMD, the same SIMDs) run role A: nothing, back-to-back MFMAs, or LDS reads.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 pk_opsel.hip -o pk_opsel && ./pk_opsel
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int ITER = 16384;
enum Form { PKFMA_PLAIN, PKFMA_S1HI_BCAST, PKFMA_S1LO_BCAST, PKFMA_S0_SWAP, PKFMA_S0HI_BCAST, PKADD_CROSS, PKMUL_S0_SWAP, PKMOV_SWAP, PKFMA_S1HI_BCAST_LDS, PKMUL_S1HI_BCAST, PKFMA_S2HI_BCAST, PKADD_S1HI_BCAST, PKFMA_S1_SWAP, PKFMA_S0LO_BCAST, PKFMA_S2LO_BCAST, PKMUL_S1LO_BCAST, PKADD_S1LO_BCAST, PKADD_S0LO_BCAST, NFORM };
static const char* form_names[] = {"v_pk_fma_f32 (no op_sel)", "v_pk_fma_f32 op_sel:[0,1,0]                    (the stem's form: src1 high half to both results)",
                                   "v_pk_fma_f32 op_sel_hi:[1,0,1]                 (src1 low half to both results)", "v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1]  (src0 halves swapped: the depthwise walk kernels)",
                                   "v_pk_fma_f32 op_sel:[1,0,0]                    (src0 high half to both results: conv3x3_dx_walk)", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]      (cross add: attn_block)",
                                   "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]      (src0 halves swapped)", "v_pk_mov_b32 op_sel:[1,0]                      (both halves from the high register)",
                                   "v_pk_fma_f32 op_sel:[0,1,0], src1 pair fresh from ds_read_b64", "v_pk_mul_f32 op_sel:[0,1]                      (src1 high half to both results)",
                                   "v_pk_fma_f32 op_sel:[0,0,1]                    (src2 high half to both results)", "v_pk_add_f32 op_sel:[0,1]                      (src1 high half to both results)",
                                   "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]  (src1 halves swapped)", "v_pk_fma_f32 op_sel_hi:[0,1,1]                 (src0 low half to both results)",
                                   "v_pk_fma_f32 op_sel_hi:[1,1,0]                 (src2 low half to both results)", "v_pk_mul_f32 op_sel_hi:[1,0]                   (src1 low half to both results)",
                                   "v_pk_add_f32 op_sel_hi:[1,0]                   (src1 low half to both results)", "v_pk_add_f32 op_sel_hi:[0,1]                   (src0 low half to both results)"};

__device__ __forceinline__ float sfma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float sadd(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float smul(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

template <int FORM> __device__ __forceinline__ bool one(f32x2 s0, f32x2 s1, f32x2 s2, const f32x2* ldsp) {
    f32x2 d, r;
    if constexpr (FORM == PKFMA_PLAIN) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(s0), "v"(s1), "v"(s2)); r = f32x2{sfma(s0.x, s1.x, s2.x), sfma(s0.y, s1.y, s2.y)}; }
    else if constexpr (FORM == PKFMA_S1HI_BCAST) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(s0), "v"(s1), "v"(s2)); r = f32x2{sfma(s0.x, s1.y, s2.x), sfma(s0.y, s1.y, s2.y)}; }
    else if constexpr (FORM == PKFMA_S1LO_BCAST) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(s0), "v"(s1), "v"(s2)); r = f32x2{sfma(s0.x, s1.x, s2.x), sfma(s0.y, s1.x, s2.y)}; }
    else if constexpr (FORM == PKFMA_S0_SWAP) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(s0), "v"(s1), "v"(s2)); r = f32x2{sfma(s0.y, s1.x, s2.x), sfma(s0.x, s1.y, s2.y)}; }
    else if constexpr (FORM == PKFMA_S0HI_BCAST) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d) : "v"(s0), "v"(s1), "v"(s2)); r = f32x2{sfma(s0.y, s1.x, s2.x), sfma(s0.y, s1.y, s2.y)}; }
    else if constexpr (FORM == PKADD_CROSS) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(s0), "v"(s1)); r = f32x2{sadd(s0.x, s1.y), sadd(s0.y, s1.x)}; }
    else if constexpr (FORM == PKMUL_S0_SWAP) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(s0), "v"(s1)); r = f32x2{smul(s0.y, s1.x), smul(s0.x, s1.y)}; }
    else if constexpr (FORM == PKMOV_SWAP) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(s0), "v"(s1)); r = f32x2{s0.y, s1.x}; }
    else if constexpr (FORM == PKMUL_S1HI_BCAST) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(s0), "v"(s1)); r = f32x2{smul(s0.x, s1.y), smul(s0.y, s1.y)}; }
    else if constexpr (FORM == PKFMA_S2HI_BCAST) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(d) : "v"(s0), "v"(s1), "v"(s2)); r = f32x2{sfma(s0.x, s1.x, s2.y), sfma(s0.y, s1.y, s2.y)}; }
    else if constexpr (FORM == PKADD_S1HI_BCAST) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(s0), "v"(s1)); r = f32x2{sadd(s0.x, s1.y), sadd(s0.y, s1.y)}; }
    else if constexpr (FORM == PKFMA_S1_SWAP) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(s0), "v"(s1), "v"(s2)); r = f32x2{sfma(s0.x, s1.y, s2.x), sfma(s0.y, s1.x, s2.y)}; }
    else if constexpr (FORM == PKFMA_S0LO_BCAST) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(s0), "v"(s1), "v"(s2)); r = f32x2{sfma(s0.x, s1.x, s2.x), sfma(s0.x, s1.y, s2.y)}; }
    else if constexpr (FORM == PKFMA_S2LO_BCAST) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(s0), "v"(s1), "v"(s2)); r = f32x2{sfma(s0.x, s1.x, s2.x), sfma(s0.y, s1.y, s2.x)}; }
    else if constexpr (FORM == PKMUL_S1LO_BCAST) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(s0), "v"(s1)); r = f32x2{smul(s0.x, s1.x), smul(s0.y, s1.x)}; }
    else if constexpr (FORM == PKADD_S1LO_BCAST) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(s0), "v"(s1)); r = f32x2{sadd(s0.x, s1.x), sadd(s0.y, s1.x)}; }
    else if constexpr (FORM == PKADD_S0LO_BCAST) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(s0), "v"(s1)); r = f32x2{sadd(s0.x, s1.x), sadd(s0.x, s1.y)}; }
    else {   // the stem's sequence: the src1 pair comes straight out of LDS
        f32x2 l;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(l) : "v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) const f32x2*)ldsp) : "memory");
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(s0), "v"(l), "v"(s2)); r = f32x2{sfma(s0.x, l.y, s2.x), sfma(s0.y, l.y, s2.y)};
    }
    return __builtin_bit_cast(unsigned long long, d) != __builtin_bit_cast(unsigned long long, r);
}

// role of waves 0..3: 0 idle (exit), 1 MFMA loop, 2 LDS read loop
template <int FORM>
__global__ __launch_bounds__(512) void k(unsigned* bad /* [4 quarters] */, unsigned* bad_lo, int role, int iters) {
    __shared__ f32x2 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = f32x2{1.0f + i * 0.001f, -2.0f - i * 0.003f};
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (role == 1) {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
            bf16x8 z = __builtin_bit_cast(bf16x8, f32x4{1.0f + lane, 2.0f, 3.0f, 4.0f});
            for (int it = 0; it < iters * 6; ++it)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z, z, acc[i], 0, 0, 0);
            float s = 0;
            for (int i = 0; i < 8; ++i) s += acc[i][0];
            if (s == 12345.678f) bad[0] = 0xffffffffu;      // keep the loop alive
        } else if (role == 2) {
            float s = 0;
            for (int it = 0; it < iters * 12; ++it) {
                f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + ((threadIdx.x * 16 + it * 1024) & 16383));
                s += v[0];
            }
            if (s == 12345.678f) bad[0] = 0xffffffffu;
        }
        return;
    }
    unsigned nbad = 0, nlo = 0;
    for (int it = 0; it < iters; ++it) {
        const float t = (float)(it & 255) * 0.125f + lane;
        const f32x2 s0 = {1.5f + t, -0.75f - 0.5f * t}, s1 = {0.3125f + 0.25f * t, 7.0f - t}, s2 = {t * t * 0.01f, 3.0f};
        const bool m = one<FORM>(s0, s1, s2, &lds[(lane * 7 + it) & 2047]);
        nbad += m ? 1 : 0;
    }
    (void)nlo;
    if (nbad) atomicAdd(&bad[lane >> 4], nbad);
}

template <int FORM> void run(unsigned* dbad) {
    for (int role = 0; role < 3; ++role) {
        hipMemset(dbad, 0, 8 * sizeof(unsigned));
        hipLaunchKernelGGL(k<FORM>, dim3(1024), dim3(512), 0, 0, dbad, dbad + 4, role, ITER);
        hipDeviceSynchronize();
        unsigned h[8];
        hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
        printf("  beside %-12s mismatches in lanes 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u  (of %llu results per quarter)\n",
               role == 0 ? "nothing:" : (role == 1 ? "MFMA waves:" : "LDS reads:"), h[0], h[1], h[2], h[3], (unsigned long long)1024 * 4 * 16 * ITER);
    }
}

int main() {
    unsigned* dbad;
    hipMalloc(&dbad, 8 * sizeof(unsigned));
#define RUN(F) printf("%s\n", form_names[F]); run<F>(dbad);
    RUN(PKFMA_PLAIN) RUN(PKFMA_S1HI_BCAST) RUN(PKFMA_S1LO_BCAST) RUN(PKFMA_S0_SWAP) RUN(PKFMA_S0HI_BCAST) RUN(PKADD_CROSS) RUN(PKMUL_S0_SWAP) RUN(PKMOV_SWAP) RUN(PKFMA_S1HI_BCAST_LDS) RUN(PKMUL_S1HI_BCAST) RUN(PKFMA_S2HI_BCAST) RUN(PKADD_S1HI_BCAST) RUN(PKFMA_S1_SWAP) RUN(PKFMA_S0LO_BCAST) RUN(PKFMA_S2LO_BCAST) RUN(PKMUL_S1LO_BCAST) RUN(PKADD_S1LO_BCAST) RUN(PKADD_S0LO_BCAST)
    return 0;
}
