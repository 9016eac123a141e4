// Round 6 prototype, measured inside the library and removed again: the A-STATIONARY form of dY W2 * GELU'(c) (uf_linear_mul_dgelu at K = C, N = 4C).
// Bit-identical to the tiled GEMM (E_STORE_T_MUL_DGELU) on every shape tested (row tails, column split, non-zero bias); over the stage shapes with C >= 128 at
// batch 32: 822 -> 766 us (-7 %); the training step: 62.65 / 62.89 / 63.19 ms against 62.68 / 63.36 / 62.53 ms -- no difference (profiles/r06_run34_*).
// The product is bound by the GELU' of its store (exp2 + rcp + ~12 more VALU instructions for each of 4C values per token: ~5 K cycles of epilogue per 64 x 64
// unit and wave against 2 K cycles of MFMAs at C = 256), not by how often the dy rows are staged.  This file is a record; it is not built (it was part of
// uformer_amd/csrc/uf_lngemm.hip, between ln_gemm_kernel and its launchers, and uses that file's helpers).
//
// ---------------------------------------------------------------------------------------------------------------
// A-stationary form of the input-gradient GEMM of linear2 with GELU'(c) in its store (round 6; training, 2-byte operand types, K = C = 128 / 256 / 512):
//
//     dc[m][n] = T( T(dy[m][:] . W2t[n][:] + bias[n]) * GELU'(c[m][n]) )          n < N = 4C          (uf_linear_mul_dgelu, model.py:666-685 backwards)
//
// The tiled GEMM (uf_gemm.hip, E_STORE_T_MUL_DGELU) walks 128 x 128 output tiles: with K = C a tile is two to eight K tiles of MFMAs between a prologue that
// stages both operands and an epilogue that reads 32 KB of c and stores 32 KB of dc, and the 8-16 column tiles of a row block each stage the dy rows again:
// 175 us at 131072 x 1024 x 256 = 3.4 TB/s of compulsory traffic, 117 us at 32768 x 2048 x 512 (profiles/r06_train_kernels_hip_events.json).  Here, as in
// ln_gemm above, a workgroup owns BM rows of dy, copies them into LDS ONCE, and its four waves walk the 64 x 64 output units without barriers: fragment-major
// weights L2 -> registers through a 3-deep ring (the next unit's first fragments requested before the epilogue), the unit's c chunks requested at the top of
// the epilogue, tile pairs exchanged with v_permlane16_swap so that a lane multiplies and stores 8 consecutive channels (16 bytes) of one token.
// Same MFMAs in the same order per accumulator, the same roundings (T(acc + bias), then * GELU'(T c), then T): bit-identical to the tiled form
// (tests/test_gpu_bwd.py::test_linear_mul_dgelu_a_stationary_bit_identical).
// ---------------------------------------------------------------------------------------------------------------
struct DcGemmParams {
    const void* A; int lda;          // T[M][lda]: dy (already scaled / cast)
    const void* Wfm;                 // fragment-major T[N][C] (uf_pack_weight_fm of the (N, K = C) weight)
    const float* bias;               // f32[N]
    const void* pre; void* out;      // T[M][N]: c (read), dc (written)
    int M, N;
};

template <typename T, int C, int BM>
__global__ __launch_bounds__(256, 2) void dc_gemm_kernel(const DcGemmParams p) {
    static_assert(sizeof(T) == 2 && C % 32 == 0 && BM % 64 == 0, "2-byte operand types");
    constexpr int SA = C * 2 + 16, KS = C / 32, MH = BM / 64, RING = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const T* Wt = reinterpret_cast<const T*>(p.Wfm);
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* pre = reinterpret_cast<const T*>(p.pre);
    T* out = reinterpret_cast<T*>(p.out);

    // column groups of this workgroup (blockIdx.y splits them when M alone gives too few workgroups)
    const int n_groups_all = p.N >> 6;
    const int g0 = (int)((long long)n_groups_all * blockIdx.y / gridDim.y), g1 = (int)((long long)n_groups_all * (blockIdx.y + 1) / gridDim.y);
    const int n_units = MH * (g1 - g0);
    const T* wrow[4];
    Frag<T> wf[RING][4];
    auto wload = [&](int ks, int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_frag(wf[slot][i], wrow[i] + ks * 512);
    };
    auto unit_prefetch = [&](int u) {
        const int nb = (g0 + u / MH) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) wrow[i] = Wt + ((size_t)((nb >> 4) + i) * KS * 64 + lane) * 8;
#pragma unroll
        for (int s = 0; s < RING - 1; ++s)
            if (s < KS) wload(s, s);
    };
    if (wave < n_units) unit_prefetch(wave);          // in flight while the dy rows are staged

    // ---- the BM dy rows into LDS: 16-byte chunks, all loads of a thread issued before its stores; rows past M are zeros
    {
        constexpr int CPR = C / 8, NCH = BM * CPR, PER = NCH / 256;
        static_assert(NCH % 256 == 0, "chunks per thread");
        u32x4 v[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = q * 256 + tid, r = idx / CPR, cb = idx - r * CPR;
            const int m = m0 + r, mc = m < p.M ? m : p.M - 1;
            v[q] = *reinterpret_cast<const u32x4*>(A + (size_t)mc * p.lda + cb * 8);
            if (m >= p.M) v[q] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = q * 256 + tid, r = idx / CPR, cb = idx - r * CPR;
            *reinterpret_cast<u32x4*>(As + r * SA + cb * 16) = v[q];
        }
    }
    lds_barrier();

#pragma unroll 1
    for (int u = wave; u < n_units; u += 4) {
        const int mh = u % MH, ng = g0 + u / MH;
        const int nbase = ng * 64, mbase = mh * 64;
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* arow = As + (mbase + fr) * SA + fg * 16;
        Frag<T> af[2][4];
        auto aload = [&](int ks, int slot) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_frag(af[slot][j], reinterpret_cast<const T*>(arow + j * 16 * SA + ks * 64));
        };
        aload(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + RING - 1 < KS) wload(ks + RING - 1, (ks + RING - 1) % RING);
            if (ks + 1 < KS) aload(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(acc[i][j], wf[ks % RING][i], af[ks & 1][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (u + 4 < n_units) unit_prefetch(u + 4);
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue: the unit's 8 chunks of c per lane requested first (clamped rows, unconditional), then T(acc + bias) * GELU'(c) -> T, 16 bytes per store
        u32x4 cv[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + mbase + j * 16 + fr, mc = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int n = nbase + (2 * ip + (fg & 1)) * 16 + (fg >> 1) * 8;
                cv[j][ip] = *reinterpret_cast<const u32x4*>(pre + (size_t)mc * p.N + n);
            }
        }
        f32x4 bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4*>(p.bias + nbase + i * 16 + fg * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + mbase + j * 16 + fr;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {                          // tile pair (2 ip, 2 ip + 1)
                const f32x4 va = acc[2 * ip][j] + bv[2 * ip], vb = acc[2 * ip + 1][j] + bv[2 * ip + 1];
                const unsigned a0 = pack2<T>(va[0], va[1]), a1 = pack2<T>(va[2], va[3]);
                const unsigned b0 = pack2<T>(vb[0], vb[1]), b1 = pack2<T>(vb[2], vb[3]);
                // after the swap: even fg lanes hold 8 channels of tile 2 ip, odd fg lanes 8 channels of tile 2 ip + 1 (as the EP_GELU store of ln_gemm)
                const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                float f[8], a[8];
                unpack8<T>(u32x4{s0[0], s1[0], s0[1], s1[1]}, f);
                unpack8<T>(cv[j][ip], a);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] *= gelu_grad_t<T>(a[e]);
                const int n = nbase + (2 * ip + (fg & 1)) * 16 + (fg >> 1) * 8;
                if (m < p.M) *reinterpret_cast<u32x4*>(out + (size_t)m * p.N + n) = pack8<T>(f);
            }
        }
    }
}

template <typename T, int C, int BM>
int launch_dc(const DcGemmParams& p, hipStream_t st) {
    constexpr int smem = BM * (C * 2 + 16);
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = dc_gemm_kernel<T, C, BM>;
    static bool lds_done[64] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, lds_done, "dc_gemm")) return rc;
    char name[96] = "";
    if (timing_enabled()) snprintf(name, sizeof(name), "dc_gemm_%s_c%d_bm%d %dx%dx%d", TypeName<T>::s, C, BM, p.M, p.N, C);
    const double mn = (double)p.M * p.N;
    {
        ScopedTimer tm(name, 2.0 * mn * C, (double)p.M * C * 2 + (double)p.N * C * 2 + 2.0 * mn * 2, st);
        const int mb = (p.M + BM - 1) / BM, groups = p.N / 64;
        int nsplit = 1;
        while (mb * nsplit < 512 && nsplit * 2 <= groups) nsplit *= 2;   // >= 2 workgroups per CU when possible
        hipLaunchKernelGGL(kern, dim3(mb, nsplit), dim3(256), smem, st, p);
    }
    return check_launch("dc_gemm");
}

// A-stationary uf_linear_mul_dgelu for the K = C products of the LeFF backward (round 6): W_fm = uf_pack_weight_fm of the (N, K) weight.  Built for the 2-byte
// operand types at K = 128 / 256 / 512 with N a multiple of 64; UF_ERR_UNSUPPORTED otherwise (the caller falls back to uf_linear_mul_dgelu).  Bit-identical to it.
extern "C" int uf_linear_mul_dgelu_fm(const void* A, const void* W_fm, const float* bias, const void* pre, void* out, int M, int N, int K, uf_dtype dtype, void* stream) {
    UF_REQUIRE(A && W_fm && bias && pre && out, UF_ERR_NULL, "uf_linear_mul_dgelu_fm: null pointer");
    UF_REQUIRE(dtype_half(dtype) && (K == 128 || K == 256 || K == 512) && M > 0 && N > 0 && N % 64 == 0, UF_ERR_UNSUPPORTED,
               "uf_linear_mul_dgelu_fm: built for bf16 / f16, K = 128 / 256 / 512, N a multiple of 64 (got dtype %d, M=%d N=%d K=%d)", (int)dtype, M, N, K);
    UF_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W_fm % 16) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)pre % 16) == 0 && ((uintptr_t)out % 16) == 0, UF_ERR_ALIGN,
               "uf_linear_mul_dgelu_fm: operands must be 16-byte aligned");
    DcGemmParams p{A, K, W_fm, bias, pre, out, M, N};
    hipStream_t st = (hipStream_t)stream;
#define UF_DC(TT)                                                                                   \
    switch (K) {                                                                                    \
        case 128: return M >= 128 * 512 ? launch_dc<TT, 128, 128>(p, st) : launch_dc<TT, 128, 64>(p, st);   \
        case 256: return M >= 128 * 512 ? launch_dc<TT, 256, 128>(p, st) : launch_dc<TT, 256, 64>(p, st);   \
        default: return launch_dc<TT, 512, 64>(p, st);                                              \
    }
    if (dtype == UF_BF16) { UF_DC(bf16) }
    UF_DC(f16)
#undef UF_DC
}
