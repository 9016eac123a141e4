// Round 6 prototype, measured inside the library and removed again: Upsample in the A-STATIONARY form of ln_gemm (input rows converted into LDS once, 64 x 64 units
// of all 4 Cout columns walked with fragment-major weights streamed L2 -> registers, the tiled form's scatter store).  Bit-identical to the tiled GEMM (same output
// hashes); the four levels at batch 16: 131-134 -> 123 us (16.7 -> 15.1, 24.2 -> 22.0, 33.4 -> 30.1, 56.6-60.0 -> 55.3-56.5 us), batch 32: 277 -> 247 us; the bench line:
// 2428 / 2435 / 2419 against 2417 / 2446 / 2426 img/s -- no difference (profiles/r06_run37_*).  Unlike the Downsample's im2col loader, the A_FROM_R loader of the tiled GEMM
// is a plain row read + conversion: staging it once instead of N / 128 times saves little.  This file is a record; it is not built (it was part of uf_lngemm.hip and uses its helpers).
//
// ---------------------------------------------------------------------------------------------------------------
// Upsample (ConvTranspose2d k2 s2 = four 1x1 GEMMs scattered over the 2 x 2 output pixels of every input pixel, model.py:765-771), A-stationary form (round 6;
// 2-byte operand types, Cin = 128 / 256 / 512).  The tiled GEMM (uf_gemm.hip: A_FROM_R loader + E_UPSAMPLE store) converts and stages the f32 input rows again
// for every one of its N / 128 column tiles: the two deep levels (4096 x 1024 x 512, 16384 x 512 x 512) ran at 195 / 318 TFLOP/s.  As in ln_gemm above a workgroup
// owns BM input rows, converts them into LDS ONCE, and its four waves walk the 64 x 64 output units without barriers (fragment-major weights L2 -> registers,
// ring of 3 k-steps); the store is the tiled form's: 4 consecutive output channels (16 bytes of f32) of one (input pixel, quadrant) per lane.
// Same k-step order, accumulators from zero, bias added at the end: bit-identical to the tiled form.
// ---------------------------------------------------------------------------------------------------------------
struct UpGemmParams {
    const float* x; int ld;          // f32[M][ld]: input token rows
    const void* Wfm;                 // fragment-major T[4 Cout][Cin]
    const float* bias;               // f32[Cout]
    float* out; int ldo;             // f32 rows of the (2H, 2W) map, stride ldo (the decoder stage's concat buffer)
    int M, N, H, W, Cout;
};

template <typename T, int C, int BM>
__global__ __launch_bounds__(256, 2) void up_gemm_kernel(const UpGemmParams p) {
    static_assert(sizeof(T) == 2 && C % 32 == 0 && BM % 64 == 0, "2-byte operand types");
    constexpr int SA = C * 2 + 16, KS = C / 32, MH = BM / 64, RING = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const T* Wt = reinterpret_cast<const T*>(p.Wfm);
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
    if (wave < n_units) unit_prefetch(wave);
    // ---- the BM input rows: f32 -> T into LDS, 8 channels per chunk, batches of 8 chunks per thread in flight; rows past M are zeros
    {
        constexpr int CPR = C / 8, NCH = BM * CPR, PER = NCH / 256, U = PER < 8 ? PER : 8;
        static_assert(NCH % 256 == 0 && PER % U == 0, "chunks per thread");
#pragma unroll 1
        for (int q0 = 0; q0 < PER; q0 += U) {
            u32x4 lo[U], hi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = (q0 + u) * 256 + tid, r = idx / CPR, cb = idx - r * CPR;
                const int m = m0 + r, mc = m < p.M ? m : p.M - 1;
                const float* src = p.x + (size_t)mc * p.ld + cb * 8;
                lo[u] = *reinterpret_cast<const u32x4*>(src);
                hi[u] = *(reinterpret_cast<const u32x4*>(src) + 1);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = (q0 + u) * 256 + tid, r = idx / CPR, cb = idx - r * CPR;
                u32x4 v = u32x4{pack2<T>(__uint_as_float(lo[u][0]), __uint_as_float(lo[u][1])), pack2<T>(__uint_as_float(lo[u][2]), __uint_as_float(lo[u][3])),
                                pack2<T>(__uint_as_float(hi[u][0]), __uint_as_float(hi[u][1])), pack2<T>(__uint_as_float(hi[u][2]), __uint_as_float(hi[u][3]))};
                if (m0 + r >= p.M) v = u32x4{0, 0, 0, 0};
                *reinterpret_cast<u32x4*>(As + r * SA + cb * 16) = v;
            }
        }
    }
    lds_barrier();

    const int hw = p.H * p.W;
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
        // ---- store (E_UPSAMPLE of the tiled form): n = (dy * 2 + dx) * Cout + co; m = (b, y, x) on the (H, W) input grid
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nbase + i * 16 + fg * 4;
            const int qd = n / p.Cout, co = n - qd * p.Cout;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + mbase + j * 16 + fr;
                if (m < p.M) {
                    const int bb = m / hw, r = m - bb * hw, y = r / p.W, x = r - y * p.W;
                    const size_t dest = (size_t)(bb * 2 * p.H + 2 * y + (qd >> 1)) * (2 * p.W) + 2 * x + (qd & 1);
                    *reinterpret_cast<f32x4*>(p.out + dest * p.ldo + co) = acc[i][j] + bv;
                }
            }
        }
    }
}

template <typename T, int C, int BM>
int launch_up(const UpGemmParams& p, hipStream_t st) {
    constexpr int smem = BM * (C * 2 + 16);
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = up_gemm_kernel<T, C, BM>;
    static bool lds_done[64] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, lds_done, "up_gemm")) return rc;
    char name[96] = "";
    if (timing_enabled()) snprintf(name, sizeof(name), "up_gemm_%s_c%d_bm%d %dx%dx%d", TypeName<T>::s, C, BM, p.M, p.N, C);
    const double mn = (double)p.M * p.N;
    {
        ScopedTimer tm(name, 2.0 * mn * C, (double)p.M * C * 4 + (double)p.N * C * 2 + mn * 4, st);
        const int mb = (p.M + BM - 1) / BM, groups = p.N / 64;
        int nsplit = 1;
        while (mb * nsplit < 512 && nsplit * 2 <= groups) nsplit *= 2;   // >= 2 workgroups per CU when possible
        hipLaunchKernelGGL(kern, dim3(mb, nsplit), dim3(256), smem, st, p);
    }
    return check_launch("up_gemm");
}

// Upsample with the weight ALSO in the fragment-major layout (round 6): the A-stationary form where it is built (bf16 / f16, Cin = 128 / 256 / 512, 4 Cout a
// multiple of 64, Cout a multiple of 4), uf_upsample_fwd otherwise and with w_fm = NULL.  Bit-identical to uf_upsample_fwd.
extern "C" int uf_upsample_fwd(const float* x, int ld_x, const void* w, const float* bias, float* out, int ld_o, int B, int H, int W, int Cin, int Cout, uf_dtype dtype, void* stream);
extern "C" int uf_upsample_fm_fwd(const float* x, int ld_x, const void* w, const void* w_fm, const float* bias, float* out, int ld_o, int B, int H,
                                  int W, int Cin, int Cout, uf_dtype dtype, void* stream) {
    const bool built = w_fm && dtype_half(dtype) && (Cin == 128 || Cin == 256 || Cin == 512) && Cout % 16 == 0 && variant("up", 2) != 1;
    if (!built) return uf_upsample_fwd(x, ld_x, w, bias, out, ld_o, B, H, W, Cin, Cout, dtype, stream);
    UF_REQUIRE(x && bias && out, UF_ERR_NULL, "uf_upsample_fm_fwd: null pointer");
    UF_REQUIRE(B > 0 && H > 0 && W > 0, UF_ERR_SHAPE, "uf_upsample_fm_fwd: bad shape");
    UF_REQUIRE(ld_x >= Cin && ld_x % 4 == 0 && ld_o >= Cout && ld_o % 4 == 0, UF_ERR_SHAPE, "uf_upsample_fm_fwd: ld_x=%d ld_o=%d", ld_x, ld_o);
    UF_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_fm % 16) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)out % 16) == 0, UF_ERR_ALIGN,
               "uf_upsample_fm_fwd: operands must be 16-byte aligned");
    UpGemmParams p{x, ld_x, w_fm, bias, out, ld_o, B * H * W, 4 * Cout, H, W, Cout};
    hipStream_t st = (hipStream_t)stream;
#define UF_UP(TT)                                                                                                   \
    switch (Cin) {                                                                                                  \
        case 128: return p.M >= 128 * 512 ? launch_up<TT, 128, 128>(p, st) : launch_up<TT, 128, 64>(p, st);         \
        case 256: return p.M >= 128 * 512 ? launch_up<TT, 256, 128>(p, st) : launch_up<TT, 256, 64>(p, st);         \
        default: return launch_up<TT, 512, 64>(p, st);                                                              \
    }
    if (dtype == UF_BF16) { UF_UP(bf16) }
    UF_UP(f16)
#undef UF_UP
}
