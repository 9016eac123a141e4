// MFMA GEMM for every dense contraction of the LeWin block and the samplers (gfx950).
//
//   D[n][m] += sum_k W[n][k] * A'[m][k]        (A' = activations through an A-loader)
//
// The MFMA "A" operand is the WEIGHT tile and the "B" operand the ACTIVATION tile, so a lane's
// four accumulator registers are four consecutive output channels n of one token m: the
// epilogue (bias / GELU / residual / window_reverse scatter / QKV split / deconv scatter)
// works on 8-16 contiguous bytes per lane.
//
// Block tile 128(m) x BN(n), 4 wave64s, K tile of 128 bytes per row (64 bf16 / 32 f32),
// double-buffered LDS with 16-byte row padding, register-staged global->LDS copies issued
// before the MFMAs of the current tile (T14 of the CDNA guide), one barrier per K tile.
#include "uf_internal.h"

namespace uf {

namespace {

constexpr int BM = 128;
constexpr int ROWB = 128 + 16;  // bytes per LDS row: 128 B of K + 16 B pad

template <typename T, int AL>
__device__ __forceinline__ u32x4 load_a_chunk(const GemmParams& p, int m, int k) {
    constexpr int EPC = 16 / sizeof(T);
    u32x4 z = {0, 0, 0, 0};
    if (m >= p.M || k >= p.K) return z;
    if constexpr (AL == A_PLAIN) {
        return *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.A) + (size_t)m * p.lda + k);
    } else {
        const float* src;
        if constexpr (AL == A_FROM_R) {
            src = reinterpret_cast<const float*>(p.A) + (size_t)m * p.lda + k;
        } else {  // A_CONV_DOWN: im2col of Conv2d(k4,s2,p1) on the token layout, k = (ky*4+kx)*C + c
            const int tap = k / p.C, c = k - tap * p.C;
            const int ky = tap >> 2, kx = tap & 3;
            const int Ho = p.H >> 1, Wo = p.W_ >> 1;
            const int b = m / (Ho * Wo), r = m - b * (Ho * Wo);
            const int oy = r / Wo, ox = r - oy * Wo;
            const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
            if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W_) return z;
            src = reinterpret_cast<const float*>(p.A) + ((size_t)(b * p.H + iy) * p.W_ + ix) * p.lda + c;
        }
        if constexpr (EPC == 4) {
            return *reinterpret_cast<const u32x4*>(src);
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(src + 4);
            u32x4 o = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b2[0], b2[1]), pack2bf(b2[2], b2[3])};
            return o;
        }
    }
}

template <typename T>
__device__ __forceinline__ u32x4 load_w_chunk(const GemmParams& p, int n, int k) {
    u32x4 z = {0, 0, 0, 0};
    if (n >= p.N || k >= p.K) return z;
    return *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.W) + (size_t)n * p.K + k);
}

template <typename T, int EP>
__device__ __forceinline__ void epilogue(const GemmParams& p, int m, int n, f32x4 acc) {
    if constexpr (EP == E_STORE_T || EP == E_STORE_T_GELU) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        f32x4 v = acc + b;
        if constexpr (EP == E_STORE_T_GELU) {
            v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
        }
        store4(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n, v);
    } else if constexpr (EP == E_QKV) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        f32x4 v = acc + b;
        const int C = p.heads * p.hd;
        const int which = n / C, c = n - which * C;
        const int h = c / p.hd, d = c - h * p.hd;
        const int bw = m >> 6, t = m & 63;
        const size_t base = ((size_t)bw * p.heads + h) * (size_t)(64 * p.hd);
        if (which == 0) {
            v *= p.qscale;  // q = q * scale (model.py:497)
            store4(reinterpret_cast<T*>(p.q) + base + t * p.hd + d, v);
        } else if (which == 1) {
            store4(reinterpret_cast<T*>(p.k) + base + t * p.hd + d, v);
        } else {
            T* vt = reinterpret_cast<T*>(p.vt) + base + (size_t)d * 64 + t;
            store1(vt, v[0]); store1(vt + 64, v[1]); store1(vt + 128, v[2]); store1(vt + 192, v[3]);
        }
    } else if constexpr (EP == E_RES_WINREV || EP == E_RES) {
        int tok = m;
        if constexpr (EP == E_RES_WINREV) tok = window_row_to_token(m, p.H, p.W_, p.shift);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.resid + (size_t)tok * p.ldr + n);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)tok * p.ldo + n) = r + (acc + b);
    } else if constexpr (EP == E_STORE_R) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n) = acc + b;
    } else {  // E_UPSAMPLE: n = (dy*2+dx)*Cout + co ; m = (b, y, x) on the (H, W) input grid
        const int qd = n / p.Cout, co = n - qd * p.Cout;
        const int dy = qd >> 1, dx = qd & 1;
        const int hw = p.H * p.W_;
        const int b = m / hw, r = m - b * hw;
        const int y = r / p.W_, x = r - y * p.W_;
        const size_t dest = ((size_t)(b * 2 * p.H + 2 * y + dy) * (2 * p.W_) + 2 * x + dx);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + co);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + dest * p.ldo + co) = acc + bb;
    }
}

template <typename T, int BN, int WGM, int WGN, int AL, int EP>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
    constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;          // elements per 128-byte K row
    constexpr int KSTEPS = BK / 32;      // MFMA k-steps (32 elements) per tile
    constexpr int TM = BM / WGM / 16, TN = BN / WGN / 16;
    constexpr int A_CH = BM / 32, W_CH = BN / 32;  // 16-byte chunks staged per thread
    static_assert(WGM * WGN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF_BYTES = (BM + BN) * ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ccol = tid & 7, crow = tid >> 3;  // staging: chunk column (0..7), first row (0..31)
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, fg = lane >> 4;

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[A_CH], rw[W_CH];
    const int nt = (p.K + BK - 1) / BK;

    auto g_load = [&](int t) {
        const int k = t * BK + ccol * EPC;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) ra[i] = load_a_chunk<T, AL>(p, m0 + crow + 32 * i, k);
#pragma unroll
        for (int i = 0; i < W_CH; ++i) rw[i] = load_w_chunk<T>(p, n0 + crow + 32 * i, k);
    };
    auto s_store = [&](int buf) {
        char* As = smem + buf * BUF_BYTES;
        char* Ws = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *reinterpret_cast<u32x4*>(As + (crow + 32 * i) * ROWB + ccol * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < W_CH; ++i) *reinterpret_cast<u32x4*>(Ws + (crow + 32 * i) * ROWB + ccol * 16) = rw[i];
    };

    g_load(0);
    s_store(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) g_load(t + 1);
        const char* As = smem + buf * BUF_BYTES + (wm * TM * 16 + fr) * ROWB + fg * (8 * (int)sizeof(T));
        const char* Ws = smem + buf * BUF_BYTES + BM * ROWB + (wn * TN * 16 + fr) * ROWB + fg * (8 * (int)sizeof(T));
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            Frag<T> af[TM], wf[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j) load_frag(af[j], reinterpret_cast<const T*>(As + j * 16 * ROWB + ks * 64));
#pragma unroll
            for (int i = 0; i < TN; ++i) load_frag(wf[i], reinterpret_cast<const T*>(Ws + i * 16 * ROWB + ks * 64));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) mma16(acc[i][j], wf[i], af[j]);
        }
        if (t + 1 < nt) s_store(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int n = n0 + (wn * TN + i) * 16 + fg * 4;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + (wm * TM + j) * 16 + fr;
            if (m < p.M && n < p.N) epilogue<T, EP>(p, m, n, acc[i][j]);
        }
    }
}

template <typename T, int BN, int WGM, int WGN, int AL, int EP>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = 2 * (BM + BN) * ROWB;
    auto kern = gemm_kernel<T, BN, WGM, WGN, AL, EP>;
    static bool attr_done = false;  // benign race: the attribute call is idempotent
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) {
            set_error("gemm: hipFuncSetAttribute(%d B) failed: %s", smem, hipGetErrorString(e));
            return UF_ERR_LAUNCH;
        }
        attr_done = true;
    }
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN);
    static char name[64] = "";
    if (!name[0]) snprintf(name, sizeof(name), "gemm_%s_bn%d_a%d_e%d", sizeof(T) == 2 ? "bf16" : "f32", BN, AL, EP);
    const double sz = sizeof(T), mn = (double)p.M * p.N, mk = (double)p.M * p.K;
    const double a_bytes = AL == A_PLAIN ? mk * sz : (AL == A_FROM_R ? mk * 4 : mk);  // conv-down reads each input once
    const double o_bytes = (EP == E_RES || EP == E_RES_WINREV) ? mn * 8 : ((EP == E_STORE_R || EP == E_UPSAMPLE) ? mn * 4 : mn * sz);
    {
        ScopedTimer tm(name, 2.0 * mn * p.K, a_bytes + (double)p.N * p.K * sz + o_bytes, stream);
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
    }
    return check_launch("gemm");
}

template <typename T, int AL, int EP>
int launch_bn(const GemmParams& p, hipStream_t stream) {
    if (p.N <= 32) return launch_cfg<T, 32, 4, 1, AL, EP>(p, stream);
    if (p.N <= 64) return launch_cfg<T, 64, 2, 2, AL, EP>(p, stream);
    return launch_cfg<T, 128, 2, 2, AL, EP>(p, stream);
}

template <typename T>
int launch_t(const GemmParams& p, int aload, int epi, hipStream_t stream) {
    if (aload == A_PLAIN) {
        switch (epi) {
            case E_STORE_T: return launch_bn<T, A_PLAIN, E_STORE_T>(p, stream);
            case E_STORE_T_GELU: return launch_bn<T, A_PLAIN, E_STORE_T_GELU>(p, stream);
            case E_QKV: return launch_bn<T, A_PLAIN, E_QKV>(p, stream);
            case E_RES_WINREV: return launch_bn<T, A_PLAIN, E_RES_WINREV>(p, stream);
            case E_RES: return launch_bn<T, A_PLAIN, E_RES>(p, stream);
            default: break;
        }
    } else if (aload == A_CONV_DOWN && epi == E_STORE_R) {
        return launch_bn<T, A_CONV_DOWN, E_STORE_R>(p, stream);
    } else if (aload == A_FROM_R && epi == E_UPSAMPLE) {
        return launch_bn<T, A_FROM_R, E_UPSAMPLE>(p, stream);
    }
    set_error("gemm: unsupported (aload=%d, epilogue=%d) combination", aload, epi);
    return UF_ERR_UNSUPPORTED;
}

}  // namespace

int launch_gemm(const GemmParams& p, int aload, int epi, uf_dtype dtype, hipStream_t stream) {
    const int epc = dtype == UF_BF16 ? 8 : 4;
    UF_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, UF_ERR_SHAPE, "gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    UF_REQUIRE(p.K % epc == 0 && p.N % 4 == 0, UF_ERR_SHAPE, "gemm: K=%d must be a multiple of %d and N=%d of 4", p.K, epc, p.N);
    UF_REQUIRE(p.A && p.W && p.bias, UF_ERR_NULL, "gemm: null operand");
    UF_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0 && ((uintptr_t)p.bias % 16) == 0,
               UF_ERR_ALIGN, "gemm: operands must be 16-byte aligned");
    if (aload == A_PLAIN) UF_REQUIRE(p.lda % epc == 0, UF_ERR_ALIGN, "gemm: lda=%d not a multiple of %d", p.lda, epc);
    else UF_REQUIRE(p.lda % 4 == 0, UF_ERR_ALIGN, "gemm: lda=%d not a multiple of 4", p.lda);
    if (aload == A_CONV_DOWN) UF_REQUIRE(p.C % 8 == 0, UF_ERR_SHAPE, "downsample: C=%d must be a multiple of 8", p.C);
    if (dtype == UF_BF16) return launch_t<bf16>(p, aload, epi, stream);
    if (dtype == UF_F32) return launch_t<float>(p, aload, epi, stream);
    set_error("gemm: unknown dtype %d", (int)dtype);
    return UF_ERR_UNSUPPORTED;
}

}  // namespace uf

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int uf_linear_fwd(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                             int act, uf_dtype dtype, void* stream) {
    uf::GemmParams p{};
    p.A = A; p.lda = K; p.W = W; p.bias = bias; p.M = M; p.N = N; p.K = K; p.out = out; p.ldo = N;
    UF_REQUIRE(out, UF_ERR_NULL, "uf_linear_fwd: null out");
    UF_REQUIRE(act == 0 || act == 1, UF_ERR_UNSUPPORTED, "uf_linear_fwd: act must be 0 or 1");
    return uf::launch_gemm(p, uf::A_PLAIN, act ? uf::E_STORE_T_GELU : uf::E_STORE_T, dtype, (hipStream_t)stream);
}

extern "C" int uf_qkv_fwd(const void* A, const void* Wqkv, const float* bqkv, void* q, void* k, void* vt, int M,
                          int C, int heads, uf_dtype dtype, void* stream) {
    UF_REQUIRE(q && k && vt, UF_ERR_NULL, "uf_qkv_fwd: null output");
    UF_REQUIRE(heads > 0 && C % heads == 0, UF_ERR_SHAPE, "uf_qkv_fwd: C=%d heads=%d", C, heads);
    const int hd = C / heads;
    UF_REQUIRE(hd == 16 || hd == 32, UF_ERR_UNSUPPORTED, "uf_qkv_fwd: head_dim %d (16 or 32 supported)", hd);
    UF_REQUIRE(M % 64 == 0, UF_ERR_SHAPE, "uf_qkv_fwd: M=%d is not a whole number of 64-token windows", M);
    uf::GemmParams p{};
    p.A = A; p.lda = C; p.W = Wqkv; p.bias = bqkv; p.M = M; p.N = 3 * C; p.K = C;
    p.q = q; p.k = k; p.vt = vt; p.heads = heads; p.hd = hd; p.qscale = (float)(1.0 / sqrt((double)hd));  // python: head_dim ** -0.5, rounded once to f32
    return uf::launch_gemm(p, uf::A_PLAIN, uf::E_QKV, dtype, (hipStream_t)stream);
}
