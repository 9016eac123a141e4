// MFMA GEMM for every dense contraction of the LeWin block and the samplers (gfx950).
//
//   D[n][m] += sum_k W[n][k] * A'[m][k]        (A' = activations through an A-loader)
//
// The MFMA "A" operand is the WEIGHT tile and the "B" operand the ACTIVATION tile, so a lane's
// four accumulator registers are four consecutive output channels n of one token m: the
// epilogue (bias / GELU / residual / window_reverse scatter / QKV split / deconv scatter)
// works on 8-16 contiguous bytes per lane.
//
// Block tile 128(m) x BN(n), 4 wave64s, K tile of 128 bytes per row (64 bf16 / 32 f32), double-buffered LDS, one barrier per K tile.
// Two staging paths:
//   * register-staged (every loader, every type): global_load_dwordx4 -> VGPRs -> ds_write_b128 into rows padded to 144 bytes, the
//     global loads of tile t+1 issued before the MFMAs of tile t (T14 of the CDNA guide);
//   * LDS-DMA (round 4; plain loader, 2-byte types, K a multiple of 64 and >= 2 tiles): `buffer_load_dwordx4 ... lds` straight into
//     UNPADDED 128-byte rows; the bank-conflict-free placement (16-byte chunk c of row r at position c ^ (r & 7)) is obtained by
//     permuting WHICH global chunk a lane fetches, not where it lands, and the fragment reads apply the same XOR.  No staging
//     registers, no ds_write pass (per K tile a workgroup spent ~420 LDS-array cycles on ds_write_b128 next to 256 on fragment
//     reads): measured as a prototype in round 3 (scripts/ubench_hip/gemm_dma.hip, profiles/r03_gemm_dma.txt) at +10...+34 % on the
//     K-heavy shapes.  Rows past M / N read zeros through the buffer descriptor's bounds check.  Same MFMAs on the same operands in
//     the same order: bit-identical results to the register path (tests/test_gpu_ops.py::test_gemm_dma_bit_identical).
#include "uf_internal.h"

namespace uf {

namespace {

constexpr int BM = 128;
constexpr int ROWB_PAD = 128 + 16;  // bytes per LDS row of the register-staged path: 128 B of K + 16 B pad

// One 16-byte chunk of the activation tile in flight.  issue() is an UNCONDITIONAL global load from
// a clamped address (a bounds-guarded load makes hipcc emit an exec-masked branch with a vmcnt(0)
// wait per load, which serialises the whole staging pipeline); finish() -- called only when the
// chunk is written to LDS, i.e. after the MFMAs of the current tile -- converts f32 sources to the
// operand type and applies the bounds / zero-padding predicate.
template <typename T, int AL>
struct AChunk {
    static constexpr bool WIDE = (AL != A_PLAIN) && sizeof(T) == 2;  // 8 f32 source values -> 8 bf16
    u32x4 lo, hi;
    bool ok;
    __device__ __forceinline__ void issue(const GemmParams& p, int m, int k) {
        ok = m < p.M && k < p.K;
        const int mc = m < p.M ? m : p.M - 1, kc = k < p.K ? k : 0;
        const void* src;
        if constexpr (AL == A_PLAIN) {
            src = reinterpret_cast<const T*>(p.A) + (size_t)mc * p.lda + kc;
        } else if constexpr (AL == A_FROM_R) {
            src = reinterpret_cast<const float*>(p.A) + (size_t)mc * p.lda + kc;
        } else {  // A_CONV_DOWN: im2col of Conv2d(k4,s2,p1) on the token layout, k = (ky*4+kx)*C + c
            const int tap = kc / p.C, c = kc - tap * p.C;
            const int ky = tap >> 2, kx = tap & 3;
            const int Ho = p.H >> 1, Wo = p.W_ >> 1;
            const int b = mc / (Ho * Wo), r = mc - b * (Ho * Wo);
            const int oy = r / Wo, ox = r - oy * Wo;
            int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
            ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W_;     // zero padding
            iy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
            ix = ix < 0 ? 0 : (ix >= p.W_ ? p.W_ - 1 : ix);
            src = reinterpret_cast<const float*>(p.A) + ((size_t)(b * p.H + iy) * p.W_ + ix) * p.lda + c;
        }
        lo = *reinterpret_cast<const u32x4*>(src);
        if constexpr (WIDE) hi = *(reinterpret_cast<const u32x4*>(src) + 1);
    }
    __device__ __forceinline__ u32x4 finish() const {
        u32x4 v = lo;
        if constexpr (WIDE) {
            v = u32x4{pack2<T>(__uint_as_float(lo[0]), __uint_as_float(lo[1])), pack2<T>(__uint_as_float(lo[2]), __uint_as_float(lo[3])),
                      pack2<T>(__uint_as_float(hi[0]), __uint_as_float(hi[1])), pack2<T>(__uint_as_float(hi[2]), __uint_as_float(hi[3]))};
        }
        return ok ? v : u32x4{0, 0, 0, 0};
    }
};

template <typename T>
struct WChunk {
    u32x4 v;
    bool ok;
    __device__ __forceinline__ void issue(const GemmParams& p, int n, int k) {
        ok = n < p.N && k < p.K;
        const int nc = n < p.N ? n : p.N - 1, kc = k < p.K ? k : 0;
        v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.W) + (size_t)nc * p.K + kc);
    }
    __device__ __forceinline__ u32x4 finish() const { return ok ? v : u32x4{0, 0, 0, 0}; }
};

template <typename T> __device__ __forceinline__ void unpack_chunk(u32x4 v, float* f) {
    if constexpr (sizeof(T) == 2) {
        unpack8<T>(v, f);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
    }
}
template <typename T> __device__ __forceinline__ u32x4 pack_chunk(const float* f) {
    if constexpr (sizeof(T) == 2) return pack8<T>(f);
    else return u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
}

// bias of the 4 output channels starting at n (E_UPSAMPLE: channel n % Cout): requested once per tile column BEFORE the guarded stores
// (inside them every (i, j) tile paid its own dependent round trip)
template <int EP>
__device__ __forceinline__ f32x4 epilogue_bias(const GemmParams& p, int n) {
    int nc = n < p.N ? n : p.N - 4;
    if constexpr (EP == E_UPSAMPLE) nc = nc % p.Cout;
    return *reinterpret_cast<const f32x4*>(p.bias + nc);
}

template <typename T, int EP>
__device__ __forceinline__ void epilogue(const GemmParams& p, int m, int n, f32x4 acc, f32x4 b) {
    if constexpr (EP == E_RES_WINREV || EP == E_RES) {
        int tok = m;
        if constexpr (EP == E_RES_WINREV) tok = window_row_to_token(m, p.H, p.W_, p.shift);
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.resid + (size_t)tok * p.ldr + n);
        f32x4 v = acc + b;
        if (p.scale) v *= p.scale[tok / p.hw];                // training: x + DropPath(branch), bernoulli(keep) / keep per sample
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)tok * p.ldo + n) = r + v;
    } else if constexpr (EP == E_STORE_R) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n) = acc + b;
    } else {  // E_UPSAMPLE: n = (dy*2+dx)*Cout + co ; m = (b, y, x) on the (H, W) input grid
        const int qd = n / p.Cout, co = n - qd * p.Cout;
        const int dy = qd >> 1, dx = qd & 1;
        const int hw = p.H * p.W_;
        const int bb = m / hw, r = m - bb * hw;
        const int y = r / p.W_, x = r - y * p.W_;
        const size_t dest = ((size_t)(bb * 2 * p.H + 2 * y + dy) * (2 * p.W_) + 2 * x + dx);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + dest * p.ldo + co) = acc + b;
    }
}

// dynamic LDS of a launch: two K-tile buffers, or the epilogue's staging area where that is larger (the f32 residual staging of the
// unpadded DMA buffers: 68 KiB at BN = 128 -- still two workgroups per CU)
#ifndef UF_GEMM_AUX_EARLY
#define UF_GEMM_AUX_EARLY 0
#endif
template <typename T, int BN, int WGM, int WGN, int EP, bool DMA>
constexpr int gemm_smem_bytes() {
    constexpr int two = 2 * (BM + BN) * (DMA ? 128 : ROWB_PAD);
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int stgf = (EP == E_RES || EP == E_RES_WINREV) ? 4 * WTM * (WTN * 4 + 16) : 0;
    constexpr int a = WTM * (WTN * (int)sizeof(T) + 16), b = WTN * (WTM * (int)sizeof(T) + 16);
    constexpr int stg = (EP == E_STORE_T || EP == E_STORE_T_GELU || EP == E_QKV || EP == E_STORE_T_PRE_GELU || EP == E_STORE_T_MUL_DGELU) ? 4 * (a > b ? a : b) : 0;
    constexpr int need = stgf > stg ? stgf : stg;
    return two > need ? two : need;
}

template <typename T, int BN, int WGM, int WGN, int AL, int EP, bool DMA = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;          // elements per 128-byte K row
    constexpr int KSTEPS = BK / 32;      // MFMA k-steps (32 elements) per tile
    constexpr int TM = BM / WGM / 16, TN = BN / WGN / 16;
    constexpr int A_CH = BM / 32, W_CH = BN / 32;  // 16-byte chunks staged per thread
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(!DMA || (sizeof(T) == 2 && AL == A_PLAIN), "LDS-DMA staging: plain loader, 2-byte operand types");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int ROWB = DMA ? 128 : ROWB_PAD;       // LDS row stride of a K tile
    constexpr int BUF_BYTES = (BM + BN) * ROWB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> scalar branches / addresses
    const int ccol = tid & 7, crow = tid >> 3;  // staging: chunk column (0..7), first row (0..31)
    // XCD-aware tile order.  Workgroup b runs on XCD b % 8 (each XCD has a private 4 MiB L2), so the
    // N-tiles of one M-tile are given consecutive sequence numbers ON ONE XCD: the activation tile is
    // fetched from HBM once and re-read by its N/BN column tiles from that XCD's L2.  Placement only
    // affects speed, never results.
    const int n_tiles = (p.N + BN - 1) / BN;
    const int seq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    // Downsample (im2col loader): output rows oy and oy + 1 share two of their four input rows, so NEIGHBOURING M tiles read the same
    // tokens -- interleaved over the XCDs (mt % 8) each L2 fetched them for itself: 1.83-2.05 x the algorithmic HBM traffic
    // (profiles/r03_pmc_traffic.json).  Every XCD therefore walks one CONTIGUOUS run of M tiles here.
    const int mt = AL == A_CONV_DOWN ? xcd * ((int)gridDim.x / (8 * n_tiles)) + seq / n_tiles : (seq / n_tiles) * 8 + xcd;
    if (mt * BM >= p.M) return;
    const int m0 = mt * BM, n0 = (seq % n_tiles) * BN;
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, fg = lane >> 4;
    const bool vwave = EP == E_QKV && (n0 + wn * TN * 16) >= 2 * p.heads * p.hd;  // wave-uniform

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    AChunk<T, AL> ra[A_CH];
    WChunk<T> rw[W_CH];
    const int nt = (p.K + BK - 1) / BK;

    // E_STORE_T_MUL_DGELU, -DUF_GEMM_AUX_EARLY=1: the pre-activation chunks this lane multiplies into its copy-out rows requested HERE, before
    // the main loop, instead of in the epilogue.  Measured on the nine Uformer-B stage shapes at batch 32 (profiles/r04_run12.txt): 7127 us
    // against 7089 us in total, slower at six shapes (the 32 extra live registers cost more than the hidden round trip) -- not the default.
    constexpr int AUX_CPR = (BN / WGN) * (int)sizeof(T) / 16, AUX_RPI = 64 / AUX_CPR, AUX_NIT = (BM / WGM) / AUX_RPI;
    u32x4 auxv[EP == E_STORE_T_MUL_DGELU ? AUX_NIT : 1];
    if constexpr (EP == E_STORE_T_MUL_DGELU && UF_GEMM_AUX_EARLY) {
#pragma unroll
        for (int it = 0; it < AUX_NIT; ++it) {
            int m = m0 + wm * (BM / WGM) + it * AUX_RPI + lane / AUX_CPR, n = n0 + wn * (BN / WGN) + (lane % AUX_CPR) * EPC;
            m = m < p.M ? m : p.M - 1;
            n = n < p.N ? n : p.N - EPC;
            auxv[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.aux) + (size_t)m * p.ldo + n);
        }
    }

    auto g_load = [&](int t) {
        const int k = t * BK + ccol * EPC;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) ra[i].issue(p, m0 + crow + 32 * i, k);
#pragma unroll
        for (int i = 0; i < W_CH; ++i) rw[i].issue(p, n0 + crow + 32 * i, k);
    };
    auto s_store = [&](int buf) {
        char* As = smem + buf * BUF_BYTES;
        char* Ws = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *reinterpret_cast<u32x4*>(As + (crow + 32 * i) * ROWB + ccol * 16) = ra[i].finish();
#pragma unroll
        for (int i = 0; i < W_CH; ++i) *reinterpret_cast<u32x4*>(Ws + (crow + 32 * i) * ROWB + ccol * 16) = rw[i].finish();
    };

    // LDS-DMA staging: wave w moves rows [32 w, 32 w + 32) of the activation tile (4 instructions of 8 rows x 128 bytes) and rows
    // [BN/4 w, ..) of the weight tile; lane l of an instruction lands at row 8 q + l / 8, position l % 8 and therefore fetches chunk
    // (l % 8) ^ (row % 8).  Buffer descriptors bound both operands: rows past M / N read zeros.
    u32x4 rsa = {0, 0, 0, 0}, rsw = {0, 0, 0, 0};
    unsigned voa[A_CH], vow[W_CH];
    unsigned lds0 = 0;
    if constexpr (DMA) {
        lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        const unsigned long long aa = (unsigned long long)(uintptr_t)p.A, wa = (unsigned long long)(uintptr_t)p.W;
        rsa = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)aa), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(aa >> 32)) & 0xffffu,
                    (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(p.M - 1) * (unsigned)p.lda + (unsigned)p.K) * 2u)), 0x00020000u};
        rsw = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wa), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wa >> 32)) & 0xffffu,
                    (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)p.N * (unsigned)p.K * 2u)), 0x00020000u};
#pragma unroll
        for (int q = 0; q < A_CH; ++q) {
            const int r = wave * 32 + q * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
            voa[q] = (m0 + r) < p.M ? ((unsigned)(m0 + r) * (unsigned)p.lda + (unsigned)c * 8u) * 2u : 0xffffff00u;
        }
#pragma unroll
        for (int q = 0; q < W_CH; ++q) {
            const int r = wave * (BN / 4) + q * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
            vow[q] = (n0 + r) < p.N ? ((unsigned)(n0 + r) * (unsigned)p.K + (unsigned)c * 8u) * 2u : 0xffffff00u;
        }
    }
    auto dma_tile = [&](int t, int buf) {
        const unsigned soff = (unsigned)t * 128u, base = lds0 + (unsigned)buf * BUF_BYTES;
#pragma unroll
        for (int q = 0; q < A_CH; ++q) dma_buffer_to_lds(rsa, voa[q], soff, base + (unsigned)(wave * 32 + q * 8) * 128u);
#pragma unroll
        for (int q = 0; q < W_CH; ++q) dma_buffer_to_lds(rsw, vow[q], soff, base + (unsigned)(BM * 128 + (wave * (BN / 4) + q * 8) * 128));
    };

    if constexpr (DMA) {
        dma_tile(0, 0);
        wait_dma<0>();
    } else {
        g_load(0);
        s_store(0);
    }
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) {
            if constexpr (DMA) dma_tile(t + 1, buf ^ 1);
            else g_load(t + 1);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the next tile's global loads ABOVE this tile's MFMAs
        const int ra0 = wm * TM * 16 + fr, rw0 = wn * TN * 16 + fr;    // first fragment rows of this lane (16 j / 16 i more per tile: same r & 7)
        const char* As = smem + buf * BUF_BYTES + ra0 * ROWB + (DMA ? 0 : fg * (8 * (int)sizeof(T)));
        const char* Ws = smem + buf * BUF_BYTES + BM * ROWB + rw0 * ROWB + (DMA ? 0 : fg * (8 * (int)sizeof(T)));
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            Frag<T> af[TM], wf[TN];
            // DMA path: chunk (4 ks + fg) of row R sits at position (4 ks + fg) ^ (R & 7)
            const int ca = DMA ? (((ks * 4 + fg) ^ (ra0 & 7)) * 16) : ks * 64, cw = DMA ? (((ks * 4 + fg) ^ (rw0 & 7)) * 16) : ks * 64;
#pragma unroll
            for (int j = 0; j < TM; ++j) load_frag(af[j], reinterpret_cast<const T*>(As + j * 16 * ROWB + ca));
#pragma unroll
            for (int i = 0; i < TN; ++i) load_frag(wf[i], reinterpret_cast<const T*>(Ws + i * 16 * ROWB + cw));
            // QKV: waves that own columns of the V third compute their tiles TRANSPOSED (activation as
            // the MFMA A operand): a lane then holds 4 consecutive TOKENS of one channel, which is
            // contiguous in the V^T layout the attention kernel reads.  Wave-uniform branch.
            if (EP == E_QKV && vwave) {
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) mma16(acc[i][j], af[j], wf[i]);
            } else {
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) mma16(acc[i][j], wf[i], af[j]);
            }
        }
        if (t + 1 < nt) {
            if constexpr (DMA) wait_dma<0>();
            else s_store(buf ^ 1);
        }
        __syncthreads();
    }

    constexpr bool STAGED = (EP == E_STORE_T || EP == E_STORE_T_GELU || EP == E_QKV || EP == E_STORE_T_PRE_GELU || EP == E_STORE_T_MUL_DGELU);
    if constexpr (STAGED) {
        // T-typed outputs: stage the wave's tile in LDS (the main-loop buffers are free after the
        // last barrier) and copy it out in 16-byte chunks so that every store instruction writes whole
        // 64-256 byte runs of the destination rows instead of 8 bytes per lane.
        constexpr int WTM = TM * 16, WTN = TN * 16;
        constexpr int SROW = WTN * (int)sizeof(T) + 16;   // staged row stride, [m][n] orientation
        constexpr int SROWT = WTM * (int)sizeof(T) + 16;  // transposed [n][m] orientation (V^T)
        constexpr int STG = (WTM * SROW > WTN * SROWT) ? WTM * SROW : WTN * SROWT;
        static_assert(4 * STG <= gemm_smem_bytes<T, BN, WGM, WGN, EP, DMA>(), "staging does not fit the kernel's LDS");
        static_assert(DMA || sizeof(T) != 2 || 4 * STG <= BUF_BYTES, "2-byte types: the staging must fit ONE buffer (single-buffer launches when K is one tile)");
        char* stg = smem + wave * STG;
        const int mw0 = m0 + wm * WTM, nw0 = n0 + wn * WTN;
        const int C = p.heads * p.hd;
        if (!vwave) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int n = nw0 + i * 16 + fg * 4;
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + (n < p.N ? n : p.N - 4));  // clamped, unconditional
                const float sc = (EP == E_QKV && n < C) ? p.qscale : 1.0f;  // q = q * scale (model.py:497)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    f32x4 v = acc[i][j] + b;
                    if constexpr (EP == E_STORE_T_GELU) {
                        gelu4<T>(v);
                    }
                    if constexpr (EP == E_QKV) v *= sc;
                    store4(reinterpret_cast<T*>(stg + (j * 16 + fr) * SROW) + i * 16 + fg * 4, v);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int n = nw0 + i * 16 + fr;
                const float b = p.bias[n < p.N ? n : p.N - 1];
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const f32x4 v = {acc[i][j][0] + b, acc[i][j][1] + b, acc[i][j][2] + b, acc[i][j][3] + b};
                    store4(reinterpret_cast<T*>(stg + (i * 16 + fr) * SROWT) + j * 16 + fg * 4, v);
                }
            }
        }
        // E_STORE_T_MUL_DGELU: the pre-activation chunks of ALL copy-out iterations are requested here, unconditionally from clamped
        // addresses, before the barrier -- inside the guarded loop below each one was a dependent HBM round trip per iteration
        constexpr int CPR_ = WTN * (int)sizeof(T) / 16, RPI_ = 64 / CPR_, NIT_ = WTM / RPI_;
        static_assert(EP != E_STORE_T_MUL_DGELU || (CPR_ == AUX_CPR && NIT_ == AUX_NIT), "aux chunk mapping");
        if constexpr (EP == E_STORE_T_MUL_DGELU && !UF_GEMM_AUX_EARLY) {
#pragma unroll
            for (int it = 0; it < NIT_; ++it) {
                int m = mw0 + it * RPI_ + lane / CPR_, n = nw0 + (lane % CPR_) * EPC;
                m = m < p.M ? m : p.M - 1;
                n = n < p.N ? n : p.N - EPC;
                auxv[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.aux) + (size_t)m * p.ldo + n);
            }
        }
        __syncthreads();
        if (!vwave) {
            constexpr int CPR = WTN * (int)sizeof(T) / 16, RPI = 64 / CPR;
#pragma unroll
            for (int it = 0; it < WTM / RPI; ++it) {
                const int r = it * RPI + lane / CPR, cb = lane % CPR;
                const int m = mw0 + r, n = nw0 + cb * EPC;
                if (m < p.M && n < p.N) {
                    const u32x4 val = *reinterpret_cast<const u32x4*>(stg + r * SROW + cb * 16);
                    T* dst;
                    if constexpr (EP == E_QKV) {
                        const int which = n / C, c = n - which * C;
                        const int h = c / p.hd, d = c - h * p.hd;
                        dst = reinterpret_cast<T*>(which == 0 ? p.q : p.k) +
                              (((size_t)(m >> 6) * p.heads + h) * 64 + (m & 63)) * p.hd + d;
                    } else {
                        dst = reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n;
                    }
                    if constexpr (EP == E_STORE_T_PRE_GELU) {       // the activation of the value AS STORED (what a separate pass would read)
                        float f[EPC];
                        unpack_chunk<T>(val, f);
                        gelu_n<T, EPC>(f);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.aux) + (size_t)m * p.ldo + n) = pack_chunk<T>(f);
                    }
                    if constexpr (EP == E_STORE_T_MUL_DGELU) {      // dy (rounded to T, as a separate pass would read it) * GELU'(a)
                        float f[EPC], a[EPC];
                        unpack_chunk<T>(val, f);
                        unpack_chunk<T>(auxv[it], a);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) f[e] *= gelu_grad_t<T>(a[e]);
                        *reinterpret_cast<u32x4*>(dst) = pack_chunk<T>(f);
                    } else
                    *reinterpret_cast<u32x4*>(dst) = val;
                }
            }
        } else {
            constexpr int CPR = WTM * (int)sizeof(T) / 16, RPI = 64 / CPR;
#pragma unroll
            for (int it = 0; it < WTN / RPI; ++it) {
                const int r = it * RPI + lane / CPR, cb = lane % CPR;
                const int n = nw0 + r, m = mw0 + cb * EPC;
                if (m < p.M && n < p.N) {
                    const u32x4 val = *reinterpret_cast<const u32x4*>(stg + r * SROWT + cb * 16);
                    const int c = n - 2 * C, h = c / p.hd, d = c - h * p.hd;
                    T* dst = reinterpret_cast<T*>(p.vt) + (((size_t)(m >> 6) * p.heads + h) * p.hd + d) * 64 + (m & 63);
                    *reinterpret_cast<u32x4*>(dst) = val;
                }
            }
        }
    } else if constexpr (EP == E_RES || EP == E_RES_WINREV) {
        // f32 residual outputs: the wave's tile goes through LDS as f32 and leaves row by row, so the residual loads and the stores are
        // whole 128-256 byte runs of the token rows (as 16 bytes per lane at 16 different rows they were 64-byte pieces: the residual
        // forms of the projection / linear2 GEMMs cost 3.4 ms more per training step than the same products with a plain store)
        constexpr int WTM = TM * 16, WTN = TN * 16;
        constexpr int SROWF = WTN * 4 + 16;
        constexpr int STGF = WTM * SROWF;
        static_assert(4 * STGF <= gemm_smem_bytes<T, BN, WGM, WGN, EP, DMA>(), "f32 staging does not fit the kernel's LDS");
        char* stg = smem + wave * STGF;
        const int mw0 = m0 + wm * WTM, nw0 = n0 + wn * WTN;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const f32x4 b = epilogue_bias<EP>(p, nw0 + i * 16 + fg * 4);
#pragma unroll
            for (int j = 0; j < TM; ++j) *reinterpret_cast<f32x4*>(stg + (j * 16 + fr) * SROWF + (i * 16 + fg * 4) * 4) = acc[i][j] + b;
        }
        constexpr int CPR = WTN / 4, RPI = 64 / CPR, NIT = WTM / RPI;     // 16-byte chunks per row, rows per pass
        const int r0 = lane / CPR, cb = lane % CPR;
        const int n = nw0 + cb * 4;
        const bool nok = n < p.N;
        const int ncl = nok ? n : p.N - 4;
        // token rows, residual values and DropPath factors of every pass requested up front, from clamped addresses
        int tok[NIT];
        f32x4 res[NIT];
        float sc[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int m = mw0 + it * RPI + r0;
            const int mc = m < p.M ? m : p.M - 1;
            tok[it] = EP == E_RES_WINREV ? window_row_to_token(mc, p.H, p.W_, p.shift) : mc;
            res[it] = *reinterpret_cast<const f32x4*>(p.resid + (size_t)tok[it] * p.ldr + ncl);
            sc[it] = p.scale ? p.scale[tok[it] / p.hw] : 1.0f;   // training: x + DropPath(branch), bernoulli(keep) / keep per sample
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = it * RPI + r0;
            if (mw0 + r < p.M && nok) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * SROWF + cb * 16);
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)tok[it] * p.ldo + n) = res[it] + v * sc[it];
            }
        }
    } else {
        f32x4 bv[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) bv[i] = epilogue_bias<EP>(p, n0 + (wn * TN + i) * 16 + fg * 4);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + (wn * TN + i) * 16 + fg * 4;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m0 + (wm * TM + j) * 16 + fr;
                if (m < p.M && n < p.N) epilogue<T, EP>(p, m, n, acc[i][j], bv[i]);
            }
        }
    }
}

template <typename T, int BN, int WGM, int WGN, int AL, int EP, bool DMA>
int launch_kern(const GemmParams& p, hipStream_t stream) {
    constexpr int smem2 = gemm_smem_bytes<T, BN, WGM, WGN, EP, DMA>();
    auto kern = gemm_kernel<T, BN, WGM, WGN, AL, EP, DMA>;
    static bool lds_done[64] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem2, lds_done, "gemm")) return rc;
    // K within ONE tile (K <= 64 for the 2-byte types: every projection of the 32- and 64-channel stages, the ones with the most tokens):
    // the second LDS buffer is never touched, and without it 3-4 workgroups fit a CU instead of 2 -- these launches are streams of
    // short load -> MFMA -> store chains, bound by how many of them are in flight.  (The staged epilogue of the 2-byte types fits one
    // buffer; the f32 one and the f32 staging of the residual stores do not.)
    constexpr int BKE = 8 * (16 / (int)sizeof(T));
    const int smem = (!DMA && sizeof(T) == 2 && EP != E_RES && EP != E_RES_WINREV && p.K <= BKE) ? smem2 / 2 : smem2;   // (the f32 staging of the residual stores needs both)
    const int m_tiles = (p.M + BM - 1) / BM, n_tiles = (p.N + BN - 1) / BN;
    dim3 grid((unsigned)(((m_tiles + 7) / 8) * 8 * n_tiles));
    char name[96] = "";
    if (timing_enabled())
        snprintf(name, sizeof(name), "gemm_%s_bn%d_a%d_e%d%s %dx%dx%d", TypeName<T>::s, BN, AL, EP, DMA ? "_dma" : "", p.M, p.N, p.K);
    const double sz = sizeof(T), mn = (double)p.M * p.N, mk = (double)p.M * p.K;
    const double a_bytes = AL == A_PLAIN ? mk * sz : (AL == A_FROM_R ? mk * 4 : mk);  // conv-down reads each input once
    const double o_bytes = (EP == E_RES || EP == E_RES_WINREV) ? mn * 8 : ((EP == E_STORE_R || EP == E_UPSAMPLE) ? mn * 4 : mn * sz);
    {
        ScopedTimer tm(name, 2.0 * mn * p.K, a_bytes + (double)p.N * p.K * sz + o_bytes, stream);
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
    }
    return check_launch("gemm");
}

// LDS-DMA staging where it pays (see the file comment): plain loader, 2-byte type, K a whole number of 64-element tiles, both operands
// addressable with 32-bit byte offsets, and a K-heavy product -- measured on the nine stage shapes at batch 32 (profiles/r04_run2.txt,
// scripts/ubench_train.py gemm, register-staged -> DMA): K >= 512 (dec0 / bottleneck level: dz 777 -> 1064 TFLOP/s, linear2 + residual
// 712 -> 882, dxn 788 -> 975, qkv 656 -> 712, dO 627 -> 703; dz / dxn of dec1 666 -> 701 / 680 -> 784) and the wide K = 256 products
// (linear1 of dec1, N = 4 K: 340 -> 383) gain; the narrow K <= 256 ones (dO 510 -> 475 at C = 256, qkv 304 -> 290 at C = 128) lose a
// little: with 2-4 K tiles the second workgroup of a CU hides the staging stores as well as the DMA does, and the DMA path's prologue
// waits for its first tile with nothing else in flight.  UF_VARIANT="gemm_dma=0" / "gemm_dma=1" force the register / DMA path wherever it applies
// (A/B runs, bit-identity test).
template <typename T, int BN, int WGM, int WGN, int AL, int EP>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
    if constexpr (sizeof(T) == 2 && AL == A_PLAIN) {
        const int e = variant("gemm_dma", -1);      // UF_VARIANT="gemm_dma=0|1", read per call: the bit-identity test flips it inside one process
        const bool can = p.K % 64 == 0 && p.K >= 128 && (long long)p.M * p.lda * 2 < 0xffffff00LL && (long long)p.N * p.K * 2 < 0xffffff00LL;
        const bool pays = p.K >= 512 || (p.K >= 256 && p.N >= 4 * p.K);
        const bool on = e >= 0 ? e != 0 : pays;
        if (can && on) return launch_kern<T, BN, WGM, WGN, AL, EP, true>(p, stream);
    }
    return launch_kern<T, BN, WGM, WGN, AL, EP, false>(p, stream);
}

template <typename T, int AL, int EP>
int launch_bn(const GemmParams& p, hipStream_t stream) {
    if (p.N <= 32) return launch_cfg<T, 32, 4, 1, AL, EP>(p, stream);
    if (p.N <= 64) return launch_cfg<T, 64, 2, 2, AL, EP>(p, stream);
    // deep levels (few tokens, long K): 128x128 tiles leave CUs idle (M=4096, N=512 -> 128 blocks); halve the tile width
    // until there are two blocks per CU
    const long long tiles128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (EP != E_QKV && tiles128 < 512) return launch_cfg<T, 64, 2, 2, AL, EP>(p, stream);
    return launch_cfg<T, 128, 2, 2, AL, EP>(p, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Downsample, second form (round 6; 2-byte operand types, C = 32 / 64 / 128 / 256, output maps that are whole tiles).
// The im2col loader above gathers every K tile of the activation operand from HBM / L2 again for every N tile (address arithmetic, two f32 loads, a
// conversion and a 16-byte LDS store per chunk), and an input pixel belongs to four patches: at 16 x 16 (M = 4096, N = 512, K = 4096) the launch ran at
// 230 TFLOP/s, bound by the loader's own instruction stream (three K tiles in flight instead of one changed nothing: profiles/r06_run26_down.txt).
// Here a workgroup owns TOY x TOX OUTPUT pixels, stages the (2 TOY + 2) x (2 TOX + 2) input pixels under them ONCE -- converted to the operand type,
// zero outside the image -- and every MFMA operand fragment of every tap is a 16-byte LDS read at (2 oy + ky, 2 ox + kx).  The four waves split the BN
// output channels; weight fragments stream L2 -> registers through a ring of RING k-steps (no LDS, no barrier in the K loop) -- from the FRAGMENT-MAJOR pack
// (uf_pack_weight_fm: one contiguous KiB per wave-instruction; FM = true, uf_downsample_fm_fwd) or from the row-major T[N][K] (16 rows x 64 bytes per
// wave-instruction: every 128-byte line is asked for twice, a k-step apart, and at C >= 128 it has left the L1 in between -- 42 / 67 us against
// 24.5 / 28 us at C = 128 / 256, profiles/r06_run29_fm.txt).
// Pixel pitch = 2 C + 16 bytes: two neighbouring output pixels are 2 pitches = 32 (C = 32: 160) bytes mod 256 apart, so the 16 lanes of a fragment read
// hit 16 different 16-byte bank groups; with 8-pixel output rows (C = 256) the row pitch is padded to a multiple of 128 bytes for the same reason.
// Same K order (tap-major, 32-channel steps), same MFMAs, accumulators from zero, bias added at the end: bit-identical to the first form
// (tests/test_gpu_ops.py::test_downsample_forms_bit_identical; UF_VARIANT="down=1" selects the first form).
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int C, int TOY, int TOX, int BN, int RING, bool FM>
__global__ __launch_bounds__(256, C == 32 ? 3 : (C == 64 && TOY == 4 ? 3 : 1)) void down_patch_kernel(const GemmParams p, int tiles_x, int tiles_y) {
    static_assert(sizeof(T) == 2 && (TOX == 16 || TOX == 8) && BN % 64 == 0, "2-byte operand types; 16- or 8-pixel tile rows; four waves split BN");
    constexpr int PH = 2 * TOY + 2, PW = 2 * TOX + 2;
    constexpr int PP = C * 2 + 16;                                              // pixel pitch in LDS (bytes)
    constexpr int RP = TOX == 8 ? (PW * PP + 127) / 128 * 128 : PW * PP;        // row pitch
    constexpr int TM = TOY * TOX / 16, TN = BN / 64;                            // 16-token tiles of the workgroup, 16-channel tiles of a wave
    constexpr int K = 16 * C, KSN = K / 32, CS = C / 32;                         // k-steps in all, k-steps per tap
    static_assert(KSN % RING == 0, "ring divides the k-steps");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int n_tiles = p.N / BN;
    const int lid = xcd_tile((int)blockIdx.x, (int)gridDim.x);                  // every XCD walks one contiguous run of (tile, N tile) pairs: halo pixels and N tiles share an L2
    const int tile = lid / n_tiles, n0 = (lid - tile * n_tiles) * BN + wave * (16 * TN);
    const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
    const int oy0 = (tr / tiles_x) * TOY, ox0 = (tr % tiles_x) * TOX;
    const int H = p.H, W = p.W_;
    const T* Wt = reinterpret_cast<const T*>(p.W);
    const T* Wf = reinterpret_cast<const T*>(p.W_fm);
    constexpr int KSTR = FM ? 512 : 32;                                          // elements between the k-steps of a lane's fragment stream

    // weight ring: the first RING - 1 k-steps are requested before the patch is staged
    const T* wrow[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) wrow[i] = FM ? Wf + ((size_t)(n0 / 16 + i) * KSN * 64 + lane) * 8 : Wt + (size_t)(n0 + 16 * i + fr) * K + fg * 8;
    Frag<T> wf[RING][TN];
#pragma unroll
    for (int r = 0; r < RING - 1; ++r)
#pragma unroll
        for (int i = 0; i < TN; ++i) load_frag(wf[r][i], wrow[i] + r * KSTR);

    // ---- stage the input patch: chunk = 8 channels of one pixel (32 bytes of f32 in, 16 bytes of T out), eight chunks per thread in flight
    {
        constexpr int CPP = C / 8, NCH = PH * PW * CPP, U = 8;
        const float* xb = reinterpret_cast<const float*>(p.A);
        const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
#pragma unroll 1
        for (int q0 = tid; q0 < NCH; q0 += 256 * U) {
            u32x4 lo[U], hi[U];
            bool ok[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + 256 * u, qc = q < NCH ? q : NCH - 1;
                const int pix = qc / CPP, cc = qc - pix * CPP, pr = pix / PW, pc = pix - pr * PW;
                const int iy = iy0 + pr, ix = ix0 + pc;
                ok[u] = q < NCH && iy >= 0 && iy < H && ix >= 0 && ix < W;
                const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
                const float* src = xb + ((size_t)(b * H + iyc) * W + ixc) * p.lda + cc * 8;
                lo[u] = *reinterpret_cast<const u32x4*>(src);
                hi[u] = *(reinterpret_cast<const u32x4*>(src) + 1);
                dst[u] = q < NCH ? pr * RP + pc * PP + cc * 16 : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                u32x4 v = u32x4{pack2<T>(__uint_as_float(lo[u][0]), __uint_as_float(lo[u][1])), pack2<T>(__uint_as_float(lo[u][2]), __uint_as_float(lo[u][3])),
                                pack2<T>(__uint_as_float(hi[u][0]), __uint_as_float(hi[u][1])), pack2<T>(__uint_as_float(hi[u][2]), __uint_as_float(hi[u][3]))};
                if (!ok[u]) v = u32x4{0, 0, 0, 0};
                if (dst[u] >= 0) *reinterpret_cast<u32x4*>(smem + dst[u]) = v;
            }
        }
    }
    __syncthreads();

    // per-lane patch offsets of the lane's token in each 16-token tile (tap (0, 0), channel group fg)
    int abase[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int oyl = TOX == 16 ? j : 2 * j + (fr >> 3), oxl = TOX == 16 ? fr : (fr & 7);
        abase[j] = 2 * oyl * RP + 2 * oxl * PP + fg * 16;
    }
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fully unrolled: across a loop back-edge the compiler's wait-count insertion falls back to "everything but the last k-step's loads has landed" at the top
    // of every turn (s_waitcnt vmcnt(TN) in the ISA), which drains the ring once per RING k-steps
#pragma unroll
    for (int ks0 = 0; ks0 < KSN; ks0 += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            const int ks = ks0 + r;
            if (ks + RING - 1 < KSN) {
#pragma unroll
                for (int i = 0; i < TN; ++i) load_frag(wf[(r + RING - 1) % RING][i], wrow[i] + (ks + RING - 1) * KSTR);
            }
            const int tap = ks / CS, koff = (tap >> 2) * RP + (tap & 3) * PP + (ks - tap * CS) * 64;     // wave-uniform
            Frag<T> af[TM];
#pragma unroll
            for (int j = 0; j < TM; ++j) load_frag(af[j], reinterpret_cast<const T*>(smem + abase[j] + koff));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) mma16(acc[i][j], wf[r][i], af[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    const int Ho = H >> 1, Wo = W >> 1;
    float* out = reinterpret_cast<float*>(p.out);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int n = n0 + 16 * i + 4 * fg;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int oyl = TOX == 16 ? j : 2 * j + (fr >> 3), oxl = TOX == 16 ? fr : (fr & 7);
            const size_t m = (size_t)(b * Ho + oy0 + oyl) * Wo + ox0 + oxl;
            *reinterpret_cast<f32x4*>(out + m * p.ldo + n) = acc[i][j] + bv;
        }
    }
}

template <typename T, int C, int TOY, int TOX, int BN, int RING, bool FM>
int launch_down_patch_fm(const GemmParams& p, hipStream_t stream) {
    constexpr int PW = 2 * TOX + 2, PP = C * 2 + 16, RP = TOX == 8 ? (PW * PP + 127) / 128 * 128 : PW * PP;
    constexpr int smem = (2 * TOY + 2) * RP;
    auto kern = down_patch_kernel<T, C, TOY, TOX, BN, RING, FM>;
    static bool lds_done[64] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, lds_done, "downsample")) return rc;
    const int Ho = p.H / 2, Wo = p.W_ / 2, tiles_x = Wo / TOX, tiles_y = Ho / TOY, B = p.M / (Ho * Wo);
    char name[96] = "";
    if (timing_enabled()) snprintf(name, sizeof(name), "down_patch_%s_c%d%s %dx%dx%d", TypeName<T>::s, C, FM ? "_fm" : "", p.M, p.N, p.K);
    {
        ScopedTimer tm(name, 2.0 * p.M * (double)p.N * p.K, (double)p.M * p.K + (double)p.N * p.K * sizeof(T) + (double)p.M * p.N * 4, stream);
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * tiles_x * tiles_y * (p.N / BN))), dim3(256), smem, stream, p, tiles_x, tiles_y);
    }
    return check_launch("downsample");
}

template <typename T, int C, int TOY, int TOX, int BN, int RING>
int launch_down_patch(const GemmParams& p, hipStream_t stream) {
    return p.W_fm ? launch_down_patch_fm<T, C, TOY, TOX, BN, RING, true>(p, stream) : launch_down_patch_fm<T, C, TOY, TOX, BN, RING, false>(p, stream);
}

// the second form where it is built: 2-byte operands, C = 32 ... 256, N = 2 C, K = 16 C, whole tiles, rows of 4 f32 channels aligned
template <typename T>
int try_down_patch(const GemmParams& p, hipStream_t stream, bool* done) {
    *done = false;
    if constexpr (sizeof(T) == 2) {
        const int dv = variant("down", 0);                  // UF_VARIANT="down=1": the first form everywhere; "down=2": the second wherever it is built (tests, A/B runs)
        if (dv == 1) return UF_OK;
        const int C = p.C, Ho = p.H / 2, Wo = p.W_ / 2;
        if (p.N != 2 * C || p.K != 16 * C || p.lda % 4 || p.ldo % 4 || Ho <= 0 || Wo <= 0 || p.M % (Ho * Wo) || (p.H & 1) || (p.W_ & 1)) return UF_OK;
        if ((long long)p.M * p.ldo >= 0x7fffffffLL * 4) return UF_OK;
        *done = true;
        if (C == 32 && Ho % 8 == 0 && Wo % 16 == 0) return launch_down_patch<T, 32, 8, 16, 64, 8>(p, stream);
        // C = 64 with the fragment-major weight: 4 x 16 output pixels (a 49 KB patch: three workgroups per CU hide each other's staging round trips) -- 35 -> 29 us at
        // batch 16, 67 -> 51 at batch 32 (profiles/r06_run32_c64.txt); with the row-major weight the doubled weight stream costs what that wins
        if (C == 64 && p.W_fm && Ho % 4 == 0 && Wo % 16 == 0) return launch_down_patch<T, 64, 4, 16, 128, 8>(p, stream);
        if (C == 64 && Ho % 8 == 0 && Wo % 16 == 0) return launch_down_patch<T, 64, 8, 16, 128, 8>(p, stream);
        // C >= 128 with ROW-MAJOR weights: one workgroup per CU (the patch is 90 KB) streaming 1-2 MB of weights -- a win while the launch is ONE round of workgroups, slower than the
        // first form beyond it (batch 32: 83 vs 71 us at C = 128, 132 vs 103 at C = 256; batch 16: 42 vs 53, 67 vs 76; profiles/r06_run28_down.txt)
        if (C == 128 && Ho % 4 == 0 && Wo % 16 == 0 && (dv == 2 || p.W_fm || p.M / 64 <= 256)) return launch_down_patch<T, 128, 4, 16, 256, 4>(p, stream);
        if (C == 256 && Ho % 4 == 0 && Wo % 8 == 0 && (dv == 2 || p.W_fm || p.M / 32 * 2 <= 256)) return launch_down_patch<T, 256, 4, 8, 256, 8>(p, stream);
        *done = false;
    }
    return UF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Input gradient of Downsample on the recipe of down_patch_kernel (round 6; training, 2-byte operand types, Cin = 32 ... 256).
// dx = ConvTranspose2d(dy; k4 s2 p1): input pixel (2a + r, 2b + s) collects FOUR taps -- ky in {3, 1} (r = 0) or {2, 0} (r = 1) from output rows a - 1 + r, a + r,
// likewise kx -- so each of the four parity classes (r, s) is a GEMM over the (a, b) grid with K = 4 taps x Cout and N = Cin.  The first form is
// uf_linear_fwd (dy W -> the 16 Cin-wide patch matrix, rounded to T) + uf_col2im: the patch matrix goes through HBM twice (1 GB per step at batch 32).
// Here a workgroup owns TOY x TOX positions (a, b), stages the (TOY + 2) x (TOX + 2) pixels of dy around them in LDS once (zero outside the map), and
// computes the four classes from it: operand fragments are 16-byte LDS reads (pixel pitch 2 Cout + 32 bytes: neighbouring pixels hit different bank
// groups), the class's weight [Cin][4 Cout] streams fragment-major L2 -> registers (down_dx_pack_kernel builds the four class weights from the transposed
// packed weight).  Every tap's product is rounded to T and the taps are added in ascending (ky, kx), as the first form does: bit-identical to it.
// ------------------------------------------------------------------------------------------------------------------
struct DownDxParams {
    const void* dyT; int ld_dy;      // T[B * Ho * Wo][ld_dy]: the output gradient in the operand type
    const void* Wc;                  // fragment-major T[4][Cin][4 Cout]
    float* dx; int ld_dx;            // f32[B * H * W][ld_dx]
    int B, H, W, accumulate;         // H, W: the INPUT map
};

template <typename T>
__global__ __launch_bounds__(256) void down_dx_pack_kernel(const T* __restrict__ w_pk_t, T* __restrict__ out, int Ci, int Co) {
    // out[cls][n-tile][k-step][lane = fg * 16 + fr][8] = W_cls[n = 16 tile + fr][k = 32 ks + 8 fg ..], W_cls[ci][t Co + co] = w_pk_t[(ky(r, ty) 4 + kx(s, tx)) Ci + ci][co]
    const int KSN = (4 * Co) / 32, NT = Ci / 16;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)4 * NT * KSN * 64) return;
    const int lane = (int)(idx & 63), fr = lane & 15, fg = lane >> 4;
    long long rest = idx >> 6;
    const int ks = (int)(rest % KSN); rest /= KSN;
    const int nt = (int)(rest % NT), cls = (int)(rest / NT);
    const int r = cls >> 1, sx = cls & 1;
    const int k0 = ks * 32 + fg * 8, t = k0 / Co, co = k0 - t * Co, ty = 1 - (t >> 1), tx = 1 - (t & 1);       // taps in ascending (ky, kx): the order uf_col2im adds them in
    const int ky = r == 0 ? (ty == 0 ? 3 : 1) : (ty == 0 ? 2 : 0), kx = sx == 0 ? (tx == 0 ? 3 : 1) : (tx == 0 ? 2 : 0);
    const int n = nt * 16 + fr;
    *reinterpret_cast<u32x4*>(out + idx * 8) = *reinterpret_cast<const u32x4*>(w_pk_t + ((size_t)((ky * 4 + kx) * Ci + n)) * Co + co);
}

template <typename T, int CI, int TOY, int TOX, int RING>
__global__ __launch_bounds__(256, CI <= 64 ? 2 : 1) void down_dx_kernel(const DownDxParams p, int tiles_x, int tiles_y) {
    static_assert(sizeof(T) == 2 && TOX == 16, "2-byte operand types; 16-position tile rows");
    constexpr int CO = 2 * CI, PH = TOY + 2, PW = TOX + 2, PP = CO * 2 + 32, TOK = TOY * TOX;
    constexpr int WN = CI / 16 >= 4 ? 4 : CI / 16, WM = 4 / WN;             // waves along the input channels / along the positions
    constexpr int TN = CI / 16 / WN, TM = TOK / 16 / WM;
    constexpr int KST = CO / 32, KSN = 4 * KST;                             // k-steps per tap, per class
    static_assert(KSN % RING == 0 && TOK % (16 * WM) == 0, "ring / tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;
    const int tile = xcd_tile((int)blockIdx.x, (int)gridDim.x);
    const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
    const int a0 = (tr / tiles_x) * TOY, b0 = (tr % tiles_x) * TOX;
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const T* dy = reinterpret_cast<const T*>(p.dyT);
    const T* Wc = reinterpret_cast<const T*>(p.Wc);
    // ---- the dy pixels around the tile: 16-byte chunks (8 channels), all of a thread's loads in flight before its stores; zero outside the map
    {
        constexpr int CPP = CO / 8, NCH = PH * PW * CPP, PER = (NCH + 255) / 256;
        u32x4 v[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = q * 256 + tid, ic = idx < NCH ? idx : NCH - 1;
            const int pix = ic / CPP, cc = ic - pix * CPP, pr = pix / PW, pc = pix - pr * PW;
            const int oy = a0 - 1 + pr, ox = b0 - 1 + pc;
            const bool ok = oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
            const int oyc = oy < 0 ? 0 : (oy >= Ho ? Ho - 1 : oy), oxc = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
            v[q] = *reinterpret_cast<const u32x4*>(dy + ((size_t)(b * Ho + oyc) * Wo + oxc) * p.ld_dy + cc * 8);
            if (!ok) v[q] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = q * 256 + tid;
            if (idx < NCH) {
                const int pix = idx / CPP, cc = idx - pix * CPP;
                *reinterpret_cast<u32x4*>(smem + pix * PP + cc * 16) = v[q];
            }
        }
    }
    __syncthreads();
    int abase[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) abase[j] = ((wm * TM + j) * PW + fr) * PP + fg * 16;      // TOX = 16: position tile j = tile row, lane fr = column
#pragma unroll 1
    for (int cls = 0; cls < 4; ++cls) {
        const int r = cls >> 1, sx = cls & 1;
        const T* wrow[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) wrow[i] = Wc + (((size_t)cls * (CI / 16) + wn * TN + i) * KSN * 64 + lane) * 8;
        Frag<T> wf[RING][TN];
#pragma unroll
        for (int q = 0; q < RING - 1; ++q)
#pragma unroll
            for (int i = 0; i < TN; ++i) load_frag(wf[q][i], wrow[i] + q * 512);
        // one accumulator per tap, folded into the f32 sum ROUNDED TO T, taps in ascending (ky, kx): exactly what the patch-matrix route computes (its GEMM
        // stores every tap's product as T, uf_col2im adds them in that order) -- the two routes are bit-identical
        f32x4 acc[TN][TM], sum[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; sum[i][j] = acc[i][j]; }
        const int cbase = (r * PW + sx) * PP;
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
            if (ks + RING - 1 < KSN) {
#pragma unroll
                for (int i = 0; i < TN; ++i) load_frag(wf[(ks + RING - 1) % RING][i], wrow[i] + (ks + RING - 1) * 512);
            }
            const int t = ks / KST, koff = cbase + ((1 - (t >> 1)) * PW + (1 - (t & 1))) * PP + (ks - t * KST) * 64;
            Frag<T> af[TM];
#pragma unroll
            for (int j = 0; j < TM; ++j) load_frag(af[j], reinterpret_cast<const T*>(smem + abase[j] + koff));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) mma16(acc[i][j], wf[ks % RING][i], af[j]);
            __builtin_amdgcn_sched_barrier(0);
            if ((ks + 1) % KST == 0) {
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) {
                        float lo0, lo1, hi0, hi1;
                        unpack2<T>(pack2<T>(acc[i][j][0], acc[i][j][1]), lo0, lo1);
                        unpack2<T>(pack2<T>(acc[i][j][2], acc[i][j][3]), hi0, hi1);
                        sum[i][j] += f32x4{lo0, lo1, hi0, hi1};
                        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int ci = (wn * TN + i) * 16 + 4 * fg;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int a = a0 + wm * TM + j, bb = b0 + fr;
                float* o = p.dx + ((size_t)(b * p.H + 2 * a + r) * p.W + 2 * bb + sx) * p.ld_dx + ci;
                f32x4 v = sum[i][j];
                if (p.accumulate) v += *reinterpret_cast<const f32x4*>(o);
                *reinterpret_cast<f32x4*>(o) = v;
            }
        }
    }
}

template <typename T, int CI, int TOY, int RING>
int launch_down_dx_t(const DownDxParams& p, hipStream_t st) {
    constexpr int CO = 2 * CI, smem = (TOY + 2) * 18 * (CO * 2 + 32);
    auto kern = down_dx_kernel<T, CI, TOY, 16, RING>;
    static bool lds_done[64] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, lds_done, "downsample dx")) return rc;
    const int Ho = p.H / 2, Wo = p.W / 2, tiles_x = Wo / 16, tiles_y = Ho / TOY;
    char name[96] = "";
    if (timing_enabled()) snprintf(name, sizeof(name), "down_dx_%s_c%d %dx%dx%d", TypeName<T>::s, CI, p.B * p.H * p.W, CI, 4 * CO);
    {
        ScopedTimer tm(name, 2.0 * p.B * p.H * p.W * CI * 4.0 * CO, (double)p.B * Ho * Wo * CO * sizeof(T) + (double)p.B * p.H * p.W * CI * 4 * (p.accumulate ? 2 : 1), st);
        hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * tiles_x * tiles_y)), dim3(256), smem, st, p, tiles_x, tiles_y);
    }
    return check_launch("downsample dx");
}

template <typename T>
int launch_t(const GemmParams& p, int aload, int epi, hipStream_t stream) {
    if (aload == A_PLAIN) {
        switch (epi) {
            case E_STORE_T: return launch_bn<T, A_PLAIN, E_STORE_T>(p, stream);
            case E_STORE_T_GELU: return launch_bn<T, A_PLAIN, E_STORE_T_GELU>(p, stream);
            case E_QKV: return launch_bn<T, A_PLAIN, E_QKV>(p, stream);
            case E_STORE_T_PRE_GELU: return launch_bn<T, A_PLAIN, E_STORE_T_PRE_GELU>(p, stream);
            case E_STORE_T_MUL_DGELU: return launch_bn<T, A_PLAIN, E_STORE_T_MUL_DGELU>(p, stream);
            case E_RES_WINREV: return launch_bn<T, A_PLAIN, E_RES_WINREV>(p, stream);
            case E_RES: return launch_bn<T, A_PLAIN, E_RES>(p, stream);
            default: break;
        }
    } else if (aload == A_CONV_DOWN && epi == E_STORE_R) {
        bool done = false;
        const int rc = try_down_patch<T>(p, stream, &done);
        if (rc || done) return rc;
        return launch_bn<T, A_CONV_DOWN, E_STORE_R>(p, stream);
    } else if (aload == A_FROM_R && epi == E_UPSAMPLE) {
        return launch_bn<T, A_FROM_R, E_UPSAMPLE>(p, stream);
    }
    set_error("gemm: unsupported (aload=%d, epilogue=%d) combination", aload, epi);
    return UF_ERR_UNSUPPORTED;
}

}  // namespace

// Input gradient of Downsample from an LDS patch of dy (see down_dx_kernel).  Returns UF_OK with *done = false where the form is not built (the caller runs
// uf_linear_fwd + uf_col2im).  wc: scratch for the four class weights, 16 Cin Cout elements of T.
int launch_down_dx(const void* dyT, int ld_dy, const void* w_pk_t, void* wc, float* dx, int ld_dx, int B, int H, int W, int Cin, int Cout, int accumulate, uf_dtype dtype,
                   hipStream_t st, bool* done) {
    *done = false;
    const int Ho = H / 2, Wo = W / 2;
    if (!dtype_half(dtype) || Cout != 2 * Cin || (Cin != 32 && Cin != 64 && Cin != 128 && Cin != 256) || (H & 1) || (W & 1) || Wo % 16 || ld_dy % 8 || ld_dx % 4 ||
        variant("downdx", 2) == 1) return UF_OK;
    const int toy = Cin <= 64 ? 8 : 4;
    if (Ho % toy || ((uintptr_t)dyT % 16) || ((uintptr_t)dx % 16) || ((uintptr_t)wc % 16) || ((uintptr_t)w_pk_t % 16)) return UF_OK;
    *done = true;
    const long long groups = (long long)4 * (Cin / 16) * ((4 * Cout) / 32) * 64;
    DownDxParams p{dyT, ld_dy, wc, dx, ld_dx, B, H, W, accumulate};
#define UF_DDX(TT)                                                                                                                                        \
    hipLaunchKernelGGL(down_dx_pack_kernel<TT>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, (const TT*)w_pk_t, (TT*)wc, Cin, Cout);         \
    if (int rc = check_launch("downsample dx pack")) return rc;                                                                                           \
    switch (Cin) {                                                                                                                                        \
        case 32: return launch_down_dx_t<TT, 32, 8, 8>(p, st);                                                                                            \
        case 64: return launch_down_dx_t<TT, 64, 8, 8>(p, st);                                                                                            \
        case 128: return launch_down_dx_t<TT, 128, 4, 8>(p, st);                                                                                          \
        default: return launch_down_dx_t<TT, 256, 4, 4>(p, st);                                                                                           \
    }
    if (dtype == UF_BF16) { UF_DDX(bf16) }
    UF_DDX(f16)
#undef UF_DDX
}

int launch_gemm(const GemmParams& p, int aload, int epi, uf_dtype dtype, hipStream_t stream) {
    const int epc = dtype_half(dtype) ? 8 : 4;
    UF_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, UF_ERR_SHAPE, "gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    UF_REQUIRE(p.K % epc == 0 && p.N % 4 == 0, UF_ERR_SHAPE, "gemm: K=%d must be a multiple of %d and N=%d of 4", p.K, epc, p.N);
    UF_REQUIRE(p.A && p.W && p.bias, UF_ERR_NULL, "gemm: null operand");
    UF_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0 && ((uintptr_t)p.bias % 16) == 0,
               UF_ERR_ALIGN, "gemm: operands must be 16-byte aligned");
    if (aload == A_PLAIN) UF_REQUIRE(p.lda % epc == 0, UF_ERR_ALIGN, "gemm: lda=%d not a multiple of %d", p.lda, epc);
    else UF_REQUIRE(p.lda % 4 == 0, UF_ERR_ALIGN, "gemm: lda=%d not a multiple of 4", p.lda);
    if (epi == E_STORE_T_PRE_GELU || epi == E_STORE_T_MUL_DGELU) UF_REQUIRE(p.aux && ((uintptr_t)p.aux % 16) == 0, UF_ERR_NULL, "gemm: this epilogue needs a 16-byte aligned aux operand");
    if (epi == E_STORE_T || epi == E_STORE_T_GELU || epi == E_QKV || epi == E_STORE_T_PRE_GELU || epi == E_STORE_T_MUL_DGELU)
        UF_REQUIRE(p.N % epc == 0, UF_ERR_SHAPE, "gemm: N=%d must be a multiple of %d for this epilogue", p.N, epc);
    if (epi == E_QKV) {
        const int C = p.heads * p.hd;
        UF_REQUIRE(C == 16 || C % 32 == 0, UF_ERR_UNSUPPORTED, "qkv: C=%d must be 16 or a multiple of 32", C);
    }
    if (aload == A_CONV_DOWN) UF_REQUIRE(p.C % 8 == 0, UF_ERR_SHAPE, "downsample: C=%d must be a multiple of 8", p.C);
    if (dtype == UF_BF16) return launch_t<bf16>(p, aload, epi, stream);
    if (dtype == UF_F16) return launch_t<f16>(p, aload, epi, stream);
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

// training forms of the projection GEMM (uformer_amd/train.py):
//   uf_linear_pre_gelu_fwd: out = a = A W^T + bias AND act_out = GELU(a as stored)  -- linear1 of the LeFF keeps the pre-activation
//     for GELU' and the activation for the depthwise stencil, one pass instead of GEMM + uf_gelu_fwd      (model.py:657-658)
//   uf_linear_mul_dgelu: out = T(A W^T) * GELU'(pre)  -- the input gradient of a Linear whose INPUT came out of a GELU: the GEMM of
//     uf_linear_fwd(dy, W^T) with uf_gelu_bwd folded into its store (W here is the transposed weight, (N=K_layer, K=N_layer)).
extern "C" int uf_linear_pre_gelu_fwd(const void* A, const void* W, const float* bias, void* out, void* act_out, int M, int N, int K, uf_dtype dtype, void* stream) {
    uf::GemmParams p{};
    p.A = A; p.lda = K; p.W = W; p.bias = bias; p.M = M; p.N = N; p.K = K; p.out = out; p.ldo = N; p.aux = act_out;
    UF_REQUIRE(out && act_out, UF_ERR_NULL, "uf_linear_pre_gelu_fwd: null out");
    return uf::launch_gemm(p, uf::A_PLAIN, uf::E_STORE_T_PRE_GELU, dtype, (hipStream_t)stream);
}

extern "C" int uf_linear_mul_dgelu(const void* A, const void* W, const float* bias, const void* pre, void* out, int M, int N, int K, uf_dtype dtype, void* stream) {
    uf::GemmParams p{};
    p.A = A; p.lda = K; p.W = W; p.bias = bias; p.M = M; p.N = N; p.K = K; p.out = out; p.ldo = N; p.aux = const_cast<void*>(pre);
    UF_REQUIRE(out && pre, UF_ERR_NULL, "uf_linear_mul_dgelu: null pointer");
    return uf::launch_gemm(p, uf::A_PLAIN, uf::E_STORE_T_MUL_DGELU, dtype, (hipStream_t)stream);
}

// out f32[tok][N] = resid[tok][N] + scale[image of tok] * (A W^T + bias), tok = the token of window row m when `windowed` (window_reverse
// + roll back), else m: the projection / linear2 GEMM of a block with the residual add (and DropPath) in its store   (model.py:975-987)
extern "C" int uf_linear_residual_fwd(const void* A, const void* W, const float* bias, const float* resid, float* out, const float* scale, int B, int H, int Wd,
                                      int N, int K, int windowed, int shift, uf_dtype dtype, void* stream) {
    UF_REQUIRE(resid && out, UF_ERR_NULL, "uf_linear_residual_fwd: null pointer");
    UF_REQUIRE(B > 0 && H > 0 && Wd > 0 && (!windowed || (H % 8 == 0 && Wd % 8 == 0)), UF_ERR_SHAPE, "uf_linear_residual_fwd: B=%d H=%d W=%d", B, H, Wd);
    UF_REQUIRE(((uintptr_t)resid % 16) == 0 && ((uintptr_t)out % 16) == 0, UF_ERR_ALIGN, "uf_linear_residual_fwd: resid / out must be 16-byte aligned");
    uf::GemmParams p{};
    p.A = A; p.lda = K; p.W = W; p.bias = bias; p.M = B * H * Wd; p.N = N; p.K = K; p.out = out; p.ldo = N; p.resid = resid; p.ldr = N;
    p.H = H; p.W_ = Wd; p.shift = shift; p.scale = scale; p.hw = H * Wd;
    return uf::launch_gemm(p, uf::A_PLAIN, windowed ? uf::E_RES_WINREV : uf::E_RES, dtype, (hipStream_t)stream);
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
