// LayerNorm-fused, A-stationary GEMM for the two K = C projections of a LeWin block:
//
//   LN1 -> roll -> window_partition -> (+modulator) -> [to_q | to_kv]      (model.py:952-969, :431-442)
//   LN2 -> linear1 -> GELU                                                  (model.py:987, :657-658)
//
// One workgroup owns BM token rows.  Phase 0 reads those rows of the f32 residual stream ONCE
// (gathered through the roll/partition index for the norm1 path), normalises them in registers
// and leaves the bf16/f32 operand tile [BM][C] in LDS -- the normalised activations never touch
// HBM.  Phase 1 has no barriers: the four waves walk over 64x64 output units; a wave streams the
// weight fragments of its unit straight from L2 into registers (3-deep software ring, each
// weight row is read by exactly one wave of the block), reads activation fragments from LDS and
// issues 16 MFMAs per k-step.  The epilogue stages 16 rows at a time in a private LDS slab and
// writes whole 128/256-byte row segments (q, k, v^T in the attention layout, or the LeFF hidden).
#include <stdlib.h>

#include <type_traits>

#include "uf_internal.h"

namespace uf {
unsigned long long* debug_get_tbuf();
namespace {

struct LnGemmParams {
    const float* x; int ld;          // residual stream rows (f32)
    const float* gamma; const float* beta; const float* modulator;
    const void* Wt; const float* bias;  // T[N][C], f32[N]
    int M, N;
    int H, W, windowed, shift;
    void* out; int ldo;              // EP_GELU: T[M][ldo]
    void* q; void* k; void* vt; int heads, hd; float qscale;  // EP_QKV
    unsigned long long* tbuf;        // optional cycle stamps (uf_debug_set_tbuf)
};

enum { EP_QKV = 0, EP_GELU = 1 };

template <typename T, int C, int BM, int EP>
__global__ __launch_bounds__(256, 2) void ln_gemm_kernel(const LnGemmParams p) {
    constexpr int SZ = sizeof(T);
    constexpr int EPC = 16 / SZ;
    constexpr int SA = C * SZ + 16;               // LDS row stride of the operand tile
    constexpr int SS = 64 * SZ + 16;              // LDS row stride of a staging slab (16 rows x 64 values)
    constexpr int KS = (C + 31) / 32;             // MFMA k-steps
    constexpr int MH = BM / 64;                   // 64-row halves per block
    constexpr int RING = 3;                       // weight-fragment prefetch depth (k-steps)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> scalar branches / addresses
    const int fr = lane & 15, fg = lane >> 4;
    char* stg = smem + BM * SA + wave * (16 * SS);
    const int m0 = blockIdx.x * BM;
    Census census; census.begin();
    unsigned long long ts0 = __builtin_readcyclecounter(), ts_ln = 0, ts_k = 0, ts_e = 0, tsx;

    // ---------------- phase 0: LayerNorm (+gather, +modulator) into LDS -------------------------
    {
        constexpr int LPR = (C / 4) < 64 ? (C / 4) : 64;
        constexpr int V4 = C / (4 * LPR);
        constexpr int RPP = 256 / LPR;                 // rows normalised per pass of the block
        constexpr int NP = BM / RPP;                   // passes
        constexpr int U = (16 / V4) < NP ? (16 / V4) : NP;  // row groups kept in flight (16 x 16-byte loads per thread issued together)
        static_assert(NP % U == 0, "pass batching");
        const int sub = tid % LPR;
#pragma unroll 1
        for (int r0 = 0; r0 < BM; r0 += RPP * U) {
            f32x4 v[U][V4];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = m0 + r0 + u * RPP + tid / LPR;
                live[u] = m < p.M;
                const int mc = live[u] ? m : p.M - 1;   // clamped: the load itself is unconditional
                const int src = p.windowed ? window_row_to_token(mc, p.H, p.W, p.shift) : mc;
#pragma unroll
                for (int i = 0; i < V4; ++i) v[u][i] = *reinterpret_cast<const f32x4*>(p.x + (size_t)src * p.ld + (i * LPR + sub) * 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = r0 + u * RPP + tid / LPR;
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < V4; ++i) sum += (v[u][i][0] + v[u][i][1]) + (v[u][i][2] + v[u][i][3]);
                sum = allreduce<RedSum, LPR>(sum);
                const float mean = sum * (1.0f / C);
                float sq = 0.f;
#pragma unroll
                for (int i = 0; i < V4; ++i) {
                    v[u][i] -= mean;
                    sq += (v[u][i][0] * v[u][i][0] + v[u][i][1] * v[u][i][1]) + (v[u][i][2] * v[u][i][2] + v[u][i][3] * v[u][i][3]);
                }
                sq = allreduce<RedSum, LPR>(sq);
                const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
#pragma unroll
                for (int i = 0; i < V4; ++i) {
                    const int c = (i * LPR + sub) * 4;
                    f32x4 y = v[u][i] * rstd * *reinterpret_cast<const f32x4*>(p.gamma + c) + *reinterpret_cast<const f32x4*>(p.beta + c);
                    if (p.modulator) y += *reinterpret_cast<const f32x4*>(p.modulator + (size_t)((m0 + row) & 63) * C + c);  // uniform branch
                    if (!live[u]) y = f32x4{0.f, 0.f, 0.f, 0.f};
                    store4(reinterpret_cast<T*>(As + row * SA) + c, y);
                }
            }
        }
    }
    lds_barrier();
    tsx = __builtin_readcyclecounter(); ts_ln = tsx - ts0; ts0 = tsx;

    // ---------------- phase 1: barrier-free walk over 64 x 64 output units -------------------------
    // blockIdx.y splits the 64-wide column groups when M alone gives too few blocks for 256 CUs
    const int n_groups_all = (p.N + 63) >> 6;
    const int g0 = (int)((long long)n_groups_all * blockIdx.y / gridDim.y);
    const int g1 = (int)((long long)n_groups_all * (blockIdx.y + 1) / gridDim.y);
    const int n_units = MH * (g1 - g0);
    const T* Wt = reinterpret_cast<const T*>(p.Wt);
    const int Cq = p.heads * p.hd;  // == C for the QKV projection
    // Weight loads are UNCONDITIONAL (hipcc wraps a guarded load in an exec-masked branch with a
    // vmcnt(0) wait, which would serialise the prefetch ring): tiles past N are clamped to a valid
    // tile -- their products land in accumulator columns that are never stored.  For C = 16 the packed
    // k-slots 16..31 are zero padding.
    // weights are FRAGMENT-MAJOR (uf_pack_weight_fm): [n-tile][k-step][lane][8] -> one wave-level load
    // reads 1 KiB (bf16) contiguous instead of 16 half cache lines at a power-of-two stride
    const T* wrow[4];
    Frag<T> wf[RING][4];
    auto wload = [&](int ks, int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_frag(wf[slot][i], wrow[i] + ks * 512);
    };
    // first RING-1 k-steps of a unit's weights; issued for unit u+4 BEFORE the epilogue of unit u so
    // the L2 round trip hides under the epilogue instead of stalling the next unit's first MFMAs.
    auto unit_prefetch = [&](int u) {
        const int nb = (g0 + u / MH) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int tile = (nb >> 4) + i;                       // clamp: tiles past N are computed but never stored
            tile = tile < (p.N >> 4) ? tile : (p.N >> 4) - 1;
            wrow[i] = Wt + ((size_t)tile * KS * 64 + lane) * 8;
        }
#pragma unroll
        for (int s = 0; s < RING - 1; ++s)
            if (s < KS) wload(s, s);
    };
    if (wave < n_units) unit_prefetch(wave);
#pragma unroll 1
    for (int u = wave; u < n_units; u += 4) {
        tsx = __builtin_readcyclecounter(); ts_e += tsx - ts0; ts0 = tsx;   // time since the previous k-loop ended = epilogue
        const int mh = u % MH, ng = g0 + u / MH;
        const int nbase = ng * 64, mbase = mh * 64;
        // number of leading n-tiles of this unit that are NOT in the V third (wave-uniform)
        int nv = 4;
        if (EP == EP_QKV) { nv = (2 * Cq - nbase) / 16; nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv); }

        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // Software pipeline, pinned with sched_barrier (left alone, hipcc sinks the prefetch loads next
        // to their uses -> vmcnt(1) before every MFMA group -> one L2 round trip per k-step):
        //   weights for k-step ks+2 and activation fragments for ks+1 are issued BEFORE the 16 MFMAs of ks.
        const char* arow = As + (mbase + fr) * SA + fg * 8 * SZ;
        Frag<T> af[2][4];
        auto aload = [&](int ks, int slot) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (fg * 8 < C) load_frag(af[slot][j], reinterpret_cast<const T*>(arow + j * 16 * SA + ks * 32 * SZ));
                else af[slot][j].zero();
            }
        };
        aload(0, 0);
        auto kloop = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;   // 0: plain tiles, 1: all V tiles, 2: mixed unit
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + RING - 1 < KS) wload(ks + RING - 1, (ks + RING - 1) % RING);
                if (ks + 1 < KS) aload(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // V tiles are computed transposed (lane = channel, 4 consecutive tokens)
                    const bool vt_ = MODE == 1 || (MODE == 2 && i >= nv);
                    if (vt_) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) mma16(acc[i][j], af[ks & 1][j], wf[ks % RING][i]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) mma16(acc[i][j], wf[ks % RING][i], af[ks & 1][j]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (EP != EP_QKV || nv == 4) kloop(std::integral_constant<int, 0>{});
        else if (nv == 0) kloop(std::integral_constant<int, 1>{});
        else kloop(std::integral_constant<int, 2>{});
        if (u + 4 < n_units) unit_prefetch(u + 4);
        __builtin_amdgcn_sched_barrier(0);
        tsx = __builtin_readcyclecounter(); ts_k += tsx - ts0; ts0 = tsx;

        // per-tile destination info for the q/k/v^T layouts, once per unit (scalar): tile i covers
        // channels [nbase+16i, +16) = 16 consecutive d of ONE head of q, k or v.
        int t_h[4], t_d[4], t_w[4];
        if constexpr (EP == EP_QKV) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nt_ = nbase + i * 16;
                t_w[i] = nt_ / Cq;
                const int c_ = nt_ - t_w[i] * Cq;
                t_h[i] = c_ / p.hd;
                t_d[i] = c_ - t_h[i] * p.hd;
            }
        }

        // ---- epilogue for the LeFF hidden (bf16): no LDS round trip.  A lane holds 4 consecutive channels
        // (8 bytes) of one token per 16-column tile; v_permlane16_swap between lane groups fg and fg^1 of a
        // tile pair turns that into 16 contiguous bytes per lane (guide T21), so one store instruction
        // writes 64 contiguous bytes per token.
        if constexpr (EP == EP_GELU && sizeof(T) == 2) {
            f32x4 bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = nbase + i * 16 + fg * 4;
                bv[i] = *reinterpret_cast<const f32x4*>(p.bias + (n < p.N ? n : p.N - 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + mbase + j * 16 + fr;
#pragma unroll
                for (int ip = 0; ip < 4; ip += 2) {              // tile pair (ip, ip+1)
                    f32x4 va = acc[ip][j] + bv[ip], vb = acc[ip + 1][j] + bv[ip + 1];
                    gelu4<T>(va); gelu4<T>(vb);
                    const unsigned a0 = pack2<T>(va[0], va[1]), a1 = pack2<T>(va[2], va[3]);
                    const unsigned b0 = pack2<T>(vb[0], vb[1]), b1 = pack2<T>(vb[2], vb[3]);
                    // after the swap: even fg lanes hold 8 channels of tile ip, odd fg lanes 8 channels of tile ip+1
                    const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    const int n = nbase + (ip + (fg & 1)) * 16 + (fg >> 1) * 8;
                    if (m < p.M && n < p.N)
                        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n) = u32x4{s0[0], s1[0], s0[1], s1[1]};   // (nontemporal stores measured 3-10 % slower)
                }
            }
        } else
        // ---- epilogue A: [token][channel] tiles (q, k, f32 LeFF hidden): one 16-row m-tile per pass via LDS ----
        if (nv > 0) {
            f32x4 bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = nbase + i * 16 + fg * 4;
                bv[i] = *reinterpret_cast<const f32x4*>(p.bias + (n < p.N ? n : p.N - 4));  // clamped, unconditional
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < nv) {
                        f32x4 v = acc[i][j] + bv[i];
                        if constexpr (EP == EP_GELU) {
                            gelu4<T>(v);
                        } else {
                            if (nbase + i * 16 < Cq) v *= p.qscale;  // q = q * scale (model.py:497)
                        }
                        store4(reinterpret_cast<T*>(stg + fr * SS) + i * 16 + fg * 4, v);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                constexpr int CPR = 64 * SZ / 16;         // 16-byte chunks per staged row
                constexpr int ITERS = 16 * CPR / 64;
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int idx = it * 64 + lane;
                    const int r = idx / CPR, cb = idx % CPR;
                    const int m = m0 + mbase + j * 16 + r, n = nbase + cb * EPC;
                    if (m < p.M && n < p.N && (cb * EPC) / 16 < nv) {
                        const u32x4 val = *reinterpret_cast<const u32x4*>(stg + r * SS + cb * 16);
                        T* dst;
                        if constexpr (EP == EP_QKV) {
                            const int ti = (cb * EPC) >> 4;   // 16-column tile of this chunk (lane-constant per pass)
                            const int h = ti == 0 ? t_h[0] : (ti == 1 ? t_h[1] : (ti == 2 ? t_h[2] : t_h[3]));
                            const int d = (ti == 0 ? t_d[0] : (ti == 1 ? t_d[1] : (ti == 2 ? t_d[2] : t_d[3]))) + ((cb * EPC) & 15);
                            const int w = ti == 0 ? t_w[0] : (ti == 1 ? t_w[1] : (ti == 2 ? t_w[2] : t_w[3]));
                            dst = reinterpret_cast<T*>(w == 0 ? p.q : p.k) + (((size_t)(m >> 6) * p.heads + h) * 64 + (m & 63)) * p.hd + d;
                        } else {
                            dst = reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n;
                        }
                        *reinterpret_cast<u32x4*>(dst) = val;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        // ---- epilogue B: V^T tiles [channel][token]: one 16-channel n-tile per pass ---------------------
        if (EP == EP_QKV && nv < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i >= nv) {
                    const int nn = nbase + i * 16 + fr;
                    const float b = p.bias[nn < p.N ? nn : p.N - 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = {acc[i][j][0] + b, acc[i][j][1] + b, acc[i][j][2] + b, acc[i][j][3] + b};
                        store4(reinterpret_cast<T*>(stg + fr * SS) + j * 16 + fg * 4, v);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    constexpr int CPR = 64 * SZ / 16;
                    constexpr int ITERS = 16 * CPR / 64;
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int idx = it * 64 + lane;
                        const int r = idx / CPR, cb = idx % CPR;
                        const int n = nbase + i * 16 + r, m = m0 + mbase + cb * EPC;
                        if (m < p.M && n < p.N) {
                            const u32x4 val = *reinterpret_cast<const u32x4*>(stg + r * SS + cb * 16);
                            const int h = t_h[i], d = t_d[i] + r;   // tile i of the unit, channel row r
                            T* dst = reinterpret_cast<T*>(p.vt) + (((size_t)(m >> 6) * p.heads + h) * p.hd + d) * 64 + (m & 63);
                            *reinterpret_cast<u32x4*>(dst) = val;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
    }
    tsx = __builtin_readcyclecounter(); ts_e += tsx - ts0;
    census.end(p.tbuf, blockIdx.y * gridDim.x + blockIdx.x);
    if (p.tbuf && lane == 0 && (blockIdx.x & 63) == 0 && blockIdx.y == 0) {
        unsigned long long* o = p.tbuf + ((blockIdx.x >> 6) * 4 + wave) * 4;
        o[0] = ts_ln; o[1] = ts_k; o[2] = ts_e; o[3] = n_units;
    }
}

template <typename T, int C, int BM, int EP>
int launch_one(const LnGemmParams& p_in, hipStream_t st) {
    constexpr int SZ = sizeof(T);
    constexpr int smem = BM * (C * SZ + 16) + 4 * 16 * (64 * SZ + 16);
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = ln_gemm_kernel<T, C, BM, EP>;
    LnGemmParams p = p_in;
    p.tbuf = debug_get_tbuf();
    static bool lds_done[64] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, lds_done, "ln_gemm")) return rc;
    char name[96] = "";
    if (timing_enabled())
        snprintf(name, sizeof(name), "ln_gemm_%s_%s_c%d_bm%d %dx%dx%d", TypeName<T>::s, EP == EP_QKV ? "qkv" : "fc1", C, BM, p.M, p.N, C);
    const double mn = (double)p.M * p.N;
    {
        ScopedTimer tm(name, 2.0 * mn * C, (double)p.M * C * 4 + (double)p.N * C * SZ + mn * SZ, st);
        const int mb = (p.M + BM - 1) / BM, groups = (p.N + 63) / 64;
        int nsplit = 1;
        while (mb * nsplit < 512 && nsplit * 2 <= groups) nsplit *= 2;   // >= 2 blocks per CU when possible
        hipLaunchKernelGGL(kern, dim3(mb, nsplit), dim3(256), smem, st, p);
    }
    return check_launch("ln_gemm");
}

template <typename T, int C, int EP>
int launch_bm(const LnGemmParams& p, hipStream_t st) {
    // 128-row blocks when the operand tile fits 64 KiB of LDS and there are enough blocks to fill
    // 256 CUs twice over; otherwise 64-row blocks.
    constexpr bool fits128 = 128 * (C * (int)sizeof(T) + 16) <= 68 * 1024;
    if constexpr (fits128) {
        if (p.M >= 128 * 512) return launch_one<T, C, 128, EP>(p, st);
    }
    return launch_one<T, C, 64, EP>(p, st);
}

template <typename T, int EP>
int launch_c(const LnGemmParams& p, int C, hipStream_t st) {
    switch (C) {
        case 16: return launch_bm<T, 16, EP>(p, st);
        case 32: return launch_bm<T, 32, EP>(p, st);
        case 64: return launch_bm<T, 64, EP>(p, st);
        case 128: return launch_bm<T, 128, EP>(p, st);
        case 256: return launch_bm<T, 256, EP>(p, st);
        case 512: return launch_bm<T, 512, EP>(p, st);
        default:
            set_error("ln_gemm: C=%d unsupported (16,32,64,128,256,512)", C);
            return UF_ERR_UNSUPPORTED;
    }
}

int check_common(const float* x, int ld, const float* g, const float* b, const void* w, const float* bias, int B, int H, int W, int C,
                 int windowed, uf_dtype dtype) {
    UF_REQUIRE(x && g && b && w && bias, UF_ERR_NULL, "ln_gemm: null pointer");
    UF_REQUIRE(B > 0 && H > 0 && W > 0, UF_ERR_SHAPE, "ln_gemm: B=%d H=%d W=%d", B, H, W);
    UF_REQUIRE(!windowed || (H % 8 == 0 && W % 8 == 0), UF_ERR_SHAPE, "ln_gemm: windowed needs H,W multiples of 8");
    UF_REQUIRE(ld >= C && ld % 4 == 0, UF_ERR_ALIGN, "ln_gemm: ld=%d", ld);
    UF_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)bias % 16) == 0, UF_ERR_ALIGN, "ln_gemm: operands must be 16-byte aligned");
    UF_REQUIRE(dtype_ok(dtype), UF_ERR_UNSUPPORTED, "ln_gemm: dtype %d", (int)dtype);
    return UF_OK;
}

}  // namespace
}  // namespace uf

using namespace uf;

extern "C" int uf_ln_qkv_fwd(const float* x, int ld, const float* gamma, const float* beta, const float* modulator,
                             const void* Wqkv, const float* bqkv, void* q, void* k, void* vt, int B, int H, int W, int C,
                             int heads, int shift, uf_dtype dtype, void* stream) {
    int rc = check_common(x, ld, gamma, beta, Wqkv, bqkv, B, H, W, C, 1, dtype);
    if (rc) return rc;
    UF_REQUIRE(q && k && vt, UF_ERR_NULL, "uf_ln_qkv_fwd: null output");
    UF_REQUIRE(heads > 0 && C % heads == 0, UF_ERR_SHAPE, "uf_ln_qkv_fwd: C=%d heads=%d", C, heads);
    const int hd = C / heads;
    UF_REQUIRE(hd == 16 || hd == 32, UF_ERR_UNSUPPORTED, "uf_ln_qkv_fwd: head_dim %d (16 or 32 supported)", hd);
    UF_REQUIRE(shift >= 0 && shift < 8, UF_ERR_SHAPE, "uf_ln_qkv_fwd: shift=%d", shift);
    LnGemmParams p{};
    p.x = x; p.ld = ld; p.gamma = gamma; p.beta = beta; p.modulator = modulator; p.Wt = Wqkv; p.bias = bqkv;
    p.M = B * H * W; p.N = 3 * C; p.H = H; p.W = W; p.windowed = 1; p.shift = shift;
    p.q = q; p.k = k; p.vt = vt; p.heads = heads; p.hd = hd; p.qscale = (float)(1.0 / sqrt((double)hd));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UF_BF16) return launch_c<bf16, EP_QKV>(p, C, st);
    if (dtype == UF_F16) return launch_c<f16, EP_QKV>(p, C, st);
    return launch_c<float, EP_QKV>(p, C, st);
}

extern "C" int uf_ln_linear_gelu_fwd(const float* x, int ld, const float* gamma, const float* beta, const void* W1,
                                     const float* b1, void* out, int M, int N, int C, uf_dtype dtype, void* stream) {
    int rc = check_common(x, ld, gamma, beta, W1, b1, 1, 1, M > 0 ? M : 1, C, 0, dtype);
    if (rc) return rc;
    UF_REQUIRE(out && M > 0 && N > 0, UF_ERR_NULL, "uf_ln_linear_gelu_fwd: bad output / shape");
    UF_REQUIRE(N % (dtype_half(dtype) ? 8 : 4) == 0, UF_ERR_SHAPE, "uf_ln_linear_gelu_fwd: N=%d", N);
    LnGemmParams p{};
    p.x = x; p.ld = ld; p.gamma = gamma; p.beta = beta; p.modulator = nullptr; p.Wt = W1; p.bias = b1;
    p.M = M; p.N = N; p.H = 1; p.W = M; p.windowed = 0; p.shift = 0; p.out = out; p.ldo = N;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UF_BF16) return launch_c<bf16, EP_GELU>(p, C, st);
    if (dtype == UF_F16) return launch_c<f16, EP_GELU>(p, C, st);
    return launch_c<float, EP_GELU>(p, C, st);
}
