// The WHOLE LeFF half of a LeWin block in one kernel, for the HBM-bound widths (reference model.py:666-685, :987):
//
//     xo = x1 + linear2( GELU( dwconv3x3( GELU( linear1( LN2(x1) ) ) ) ) )
//
// leff2 (uf_leff2.hip) reads the 4C-wide hidden tensor h1 = GELU(linear1(LN2(x1))) that attn_block wrote: at C = 64 that round trip is
// 1 KB of the ~2 KB per token a block moves through HBM, and these stages run at HBM speed.  Here a workgroup RECOMPUTES h1 on the 10 x 10
// halo of its 8 x 8 pixel tile (1.56 x the linear1 flops, on a matrix pipe that is 6-9 % busy at these widths) and h1 never
// exists in HBM: per token the block reads x, writes x1, reads x1 (+ halo, mostly from L2) and writes x.
//
// A workgroup = 8 waves walks tiles (persistent, XCD-aware order).  Per tile:
//   P0  x1 rows of the 100 halo pixels -> LN2 -> Xn [112][C] in LDS (rows 100..111 padding; pixels outside the image flagged);
//   per 64-channel chunk of the hidden width (4C / 64 intervals):
//   P1  h1 chunk = GELU(Xn W1[chunk]^T + b1) on the MFMA (weights as the A operand: a lane gets 4 channels of one pixel), ZERO outside the
//       image (the convolution pads h1, not x), written as the halo tile [100][64] in the bank-conflict-free piece order of the MFMA
//       stencil (piece q of halo pixel (hy, hx) at slot q ^ (hx & 6): uf_leff2.hip);
//   P2  depthwise 3 x 3 + bias + GELU on the MFMA (uf_mconv.h, taps rounded to the operand type) -> operand tile [64][64];
//   P3  out[64][C] += operand tile x W2[:, chunk]^T, accumulators in registers across the chunks;
//   epilogue: + bias (x DropPath scale) + x1 rows -> xo.
// Out of place by construction: the halo rows a tile normalises belong to its neighbours, which must not have been updated yet.
// 2-byte operand types only (the f32 parity mode keeps the three-kernel path).
//
// STATUS (round 5): parity-tested (tests/test_gpu_ops.py::test_leff_halo_recompute; the reference's block fixtures under UF_LEFF3=1) and
// MEASURED SLOWER than attn_block(+fc1) + leff2 at both widths -- the stages it targets are bound by VALU issue, and the halo costs 1.75 x the linear1 /
// GELU / LN2 work -- so whole-block calls take it only with UF_LEFF3=1; uf_leff_halo_fwd reaches it directly.
#include <stdlib.h>

#include "uf_internal.h"
#include "uf_mconv.h"

namespace uf {
namespace {

struct Leff3Params {
    const float* x1; int ld1;                 // block input of this half (after the attention residual), f32 rows
    const float* gamma2; const float* beta2;
    const void* W1; const float* b1;          // T [4C][C] fragment-major, f32 [4C]
    const float* w9; const float* bdw;        // f32 [9][4C], [4C]
    const void* W2; const float* b2;          // T [C][4C] fragment-major, f32 [C]
    float* xo; int ldo;                       // output rows (may NOT alias x1)
    const float* drop;                        // per-image DropPath scale of the branch (model.py:987) or NULL
    int B, H, W, n_tiles;
};

#ifndef UF_LEFF3_WPS
#define UF_LEFF3_WPS 0
#endif

template <typename T, int C, int WPS>
__global__ __launch_bounds__(512, WPS) void leff3_kernel(const Leff3Params p) {
    static_assert(sizeof(T) == 2, "2-byte operand types");
    constexpr int SZ = 2, NT = 512, WAVES = 8;
    constexpr int HID = 4 * C, NIT = HID / 64, KS1 = C / 32;
    constexpr int HW_ = 10, HT = 100, HR = 112;             // halo tile 10 x 10, padded to 7 MFMA row tiles
    constexpr int SA = C * SZ + 16;                         // LDS row stride of Xn
    constexpr int PS = 128;                                 // halo pixel stride (64 channels)
    constexpr int SAT = 64 * SZ + 16;                       // operand tile row stride
    constexpr int WN = (C / 16) < 8 ? (C / 16) : 8, WM = 8 / WN, TNW = (C / 16) / WN, TMW = 4 / WM;   // linear2: wave grid over (out tiles, pixel tiles)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xn = smem;                                        // [HR][SA]
    char* Hs = Xn + HR * SA;                                // [2][HT][64] T, swizzled pieces
    char* At = Hs + 2 * HT * PS;                            // [2][64][SAT]
    float* Tap = reinterpret_cast<float*>(At + 2 * 64 * SAT);   // [10][HID]: 9 tap rows + conv bias
    float* B1s = Tap + 10 * HID;                            // [HID]
    float* Gb = B1s + HID;                                  // [2][C] norm2 weight, bias

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = p.W / 8, tiles_y = p.H / 8;
    const T* W1 = reinterpret_cast<const T*>(p.W1);
    const T* W2 = reinterpret_cast<const T*>(p.W2);

    // once per workgroup: tap table, conv bias and linear1 bias -> LDS
    for (int i = tid; i < 9 * HID; i += NT) Tap[i] = p.w9[i];
    for (int i = tid; i < HID; i += NT) { Tap[9 * HID + i] = p.bdw[i]; B1s[i] = p.b1[i]; }
    for (int i = tid; i < C; i += NT) { Gb[i] = p.gamma2[i]; Gb[C + i] = p.beta2[i]; }
    lds_barrier();

    // MFMA stencil constants of this wave (uf_leff2.hip): 16-channel group gq of the chunk, pixel tiles pt0, pt0 + 1
    const int gq = wave & 3, pt0 = (wave >> 2) * 2;
    const unsigned hshift = (fr & 1) * 16;
    unsigned msk[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) msk[d] = ((fg & 1) == (fr >> 3) && d == ((fr & 7) >> 1)) ? 0xffffffffu : 0u;
    int boff[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        int tap = 2 * ks + (fg >> 1);
        tap = tap < 9 ? tap : 8;
        const int hy = (fr >> 3) + tap / 3, hx = (fr & 7) + tap % 3;
        boff[ks] = ((hy * HW_ + hx) * 8 + ((gq * 2 + (fg & 1)) ^ (hx & 6))) * 16;
    }
    // P1 roles: hidden tile ct of the chunk (16 channels), halo row tiles rt0 .. rt0 + NR1 - 1
    const int ct = wave & 3, rt0 = (wave >> 2) * 4;
    // this lane's halo pixel in each of its row tiles: LDS byte offset of its 8-byte store (or -1: padding row)
    int hoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int hp = (rt0 + j) * 16 + fr;
        const int hx = hp % HW_;
        hoff[j] = (rt0 + j < 7 && hp < HT) ? hp * PS + (((2 * ct + (fg >> 1)) ^ (hx & 6)) * 16) + (fg & 1) * 8 : -1;
    }
    // P3 roles
    const int wn = wave % WN, wm = wave / WN;
    f32x4 acc[TNW][TMW];
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TMW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Frag<T> w1f[KS1];
    auto w1_load = [&](int it) {                            // fragment-major W1: tile (it * 4 + ct), k-step ks -> 1 KiB per wave load
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) load_frag(w1f[ks], W1 + (((size_t)(it * 4 + ct) * KS1 + ks) * 64 + lane) * 8);
    };
    w1_load(0);

    // ---- the schedule: ONE barrier per slot, three independent pieces of work per wave between two barriers -------------------------------
    // Stream A walks the tiles of this workgroup: slot 0 of a tile = P0 (LN2 of its halo rows -> Xn), slots 1 .. NIT = P1 of chunk 0 .. NIT-1
    // (h1 chunk -> halo tile Hs[chunk & 1]).  Stream B runs P2 (stencil: Hs[chunk & 1] -> operand tile At[chunk & 1]) of the chunk A produced one
    // slot earlier, stream C runs P3 (linear2 partial sums from At[chunk & 1]; after the last chunk of a tile: the epilogue) of the chunk B
    // produced one slot earlier.  The three pieces of a slot touch different buffers, so a wave's LDS round trips, MFMAs and GELU arithmetic of
    // one piece overlap those of the others, and the pipeline runs across tile borders (the x1 rows of the next tile are requested at the start
    // of its P0 slot and normalised at its end, behind the stencil and linear2 work of the previous tile).
    const int G = (int)gridDim.x;
    const int nk = (p.n_tiles - (int)blockIdx.x + G - 1) / G;           // tiles of this workgroup: blockIdx.x + k G
    constexpr int LPR = C / 4;                                            // P0: lanes per row, one f32x4 each
    constexpr int RPP = NT / LPR;                                         // rows per pass
    constexpr int NPASS = (HR + RPP - 1) / RPP;
    const int sub = tid % LPR;
    int cb = 0, cy0 = 0, cx0 = 0;                                         // tile of stream A
    int eb = 0, ey0 = 0, ex0 = 0;                                         // tile whose epilogue is pending
    bool inimg[4] = {false, false, false, false};
    int ja = 0, ka = 0;                                                   // stream A: slot within the tile, tile index
    int itB = -1, itC = -1;                                               // chunk of P2 / P3 in this slot (-1: none)
    const int NSLOT = nk * (NIT + 1) + 2;
#pragma unroll 1
    for (int m = 0; m < NSLOT; ++m) {
        const bool a_p0 = ka < nk && ja == 0;
        const int itA = (ka < nk && ja > 0) ? ja - 1 : -1;
        const bool epi = itC == NIT - 1;
        // ---- requests first: linear2 weight fragments, the epilogue's own x1 rows, the next tile's halo rows ----
        Frag<T> w2f[2][TNW];
        if (itC >= 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TNW; ++i)
                    load_frag(w2f[ks][i], W2 + (((size_t)(wn * TNW + i) * (HID / 32) + itC * 2 + ks) * 64 + lane) * 8);
        }
        f32x4 xres[TNW][TMW];
        if (epi) {                                                        // L2: this workgroup's P0 read them
            const float* xe = p.x1 + (size_t)eb * p.H * p.W * p.ld1;
#pragma unroll
            for (int i = 0; i < TNW; ++i)
#pragma unroll
                for (int j = 0; j < TMW; ++j) {
                    const int pm = (wm * TMW + j) * 16 + fr;
                    xres[i][j] = *reinterpret_cast<const f32x4*>(xe + ((size_t)(ey0 + (pm >> 3)) * p.W + ex0 + (pm & 7)) * p.ld1 + (wn * TNW + i) * 16 + fg * 4);
                }
        }
        auto stream_b = [&]() {
            // ---------------- stream B, P2: depthwise 3 x 3 + GELU of chunk itB: Hs[itB & 1] -> At[itB & 1] ------------------------
            if (itB >= 0)
                mconv_job<T, 1, 2, HID, SAT, 2 * HW_ * PS>(Hs + (itB & 1) * (HT * PS), Tap + itB * 64, At + (itB & 1) * (64 * SAT), gq, pt0, boff, msk, hshift, fr, fg);
        };
        auto stream_c = [&]() {
            // ---------------- stream C, P3: linear2 partial sums of chunk itC from At[itC & 1] (+ epilogue) ------------------------
            if (itC >= 0) {
                const char* Ar = At + (itC & 1) * (64 * SAT);
    #pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    Frag<T> af[TMW];
    #pragma unroll
                    for (int j = 0; j < TMW; ++j) load_frag(af[j], reinterpret_cast<const T*>(Ar + ((wm * TMW + j) * 16 + fr) * SAT + (ks * 32 + fg * 8) * SZ));
    #pragma unroll
                    for (int i = 0; i < TNW; ++i)
    #pragma unroll
                        for (int j = 0; j < TMW; ++j) mma16(acc[i][j], w2f[ks][i], af[j]);
                }
                if (epi) {
                    const float dscale = p.drop ? p.drop[eb] : 1.0f;
                    float* ob = p.xo + (size_t)eb * p.H * p.W * p.ldo;
    #pragma unroll
                    for (int i = 0; i < TNW; ++i) {
                        const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.b2 + (wn * TNW + i) * 16 + fg * 4);
    #pragma unroll
                        for (int j = 0; j < TMW; ++j) {
                            const int pm = (wm * TMW + j) * 16 + fr;
                            *reinterpret_cast<f32x4*>(ob + ((size_t)(ey0 + (pm >> 3)) * p.W + ex0 + (pm & 7)) * p.ldo + (wn * TNW + i) * 16 + fg * 4) =
                                xres[i][j] + (acc[i][j] + b2) * dscale;
                            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
                }
            }
        };
        if (a_p0) {
            f32x4 vv[NPASS];
            const int t = xcd_tile((int)blockIdx.x + ka * G, p.n_tiles);
            cb = t / (tiles_x * tiles_y);
            const int tr = t - cb * (tiles_x * tiles_y);
            cy0 = (tr / tiles_x) * 8; cx0 = (tr % tiles_x) * 8;
            const float* xb = p.x1 + (size_t)cb * p.H * p.W * p.ld1;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int row = ps * RPP + tid / LPR;
                const int hy = row / HW_, hx = row - hy * HW_;
                const int iy = cy0 + hy - 1, ix = cx0 + hx - 1;
                const bool ok = row < HT && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                vv[ps] = ok ? *reinterpret_cast<const f32x4*>(xb + ((size_t)iy * p.W + ix) * p.ld1 + sub * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            stream_b();
            stream_c();
        // ---------------- stream A, P0: LN2 of the halo rows requested above -> Xn ------------------------------------------------
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int row = ps * RPP + tid / LPR;
                float sm = (vv[ps][0] + vv[ps][1]) + (vv[ps][2] + vv[ps][3]);
                sm = allreduce<RedSum, LPR>(sm);
                const float mean = sm * (1.0f / C);
                const f32x4 d = vv[ps] - mean;
                float sq = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                sq = allreduce<RedSum, LPR>(sq);
                const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
                const f32x4 gm = *reinterpret_cast<const f32x4*>(Gb + sub * 4), bt = *reinterpret_cast<const f32x4*>(Gb + C + sub * 4);
                if (row < HR) store4(reinterpret_cast<T*>(Xn + row * SA) + sub * 4, d * rstd * gm + bt);
            }
            // which of this lane's halo pixels of the P1 role lie inside the image
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hp = (rt0 + j) * 16 + fr;
                const int hy = hp / HW_, hx = hp - hy * HW_;
                const int iy = cy0 + hy - 1, ix = cx0 + hx - 1;
                inimg[j] = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            }
        // ---------------- stream A, P1: h1 chunk itA on the halo -> Hs[itA & 1] --------------------------------------------------
        } else if (itA >= 0) {
            char* Hw = Hs + (itA & 1) * (HT * PS);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(B1s + itA * 64 + ct * 16 + fg * 4);
#pragma unroll
            for (int jh = 0; jh < 4; jh += 2) {                           // two row tiles at a time (registers)
                if (rt0 + jh < 7) {
                    f32x4 a1[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) a1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        Frag<T> af[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j)                        // (row tile 7 of the second half does not exist: its rows are the padding of the last)
                            load_frag(af[j], reinterpret_cast<const T*>(Xn + ((rt0 + jh + j < 7 ? rt0 + jh + j : 6) * 16 + fr) * SA + (ks * 32 + fg * 8) * SZ));
#pragma unroll
                        for (int j = 0; j < 2; ++j) mma16(a1[j], w1f[ks], af[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (hoff[jh + j] >= 0) {
                            f32x4 h = a1[j] + bv;
                            gelu4<T>(h);
                            u32x2 o = {pack2<T>(h[0], h[1]), pack2<T>(h[2], h[3])};
                            if (!inimg[jh + j]) o = u32x2{0u, 0u};        // the convolution pads h1 with zeros (model.py:659)
                            *reinterpret_cast<u32x2*>(Hw + hoff[jh + j]) = o;
                        }
                    }
                }
            }
            // the next chunk's (or the next tile's first) weight fragments: in flight until the next slot
            w1_load(itA + 1 < NIT ? itA + 1 : 0);
            if (itA == NIT - 1) { eb = cb; ey0 = cy0; ex0 = cx0; }       // this tile's epilogue runs two slots from now
            stream_b();
            stream_c();
        } else {
            stream_b();
            stream_c();
        }
        lds_barrier();
        itC = itB; itB = itA;
        if (ka < nk) { if (++ja > NIT) { ja = 0; ++ka; } }
    }
}

template <typename T, int C>
int launch_c(const Leff3Params& p, hipStream_t st) {
    constexpr int HID = 4 * C;
    constexpr int smem = 112 * (C * 2 + 16) + 2 * 100 * 128 + 2 * 64 * (64 * 2 + 16) + 10 * HID * 4 + HID * 4 + 2 * C * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    // two 8-wave workgroups per CU: the kernel needs 96 (C = 32) / 121 (C = 64) registers, i.e. 4 waves per SIMD; a third workgroup (<= 80 registers)
    // spills 15 / 48 of them (UF_LEFF3_WPS=6 builds that variant for A/B runs)
    constexpr int per_cu_lds = (160 * 1024) / smem;
    constexpr int WPS = UF_LEFF3_WPS > 0 ? UF_LEFF3_WPS : 4;                // waves per SIMD the register allocation must allow
    constexpr int per_cu = (WPS / 2) < per_cu_lds ? (WPS / 2) : per_cu_lds;
    auto kern = leff3_kernel<T, C, WPS>;
    static bool lds_done[64] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, lds_done, "leff3")) return rc;
    const long long M = (long long)p.B * p.H * p.W;
    char name[96] = "";
    if (timing_enabled()) snprintf(name, sizeof(name), "leff3_%s_c%d %lldx%dx%d", TypeName<T>::s, C, M, C, HID);
    {
        // algorithmic work: linear1 + linear2 + the stencil once per token (the halo recomputation is overhead, not work); bytes: x1 in, x out
        ScopedTimer tm(name, 2.0 * M * C * HID * 2 + 18.0 * M * HID, (double)M * C * 8, st);
        const int resident = 256 * per_cu;
        int grid = p.n_tiles;
        if (p.n_tiles > resident) {
            const int rounds = (p.n_tiles + resident - 1) / resident;
            grid = ((p.n_tiles + rounds - 1) / rounds + 7) / 8 * 8;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, p);
    }
    return check_launch("leff3");
}

}  // namespace

bool leff3_covers(uf_dtype dtype, int C) { return dtype_half(dtype) && (C == 32 || C == 64); }

// Whether whole-block calls take this kernel.  OFF by default: measured on MI355X (profiles/r05_run1_ab.txt, r05_run2_ab.txt) the pair
// attn_block (without its fc1 phase) + leff3 is SLOWER than attn_block(+fc1) + leff2 at every width it covers -- dec3 (1 M tokens, C = 64)
// 196 + 538 us against 366 + 269, enc1 52 + 139 against 90 + 71, enc0 (C = 32) 97 + 263 against 175 + 133 -- although it moves half the
// bytes: these stages are bound by VALU issue (GELU, LayerNorm, conversions), not by HBM, and the halo costs 1.75 x the linear1 / GELU / LN2
// work.  UF_LEFF3=1 selects it (A/B runs); uf_leff_halo_fwd reaches it directly.
bool leff3_supported(uf_dtype dtype, int C) {
    static const char* e = getenv("UF_LEFF3");
    return e && e[0] == '1' && leff3_covers(dtype, C);
}

int launch_leff3(const uf_block_params* bp, const float* x1, int ld1, float* xo, int ldo, int B, int H, int W, int C, uf_dtype dtype,
                 const float* drop, hipStream_t st) {
    UF_REQUIRE(bp && x1 && xo, UF_ERR_NULL, "leff3: null pointer");
    UF_REQUIRE(x1 != xo, UF_ERR_SHAPE, "leff3: the output may not alias the input (a tile reads its neighbours' rows)");
    UF_REQUIRE(B > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0, UF_ERR_SHAPE, "leff3: B=%d H=%d W=%d (multiples of 8)", B, H, W);
    UF_REQUIRE(ld1 >= C && ld1 % 4 == 0 && ldo >= C && ldo % 4 == 0, UF_ERR_ALIGN, "leff3: ld1=%d ldo=%d", ld1, ldo);
    UF_REQUIRE(((uintptr_t)x1 % 16) == 0 && ((uintptr_t)xo % 16) == 0, UF_ERR_ALIGN, "leff3: rows must be 16-byte aligned");
    UF_REQUIRE((long long)B * H * W < 0x7fffffffLL / 4, UF_ERR_SHAPE, "leff3: too many tokens");
    Leff3Params p{};
    p.x1 = x1; p.ld1 = ld1; p.gamma2 = bp->norm2_w; p.beta2 = bp->norm2_b; p.W1 = bp->w1_fm; p.b1 = bp->b1; p.w9 = bp->wdw9; p.bdw = bp->bdw;
    p.W2 = bp->w2_fm; p.b2 = bp->b2; p.xo = xo; p.ldo = ldo; p.drop = drop; p.B = B; p.H = H; p.W = W; p.n_tiles = B * (H / 8) * (W / 8);
#define UF_L3(TT) switch (C) { case 32: return launch_c<TT, 32>(p, st); case 64: return launch_c<TT, 64>(p, st); }
    if (dtype == UF_BF16) { UF_L3(bf16) } else if (dtype == UF_F16) { UF_L3(f16) }
#undef UF_L3
    set_error("leff3: unsupported C=%d for dtype %d", C, (int)dtype);
    return UF_ERR_UNSUPPORTED;
}

}  // namespace uf
