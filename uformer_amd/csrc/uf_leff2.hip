// Second half of LeFF in ONE kernel (reference model.py:674-683, :987):
//
//     x += linear2( GELU( dwconv3x3( h1 ) ) )        h1 = GELU(linear1(LN2(x)))  [B][H][W][4C]
//
// The depthwise 3x3 runs over the whole H x W map (it crosses window borders), so a workgroup owns a
// SPATIAL tile of TH x TW pixels and walks over the 4C hidden channels in chunks of 64:
//   1. the (TH+2) x (TW+2) halo tile of the chunk is staged in LDS (zero outside the image =
//      the convolution's zero padding); loads for chunk c+1 are issued before the stencil of chunk c;
//   2. stencil + bias + GELU on the VALU: a thread owns 8 channels x a 2-row column strip, taps from an
//      LDS table, and writes the MFMA operand tile [pixels][64] to LDS;
//   3. MFMA: out[pixels][C] += tile x W2[:, chunk]^T, W2 fragments streamed L2 -> registers (issued
//      before the stencil so the round trip hides under it), accumulators stay in registers.
// The conv output (the largest tensor of the block, 4C per token) never goes to HBM.
#include <stdlib.h>

#include "uf_internal.h"

namespace uf {
namespace {

struct Leff2Params {
    const void* h1;                 // T [B][H][W][4C]
    const float* w9; const float* bdw;  // f32 [9][4C], [4C]
    const void* W2; const float* b2;    // T [C][4C], f32 [C]
    float* x; int ld;               // residual stream rows, in place
    int B, H, W;
    unsigned long long* tbuf;   // optional per-role cycle totals of sampled blocks (uf_debug_set_tbuf)
};

constexpr int KC = 64;  // hidden channels per chunk

template <typename T> __device__ __forceinline__ void cvt8(const char* p, float* f);
template <> __device__ __forceinline__ void cvt8<bf16>(const char* p, float* f) {
    const u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(r[i] << 16); f[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void cvt8<float>(const char* p, float* f) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 16);
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}
__device__ __forceinline__ void put8(bf16* p, const float* f) {
    *reinterpret_cast<u32x4*>(p) = u32x4{pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
}
__device__ __forceinline__ void put8(float* p, const float* f) {
    *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
}

// Wave-specialised version.  NP + NC waves; the hardware places waves w, w+4, w+8 of a workgroup on the same
// SIMD, so every SIMD hosts one PRODUCER wave (0..NP-1: depthwise stencil + GELU, pure VALU/LDS) and NC/4 CONSUMER
// waves (halo + tap + W2 traffic, MFMAs, epilogue; NC = 4, or 8 where the accumulators of C >= 256 output channels
// would otherwise make the consumers the long pole).  The VALU and matrix pipes of a SIMD run concurrently for
// different waves, so the stencil of chunk i overlaps the MFMAs of chunk i-1; halo tile, tap table and operand tile
// are double-buffered in LDS, one barrier per chunk.
template <typename T, int C, int NP, int NC>
// C <= 64: cap registers at 85 (6 waves per SIMD) so THREE workgroups fit a CU (LDS allows it: 3 x 52 KiB); the few chunks
// per tile at small C make pipeline fill/drain a third of a workgroup's life, and a third resident workgroup hides it
// (A/B on one box: 369 -> 308 us at C=64, 1 M tokens).  C = 128 spills under that cap and stays at two.
__global__ __launch_bounds__((NP + NC) * 64, (C <= 64 ? 6 : (NP + NC) / 4)) void leff2_kernel(const Leff2Params p) {
    constexpr int SR = 8 / NP;                    // rows of the column strip one producer thread convolves (NP = 4 or 8 producer waves)
    constexpr int SZ = sizeof(T);
    constexpr int TH = 8, TW = 8, BM = 64;
    constexpr int HID = 4 * C;
    constexpr int NCH = HID / KC;                 // 64-channel chunks
    constexpr int HW_ = TW + 2, HT = (TH + 2) * HW_;  // 10 x 10 halo tile
    constexpr int SH = KC * SZ + 16;              // LDS row stride, halo tile [HT][KC]
    constexpr int SAT = KC * SZ + 16;             // LDS row stride, operand tile [BM][KC]
    constexpr int CPP = KC * SZ / 16;             // 16-byte pieces per pixel per chunk
    constexpr int NCT = NC * 64;                  // consumer threads
    constexpr int NLD = (HT * CPP + NCT - 1) / NCT;   // staged 16-byte loads per consumer thread
    constexpr int WN = (C / 16) < NC ? (C / 16) : NC, WM = NC / WN;   // consumer wave grid (pixels x out channels)
    static_assert(WM <= 4 && NC * 64 >= 160, "consumer grid");
    constexpr int TMW = 4 / WM, TNW = (C / 16) / WN;               // 16x16 tiles per consumer wave
    constexpr int HS_BYTES = HT * SH, AT_BYTES = BM * SAT, WL_BYTES = 10 * KC * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Hs0 = smem;                             // halo tiles [2]
    char* At0 = smem + 2 * HS_BYTES;              // operand tiles [2]
    char* Wl0 = smem + 2 * HS_BYTES + 2 * AT_BYTES;   // taps [9][64] + bias [64], [2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < NP;
    const int ct = tid - NP * 64;                 // consumer thread index (0..255) when !producer
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = p.W / TW, tiles_y = p.H / TH;
    const int bt = blockIdx.x;
    const int b = bt / (tiles_x * tiles_y), tr = bt - b * (tiles_x * tiles_y);
    const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) * TW;
    const T* h1 = reinterpret_cast<const T*>(p.h1) + (size_t)b * p.H * p.W * HID;
    const T* W2 = reinterpret_cast<const T*>(p.W2);

    Census census; census.begin();
    if (producer) {
        // ------------------------------ producers: stencil ------------------------------------------
        // a thread owns 8 channels x SR rows of one tile column; the 3 taps of a column come from the LDS tap
        // table (same address for all pixels of a channel group: broadcast).  Measured alternatives that were
        // slower: 8 producer waves (LDS-read bound, no gain) and wave-uniform taps via scalar loads (s_load
        // shares lgkmcnt with the LDS reads and 72 taps exceed the SGPR budget: 2x slower).
        const int cvec = tid & 7, sx = (tid >> 3) & 7, sy0 = (tid >> 6) * SR;
        lds_barrier();                                                          // B0: halo(0), taps(0) staged
        unsigned long long tw = 0, tbar = 0, t0 = __builtin_readcyclecounter(), t1;
#pragma unroll 1
        for (int i = 0; i <= NCH; ++i) {
            if (i < NCH) {
                const char* Hs = Hs0 + (i & 1) * HS_BYTES;
                const float* Wl = reinterpret_cast<const float*>(Wl0 + (i & 1) * WL_BYTES);
                char* At = At0 + (i & 1) * AT_BYTES;
                float o[SR][8];
#pragma unroll
                for (int r = 0; r < SR; ++r)
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[r][k] = Wl[9 * KC + cvec * 8 + k];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float wk[3][8];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int k = 0; k < 8; ++k) wk[ky][k] = Wl[(ky * 3 + kx) * KC + cvec * 8 + k];
#pragma unroll
                    for (int r = -1; r <= SR; ++r) {      // halo row (sy0 + r + 1) feeds output rows r+1-ky
                        float f[8];
                        cvt8<T>(Hs + ((sy0 + r + 1) * HW_ + sx + kx) * SH + cvec * 8 * SZ, f);
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const int orow = r + 1 - ky;
                            if (orow < 0 || orow >= SR) continue;
#pragma unroll
                            for (int k = 0; k < 8; ++k) o[orow][k] = fmaf(f[k], wk[ky][k], o[orow][k]);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < SR; ++r) {
                    gelu_n<T, 8>(o[r]);
                    put8(reinterpret_cast<T*>(At + ((sy0 + r) * TW + sx) * SAT) + cvec * 8, o[r]);
                }
            }
            t1 = __builtin_readcyclecounter(); tw += t1 - t0; t0 = t1;
            lds_barrier();
            t1 = __builtin_readcyclecounter(); tbar += t1 - t0; t0 = t1;
        }
        if (p.tbuf && lane == 0 && (bt & 63) == 0) { p.tbuf[((bt >> 6) * 16 + wave) * 4] = tw; p.tbuf[((bt >> 6) * 16 + wave) * 4 + 1] = tbar; p.tbuf[((bt >> 6) * 16 + wave) * 4 + 2] = producer; }
        return;
    }

    // ---------------------------------- consumers ----------------------------------------------------
    const int cw = wave - NP;
    const int wm = cw / WN, wn = cw % WN;
    // staging bookkeeping: which halo pixel / 16-byte piece each of this thread's loads covers
    u32x4 stage[NLD];
    int s_off[NLD];
    bool s_ok[NLD];
    int s_lds[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = ct + NCT * i;
        const int hp = idx / CPP, piece = idx - hp * CPP;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        s_ok[i] = idx < HT * CPP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;   // else zero = conv padding
        const int cy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy), cx = ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix);
        s_off[i] = (cy * p.W + cx) * HID + piece * (16 / SZ);
        s_lds[i] = idx < HT * CPP ? hp * SH + piece * 16 : -1;
    }
    f32x4 wstage;
    const int wl_row = ct >> 4, wl_c4 = (ct & 15) * 4;    // taps: 10 rows x 64 floats = 160 float4 (ct < 160)
    auto stage_issue = [&](int ch) {                      // unconditional loads from clamped addresses
#pragma unroll
        for (int i = 0; i < NLD; ++i) stage[i] = *reinterpret_cast<const u32x4*>(h1 + s_off[i] + ch * KC);
        const int row = wl_row < 10 ? wl_row : 9;
        const float* src = row < 9 ? p.w9 + (size_t)row * HID : p.bdw;
        wstage = *reinterpret_cast<const f32x4*>(src + ch * KC + wl_c4);
    };
    auto stage_store = [&](int ch) {
        char* Hs = Hs0 + (ch & 1) * HS_BYTES;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (s_lds[i] >= 0) *reinterpret_cast<u32x4*>(Hs + s_lds[i]) = s_ok[i] ? stage[i] : u32x4{0, 0, 0, 0};
        if (ct < 160) *reinterpret_cast<f32x4*>(Wl0 + (ch & 1) * WL_BYTES + (wl_row * KC + wl_c4) * 4) = wstage;
    };
    Frag<T> wf[2][TNW];
    auto w2_issue = [&](int ch) {                          // fragment-major W2: 1 KiB contiguous per wave load
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TNW; ++i)
                load_frag(wf[ks][i], W2 + (((size_t)(wn * TNW + i) * (HID / 32) + ch * 2 + ks) * 64 + lane) * 8);
    };

    f32x4 acc[TNW][TMW];
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TMW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Schedule of iteration j (producers run stencil(j) meanwhile):
    //   (a) store halo/taps of chunk j+1 (loaded during iteration j-1) into buffer (j+1)&1 -- last read by
    //       the producers in iteration j-1;  (b) issue the loads of chunk j+2 (a whole iteration to land:
    //       raw barriers do not drain them);  (c) MFMAs of chunk j-1, then issue chunk j's W2 fragments.
    stage_issue(0);
    stage_store(0);
    if (NCH > 1) stage_issue(1);
    w2_issue(0);
    lds_barrier();                                         // B0
    unsigned long long tw = 0, tbar = 0, t0 = __builtin_readcyclecounter(), t1;
#pragma unroll 1
    for (int j = 0; j <= NCH; ++j) {
        if (j + 1 < NCH) stage_store(j + 1);
        if (j + 2 < NCH) stage_issue(j + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (j >= 1) {
            const char* At = At0 + ((j - 1) & 1) * AT_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<T> af[TMW];
#pragma unroll
                for (int jj = 0; jj < TMW; ++jj)
                    load_frag(af[jj], reinterpret_cast<const T*>(At + ((wm * TMW + jj) * 16 + fr) * SAT + (ks * 32 + fg * 8) * SZ));
#pragma unroll
                for (int ii = 0; ii < TNW; ++ii)
#pragma unroll
                    for (int jj = 0; jj < TMW; ++jj) mma16(acc[ii][jj], wf[ks][ii], af[jj]);
            }
            if (j < NCH) w2_issue(j);                      // fragments for the next iteration's MFMAs
        }
        t1 = __builtin_readcyclecounter(); tw += t1 - t0; t0 = t1;
        lds_barrier();
        t1 = __builtin_readcyclecounter(); tbar += t1 - t0; t0 = t1;
    }
    if (p.tbuf && lane == 0 && (bt & 63) == 0) { p.tbuf[((bt >> 6) * 16 + wave) * 4] = tw; p.tbuf[((bt >> 6) * 16 + wave) * 4 + 1] = tbar; p.tbuf[((bt >> 6) * 16 + wave) * 4 + 2] = producer; }

    // ---- epilogue: + bias + residual, in place on the f32 stream (model.py:987) ----
#pragma unroll
    for (int i = 0; i < TNW; ++i) {
        const int n = (wn * TNW + i) * 16 + fg * 4;
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.b2 + n);
#pragma unroll
        for (int j = 0; j < TMW; ++j) {
            const int pm = (wm * TMW + j) * 16 + fr;
            const int ty = pm >> 3, tx = pm & 7;
            float* xp = p.x + ((size_t)(b * p.H + y0 + ty) * p.W + x0 + tx) * p.ld + n;
            *reinterpret_cast<f32x4*>(xp) = *reinterpret_cast<const f32x4*>(xp) + (acc[i][j] + b2);
        }
    }
    if (ct == 0) {   // census entry written by the first consumer thread (thread 0 is a producer and has returned)
        Census c2 = census;
        if (p.tbuf) { unsigned long long* o = p.tbuf + 65536 + (size_t)bt * 8; o[0] = c2.t0; o[1] = __builtin_readcyclecounter(); o[2] = c2.r0; o[3] = wall_clock64();
            o[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); }
    }
}

template <typename T, int C>
int launch_c(const Leff2Params& p, hipStream_t st) {
    constexpr int SZ = sizeof(T);
    // one producer (stencil) wave + one consumer (MFMA) wave per SIMD
    // one producer (stencil) wave per SIMD; C >= 256 cannot fit two workgroups per CU anyway (accumulators), so it gets
    // 8 consumer waves (half the accumulators, W2 fragments and MFMAs per wave): the consumers stop being the long pole
    constexpr int NP = 4, NC = (SZ == 2 && C >= 256) ? 8 : 4;
    constexpr int smem = 2 * 100 * (KC * SZ + 16) + 2 * 64 * (KC * SZ + 16) + 2 * 10 * KC * 4;
    auto kern = leff2_kernel<T, C, NP, NC>;
    static bool lds_done[64] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, lds_done, "leff2")) return rc;
    const long long M = (long long)p.B * p.H * p.W;
    char name[96] = "";
    if (timing_enabled()) snprintf(name, sizeof(name), "leff2_%s_c%d_np%d_nc%d %lldx%dx%d", SZ == 2 ? "bf16" : "f32", C, NP, NC, M, C, 4 * C);
    {
        ScopedTimer tm(name, 2.0 * M * C * 4 * C + 18.0 * M * 4 * C, (double)M * 4 * C * SZ + (double)M * C * 8, st);
        hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * (p.H / 8) * (p.W / 8))), dim3((NP + NC) * 64), smem, st, p);
    }
    return check_launch("leff2");
}

template <typename T>
int launch_t(const Leff2Params& p, int C, hipStream_t st) {
    switch (C) {
        case 16: return launch_c<T, 16>(p, st);
        case 32: return launch_c<T, 32>(p, st);
        case 64: return launch_c<T, 64>(p, st);
        case 128: return launch_c<T, 128>(p, st);
        case 256: return launch_c<T, 256>(p, st);
        case 512: return launch_c<T, 512>(p, st);
        default:
            set_error("leff2: C=%d unsupported (16,32,64,128,256,512)", C);
            return UF_ERR_UNSUPPORTED;
    }
}

}  // namespace
}  // namespace uf

namespace uf { unsigned long long* debug_get_tbuf(); }
using namespace uf;

extern "C" int uf_dwconv_linear2_fwd(const void* h1, const float* w9, const float* bdw, const void* W2, const float* b2,
                                     float* x, int ld, int B, int H, int W, int C, uf_dtype dtype, void* stream) {
    UF_REQUIRE(h1 && w9 && bdw && W2 && b2 && x, UF_ERR_NULL, "uf_dwconv_linear2_fwd: null pointer");
    UF_REQUIRE(B > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0, UF_ERR_SHAPE, "uf_dwconv_linear2_fwd: B=%d H=%d W=%d (multiples of 8)", B, H, W);
    UF_REQUIRE(ld >= C && ld % 4 == 0, UF_ERR_ALIGN, "uf_dwconv_linear2_fwd: ld=%d", ld);
    UF_REQUIRE(((uintptr_t)h1 % 16) == 0 && ((uintptr_t)W2 % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w9 % 16) == 0 &&
                   ((uintptr_t)bdw % 16) == 0 && ((uintptr_t)b2 % 16) == 0,
               UF_ERR_ALIGN, "uf_dwconv_linear2_fwd: operands must be 16-byte aligned");
    UF_REQUIRE((long long)B * H * W * 4LL * C < 0x7fffffffLL, UF_ERR_SHAPE, "uf_dwconv_linear2_fwd: tensor too large for 32-bit indexing");
    Leff2Params p{};
    p.tbuf = uf::debug_get_tbuf();
    p.h1 = h1; p.w9 = w9; p.bdw = bdw; p.W2 = W2; p.b2 = b2; p.x = x; p.ld = ld; p.B = B; p.H = H; p.W = W;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UF_BF16) return launch_t<bf16>(p, C, st);
    if (dtype == UF_F32) return launch_t<float>(p, C, st);
    set_error("uf_dwconv_linear2_fwd: dtype %d", (int)dtype);
    return UF_ERR_UNSUPPORTED;
}
