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
#include "uf_mconv.h"

namespace uf {
namespace {

struct Leff2Params {
    const void* h1;                 // T [B][H][W][4C]
    const float* w9; const float* bdw;  // f32 [9][4C], [4C]
    const void* W2; const float* b2;    // T [C][4C], f32 [C]
    float* x; int ld;               // residual stream rows, in place
    const float* drop;              // training: per-image DropPath scale of the branch (model.py:987) or NULL
    int B, H, W;
    int n_tiles;                // B * (H / 8) * (W / 8): a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (persistent launch: gridDim.x < n_tiles)
    unsigned long long* tbuf;   // optional per-role cycle totals of sampled blocks (uf_debug_set_tbuf)
};

constexpr int KCW = 64;  // hidden channels one producer group (4 waves) convolves per interval

// UF_MCONV: how the producers of the 2-byte operand types evaluate the depthwise 3x3 (round 4).
//   0  VALU stencil (rounds 1-3): 9 FMAs per output + one bf16->f32 unpack per (output, column tap) -- 298 VALU instructions per producer
//      wave and interval; the PMC passes of round 3 show the VALU pipe of a SIMD (one wave-instruction per 4 cycles) as the busiest
//      resource of this kernel (47-55 % at every width), the MFMA pipe at 5-19 %.
//   1  MFMA stencil, taps rounded to the operand type: a 16-channel group of a 16-pixel tile is D[c][p] = sum_k A[c][k] B[k][p] with
//      k = (tap, c'), A[c][(tap, c')] = w[tap][c] delta(c, c') (a block-diagonal weight operand built in registers: one non-zero 16-bit
//      slot per lane) and B[(tap, c')][p] = halo[p + tap][c'] = a 16-byte ds_read_b128 of the DMA-staged halo tile AS IT LIES in LDS --
//      no unpack, no FMA on the VALU; 15/16 of the MFMA's multiplies are zeros, on a pipe that is idle.  K = 9 taps x 16 = 5 k-steps
//      of 32 (two taps per step).  The accumulator starts at the conv bias, GELU runs on the accumulator registers, and the result is
//      written to the operand tile of linear2 exactly as before.
//   2  the same with every tap split into hi + lo parts of the operand type (9 k-steps: slots = {hi, lo} x 16 channels): products of
//      2-byte operands are exact in the f32 accumulator, so the taps count with 16 (bf16) / 22 (f16) mantissa bits -- the f32 taps of
//      the VALU form to 2^-17; only the summation order differs.
// Measured (profiles/r04_run1.txt, same box, two interleaved runs): leff2 per step 2.948 ms (0) / 2.652 (1) / 3.097 (2); Uformer-B f16 error vs the
// oracle 2.95e-4 / 2.72e-4 / 2.72e-4, bf16 1.89e-3 / 2.06e-3 / 1.92e-3.  Default 1: the split form's four extra k-steps and its per-tap
// hi / lo arithmetic cost more issue slots than the VALU stencil it replaces.
#ifndef UF_MCONV
#define UF_MCONV 1
#endif
#ifndef UF_LEFF2_CP_DEFAULT
#define UF_LEFF2_CP_DEFAULT false
#endif
template <typename T> __device__ __forceinline__ void cvt8(const char* p, float* f);
template <> __device__ __forceinline__ void cvt8<bf16>(const char* p, float* f) { unpack8<bf16>(*reinterpret_cast<const u32x4*>(p), f); }
template <> __device__ __forceinline__ void cvt8<f16>(const char* p, float* f) { unpack8<f16>(*reinterpret_cast<const u32x4*>(p), f); }
template <> __device__ __forceinline__ void cvt8<float>(const char* p, float* f) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 16);
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}
__device__ __forceinline__ void put8(bf16* p, const float* f) { *reinterpret_cast<u32x4*>(p) = pack8<bf16>(f); }
__device__ __forceinline__ void put8(f16* p, const float* f) { *reinterpret_cast<u32x4*>(p) = pack8<f16>(f); }
__device__ __forceinline__ void put8(float* p, const float* f) {
    *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
}

// Wave-specialised kernel.  4*NPG PRODUCER waves (depthwise stencil + GELU on the VALU, and the LDS-DMA prefetch of the halo /
// tap tiles) and NC CONSUMER waves (W2 fragments L2 -> registers, MFMAs, epilogue).  The hardware spreads the waves of a
// workgroup over the 4 SIMDs, and the VALU and matrix pipes of a SIMD run concurrently for different waves, so the stencil
// of interval i overlaps the MFMAs of interval i-1.  An interval = KC = 64*NPG hidden channels: producer group g (4 waves)
// convolves channels [64g, 64g+64) of it.  NPG = 2 (8 stencil waves) is for grids that cannot put a second workgroup on a CU
// (<= 256 tiles, or C = 512 whose consumers need the registers): two stencil waves per SIMD hide each other's LDS latency.
// Halo / tap tiles: NBUF-deep ring in LDS filled by DMA NBUF-1 intervals ahead; operand tile: double buffered; one barrier
// per interval.
// CP (round 4): pixel tiles (of the 4 of an interval) whose stencil the CONSUMER waves take over, one (16-channel group, pixel tile) job per consumer
// wave: 1 with 4 consumers, 2 with 8.  The role stamps (profiles/r04_run2.txt) show the producers as the critical path of an interval (a chain of
// ~2 K cycles) with the consumers parked 50-76 % of the time at C <= 256; a consumer's job runs behind its MFMAs of the previous interval.
template <typename T, int C, int NPG, int NC, int NBUF, int WPS, int PW = 4, int CP = 0>
__global__ __launch_bounds__((PW * NPG + NC) * 64, WPS) void leff2_kernel(const Leff2Params p) {
    constexpr int NP = PW * NPG;                  // producer waves: PW (4 or 8) per 64-channel group of the interval
    constexpr int MC = sizeof(T) == 2 ? UF_MCONV : 0;   // depthwise 3x3 on the MFMA (2-byte operand types), see UF_MCONV
    static_assert(PW == 4 || (PW == 8 && MC != 0), "8 producer waves per group: MFMA stencil only (two pixel tiles per wave)");
    static_assert(CP == 0 || (MC != 0 && PW == 4 && NPG == 1 && NC == 4 * CP), "consumer stencil jobs: MFMA stencil, 4 producers, one job per consumer wave");
    constexpr int PTW = 16 / PW - CP;             // pixel tiles (16 pixels) per producer wave
    constexpr int SR = 2;                         // rows of the column strip one producer thread convolves
    constexpr int SZ = sizeof(T);
    constexpr int TH = 8, TW = 8, BM = 64;
    constexpr int HID = 4 * C;
    constexpr int KC = KCW * NPG;                 // hidden channels per interval
    constexpr int NIT = HID / KC;                 // intervals
    static_assert(HID % KC == 0, "interval width");
    constexpr int HW_ = TW + 2, HT = (TH + 2) * HW_;  // 10 x 10 halo tile
    constexpr int PS = KCW * SZ;                  // LDS pixel stride of a group's halo tile [HT][64], unpadded (DMA image is lane-linear)
    constexpr int CPP = PS / 16;                  // 16-byte pieces per pixel
    constexpr int HGB = (HT * PS + 1023) / 1024 * 1024;   // bytes of one group's halo tile, in whole DMA instructions (1 KiB)
    constexpr int NHG = HGB / 1024;               // DMA instructions per group halo tile
    constexpr int TRB = KC * 4;                   // bytes of one tap-table row [KC] f32; table = 9 tap rows + 1 bias row
    constexpr int TGB = (10 * TRB + 1023) / 1024 * 1024;
    constexpr int NTI = TGB / 1024;
    constexpr int NHI = NPG * NHG;                // halo DMA instructions per interval
    constexpr int NI = NHI + NTI;                 // DMA instructions per interval
    constexpr int NS = (NI + NP - 1) / NP;        // DMA slots per producer wave per interval (uniform: counted vmcnt)
    constexpr int BUFB = NPG * HGB + TGB;         // bytes of one ring slot
    constexpr int SAT = KC * SZ + 16;             // LDS row stride, operand tile [BM][KC]
    constexpr int AT_BYTES = BM * SAT;
    constexpr int WN = (C / 16) < NC ? (C / 16) : NC, WM = NC / WN;   // consumer wave grid (pixels x out channels)
    static_assert(WM <= 4, "consumer grid");
    constexpr int TMW = 4 / WM, TNW = (C / 16) / WN;               // 16x16 tiles per consumer wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ring = smem;                            // [NBUF] {halo tiles [NPG][HT][64] T, taps [10][KC] f32}
    char* At0 = smem + NBUF * BUFB;               // operand tiles [2]
    // last 1 KiB: landing zone of padding DMA slots (zeros)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < NP;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = p.W / TW, tiles_y = p.H / TH;
    // Persistent walk (round 4): this workgroup owns tiles v = blockIdx.x + k * gridDim.x (k = 0 .. nk - 1) of the XCD-aware order, and the interval
    // stream -- DMA ring, operand tiles, one barrier per interval -- runs ACROSS its tiles: flattened interval m = k * NIT + i.  The DMA of the
    // next tile's first intervals flies under the last intervals of the current one and the consumers' epilogue (with the residual rows requested an
    // interval ahead where registers allow) runs beside the producers' first stencil of the next tile, instead of a drained pipeline, a prologue that
    // waits for its first halo tile and an exposed read-modify-write per tile (29-50 % of a workgroup's life at C <= 128: census and role stamps
    // of profiles/r04_run2.txt).  gridDim.x = n_tiles reproduces the one-tile-per-workgroup launch.
    const int G = (int)gridDim.x;
    const int nk = (p.n_tiles - (int)blockIdx.x + G - 1) / G;
    const int NTOT = nk * NIT;
    const int bt = xcd_tile(blockIdx.x, p.n_tiles);               // first tile: census / stamp index
    auto tile_coords = [&](int k, int& b, int& y0, int& x0) {
        const int t = xcd_tile((int)blockIdx.x + k * G, p.n_tiles);
        b = t / (tiles_x * tiles_y);
        const int tr = t - b * (tiles_x * tiles_y);
        y0 = (tr / tiles_x) * TH; x0 = (tr % tiles_x) * TW;
    };
    const T* W2 = reinterpret_cast<const T*>(p.W2);

    // MFMA stencil constants of a wave: the lane's non-zero slot of the block-diagonal weight fragments and the byte offsets of its B fragments
    // (pixel tile 0; tile pt adds two halo rows) per k-step, for the wave's 16-channel group gq
    const unsigned hshift = (fr & 1) * 16;
    unsigned msk[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) msk[d] = ((fg & 1) == (fr >> 3) && d == ((fr & 7) >> 1)) ? 0xffffffffu : 0u;
    const int gq = (producer ? wave : wave - NP) & 3;
    int boff[MC == 2 ? 9 : 5];
#pragma unroll
    for (int ks = 0; ks < (MC == 2 ? 9 : 5); ++ks) {
        int tap = MC == 2 ? ks : 2 * ks + (fg >> 1);
        tap = tap < 9 ? tap : 8;
        const int hy = (fr >> 3) + tap / 3, hx = (fr & 7) + tap % 3;
        boff[ks] = ((hy * HW_ + hx) * 8 + ((gq * 2 + (fg & 1)) ^ (hx & 6))) * 16;
    }
    Census census; census.begin();
    if (producer) {
        // ------------------------------ producers: DMA prefetch + stencil ---------------------------
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the dynamic region
        const unsigned lds_dummy = lds0 + NBUF * BUFB + 2 * AT_BYTES;
        // buffer descriptor over ONE image of h1 (raw, 32-bit offsets, out-of-range lanes read 0 = the conv's zero padding) and this lane's source
        // offset in each of the wave's DMA slots (slot s covers instruction wave + s*NP of the interval): both follow the tile the ISSUE stream is in
        u32x4 rsrc = {0u, 0u, (unsigned)__builtin_amdgcn_readfirstlane(p.H * p.W * HID * SZ), 0x00020000u};
        unsigned voff[NS];
        int issue_k = -1;
        auto set_issue_tile = [&](int k) {
            int b, y0, x0;
            tile_coords(k, b, y0, x0);
            const unsigned long long ia = (unsigned long long)(uintptr_t)(reinterpret_cast<const char*>(p.h1) + (size_t)b * p.H * p.W * HID * SZ);
            rsrc[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ia);
            rsrc[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ia >> 32)) & 0xffffu;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int idx = wave + s * NP;
                voff[s] = 0xffffff00u;                              // out of range -> zeros
                if (idx < NHI) {
                    const int g = idx / NHG, q = (idx - g * NHG) * 64 + lane;    // piece q of group g's halo tile
                    const int hp = q / CPP, part = q - hp * CPP;
                    const int hy = hp / HW_, hx = hp - hy * HW_;
                    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
                    // MFMA stencil: slot `part` of a halo pixel holds its 16-byte piece part ^ (hx & 6) -- the placement that makes the B-fragment
                    // reads (16 lanes = 16 pixels at a 128-byte stride) conflict-free; a lane fetches the piece that belongs where it lands
                    const int piece = MC ? (part ^ (hx & 6)) : part;
                    if (hp < HT && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) voff[s] = (unsigned)(((iy * p.W + ix) * HID + g * KCW) * SZ + piece * 16);
                }
            }
            issue_k = k;
        };
        auto issue = [&](int m) {                               // all DMA of flattened interval m into ring slot m % NBUF
            const int k = m / NIT, it = m - k * NIT;
            if (k != issue_k) set_issue_tile(k);
            const unsigned slot = lds0 + (unsigned)(m % NBUF) * BUFB;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int idx = wave + s * NP;                  // wave-uniform
                if (idx < NHI) {
                    dma_buffer_to_lds(rsrc, voff[s], (unsigned)(it * KC * SZ), slot + (unsigned)idx * 1024u);
                } else if (idx < NI) {
                    const int q = (idx - NHI) * 64 + lane;      // piece q of the tap table: row q / (KC/4), 4 floats
                    int row = q / (KC / 4);
                    const int c4 = (q - row * (KC / 4)) * 4;
                    row = row > 9 ? 9 : row;                    // pieces past the table re-read the bias row into the slack of the slot
                    const float* src = (row < 9 ? p.w9 + (size_t)row * HID : p.bdw) + it * KC + c4;
                    dma_global_to_lds(src, slot + NPG * HGB + (unsigned)(idx - NHI) * 1024u);
                } else {
                    dma_buffer_to_lds(rsrc, 0xffffff00u, 0u, lds_dummy);     // padding slot: keeps the per-wave DMA count uniform
                }
            }
        };
        const int grp = wave / PW;                              // producer group
        const int gt = tid & 255;                               // thread index within the group (VALU stencil)
        const int cvec = gt & 7, sx = (gt >> 3) & 7, sy0 = (gt >> 6) * SR;
        const int pt0 = ((wave % PW) >> 2) * (16 / PW);          // first pixel tile of this producer wave (MFMA stencil)
#pragma unroll
        for (int it = 0; it < NBUF - 1; ++it)
            if (it < NTOT) issue(it);
        wait_dma<0>();
        lds_barrier();                                                          // B0: intervals 0 .. NBUF-2 staged
        unsigned long long tw = 0, tbar = 0, t0 = __builtin_readcyclecounter(), t1;
#pragma unroll 1
        for (int i = 0; i <= NTOT; ++i) {                       // i = flattened interval
            if (i + NBUF - 1 < NTOT) issue(i + NBUF - 1);       // ring slot last read in iteration i-1
            if (i < NTOT) {
                const char* Hs = Ring + (i % NBUF) * BUFB + grp * HGB;
                const float* Wl = reinterpret_cast<const float*>(Ring + (i % NBUF) * BUFB + NPG * HGB) + grp * KCW;
                char* At = At0 + (i & 1) * AT_BYTES + grp * KCW * SZ;
                if constexpr (MC != 0) {
                    // ---- depthwise 3x3 on the MFMA: this wave = the 16-channel group gq of its 64-channel halo tile, pixel tiles pt0 .. pt0 + PTW - 1 ----
                    mconv_job<T, MC, PTW, KC, SAT, 2 * HW_ * PS>(Hs, Wl, At, gq, pt0, boff, msk, hshift, fr, fg);
                } else {
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
                        cvt8<T>(Hs + ((sy0 + r + 1) * HW_ + sx + kx) * PS + cvec * 8 * SZ, f);
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
            }
            t1 = __builtin_readcyclecounter(); tw += t1 - t0; t0 = t1;
            // interval i+1 must have landed before anyone passes the barrier; later intervals stay in flight
            if (NBUF >= 3 && i + NBUF - 1 < NTOT) wait_dma<(NBUF - 2) * NS>(); else wait_dma<0>();
            lds_barrier();
            t1 = __builtin_readcyclecounter(); tbar += t1 - t0; t0 = t1;
        }
        if (p.tbuf && lane == 0 && (bt & 63) == 0) { p.tbuf[((bt >> 6) * 16 + wave) * 4] = tw; p.tbuf[((bt >> 6) * 16 + wave) * 4 + 1] = tbar; p.tbuf[((bt >> 6) * 16 + wave) * 4 + 2] = producer; }
        return;
    }

    // ---------------------------------- consumers ----------------------------------------------------
    const int cw = wave - NP;
    const int wm = cw / WN, wn = cw % WN;
    // W2 fragments through a buffer descriptor: ONE per-lane offset register (lane * fragment bytes) and scalar offsets per
    // (tile, k-step) instead of a 64-bit pointer per tile -- the consumers of C >= 256 have no registers to spare
    // WKS = k-steps of a sub-chunk whose fragments are prefetched together (across the barrier).  The f32 parity variants
    // with 4 output tiles per wave have no registers for that (8 per fragment): they fetch one k-step at a time -- their
    // exact-f32 MFMAs are 16x slower than bf16, so the exposed L2 round trip is a small part of the step.
    constexpr int WKS = (SZ == 4 && TNW >= 4) ? 1 : 2;
    Frag<T> wf[WKS][TNW];
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W2), 0, C * HID * SZ, 0x00020000);
    const int wlane = lane * 8 * SZ;
    auto w2_load = [&](int sub, int ks, int slot) {        // fragment-major W2: 1 KiB (bf16) contiguous per wave load; sub = 64-channel sub-chunk
#pragma unroll
        for (int i = 0; i < TNW; ++i) {
            const int soff = (((wn * TNW + i) * (HID / 32) + sub * 2 + ks) * 64) * 8 * SZ;     // wave-uniform
            if constexpr (SZ == 2) {
                wf[slot][i].v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, soff, 0));
            } else {
                wf[slot][i].lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, soff, 0));
                wf[slot][i].hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane + 16, soff, 0));
            }
        }
    };
    auto w2_issue = [&](int sub) {
        if constexpr (WKS == 2) { w2_load(sub, 0, 0); w2_load(sub, 1, 1); }
    };

    f32x4 acc[TNW][TMW];
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TMW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // iteration j: MFMAs of flattened interval j-1 (operand tile written by the producers in iteration j-1), sub-chunk by sub-chunk;
    // the W2 fragments of the next sub-chunk are requested as soon as the registers are free -- across the barrier between
    // intervals, so their L2 round trip hides under the wait for the stencil.  After the last interval of a tile: epilogue
    // (+ bias + residual, in place on the f32 stream, model.py:987), accumulators back to zero.  Where the registers allow
    // (C <= 128) the residual rows of a tile are requested at the start of its last interval.
    constexpr bool XPRE = SZ == 2 && TNW * TMW <= 4 && CP == 0;        // C <= 64 (16 registers); at C = 128 the 32 registers spill under the 80-register bound
    f32x4 xres[XPRE ? TNW : 1][XPRE ? TMW : 1];
    auto x_ptr = [&](int b, int y0, int x0, int i, int j) -> float* {
        const int n = (wn * TNW + i) * 16 + fg * 4;
        const int pm = (wm * TMW + j) * 16 + fr;
        return p.x + ((size_t)(b * p.H + y0 + (pm >> 3)) * p.W + x0 + (pm & 7)) * p.ld + n;
    };
    w2_issue(0);
    lds_barrier();                                         // B0
    unsigned long long ctw = 0, ctbar = 0, ct0 = __builtin_readcyclecounter(), ct1;
#pragma unroll 1
    for (int j = 0; j <= NTOT; ++j) {
        if (j >= 1) {
            const int m = j - 1, k = m / NIT, it = m - k * NIT;
            int tb = 0, ty0 = 0, tx0 = 0;
            if (it == NIT - 1) {
                tile_coords(k, tb, ty0, tx0);
                if constexpr (XPRE) {
#pragma unroll
                    for (int i = 0; i < TNW; ++i)
#pragma unroll
                        for (int jj = 0; jj < TMW; ++jj) xres[i][jj] = *reinterpret_cast<const f32x4*>(x_ptr(tb, ty0, tx0, i, jj));
                }
            }
            const char* At = At0 + (m & 1) * AT_BYTES;
#pragma unroll
            for (int g = 0; g < NPG; ++g) {
                const int sub = it * NPG + g;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if constexpr (WKS == 1) w2_load(sub, ks, 0);
                    Frag<T> af[TMW];
#pragma unroll
                    for (int jj = 0; jj < TMW; ++jj)
                        load_frag(af[jj], reinterpret_cast<const T*>(At + ((wm * TMW + jj) * 16 + fr) * SAT + (g * KCW + ks * 32 + fg * 8) * SZ));
                    __builtin_amdgcn_sched_barrier(0);   // keep the operand reads of the next k-step behind these MFMAs (registers)
#pragma unroll
                    for (int ii = 0; ii < TNW; ++ii)
#pragma unroll
                        for (int jj = 0; jj < TMW; ++jj) mma16(acc[ii][jj], wf[WKS == 2 ? ks : 0][ii], af[jj]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (sub + 1 < NIT * NPG) w2_issue(sub + 1);
                else if (j < NTOT) w2_issue(0);           // first sub-chunk of the next tile
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it == NIT - 1) {
                const float dscale = p.drop ? p.drop[tb] : 1.0f;
#pragma unroll
                for (int i = 0; i < TNW; ++i) {
                    const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.b2 + (wn * TNW + i) * 16 + fg * 4);
#pragma unroll
                    for (int jj = 0; jj < TMW; ++jj) {
                        float* xp = x_ptr(tb, ty0, tx0, i, jj);
                        f32x4 r;
                        if constexpr (XPRE) r = xres[i][jj]; else r = *reinterpret_cast<const f32x4*>(xp);
                        *reinterpret_cast<f32x4*>(xp) = r + (acc[i][jj] + b2) * dscale;
                        acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
        if constexpr (CP > 0) {
            if (j < NTOT) {       // this wave's stencil job of interval j: group gq, pixel tile 4 - CP + cw / 4 (halo tile landed before the last barrier)
                const char* Hs = Ring + (j % NBUF) * BUFB;
                const float* Wl = reinterpret_cast<const float*>(Ring + (j % NBUF) * BUFB + NPG * HGB);
                mconv_job<T, MC, 1, KC, SAT, 2 * HW_ * PS>(Hs, Wl, At0 + (j & 1) * AT_BYTES, gq, 4 - CP + (cw >> 2), boff, msk, hshift, fr, fg);
            }
        }
        ct1 = __builtin_readcyclecounter(); ctw += ct1 - ct0; ct0 = ct1;
        lds_barrier();
        ct1 = __builtin_readcyclecounter(); ctbar += ct1 - ct0; ct0 = ct1;
    }
    if (p.tbuf && lane == 0 && (bt & 63) == 0) { p.tbuf[((bt >> 6) * 16 + wave) * 4] = ctw; p.tbuf[((bt >> 6) * 16 + wave) * 4 + 1] = ctbar; p.tbuf[((bt >> 6) * 16 + wave) * 4 + 2] = 0; }
    if (cw == 0 && lane == 0) {   // census entry written by the first consumer thread (thread 0 is a producer and has returned)
        Census c2 = census;
        if (p.tbuf) { unsigned long long* o = p.tbuf + 65536 + (size_t)bt * 8; o[0] = c2.t0; o[1] = __builtin_readcyclecounter(); o[2] = c2.r0; o[3] = wall_clock64();
            o[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); }
    }
}

template <typename T, int C, int NPG, int NC, int NBUF, int WPS, int PW = 4, int CP = 0>
int launch_v(const Leff2Params& p, hipStream_t st) {
    constexpr int SZ = sizeof(T), KC = KCW * NPG;
    constexpr int HGB = (100 * KCW * SZ + 1023) / 1024 * 1024, TGB = (10 * KC * 4 + 1023) / 1024 * 1024;
    constexpr int smem = NBUF * (NPG * HGB + TGB) + 2 * 64 * (KC * SZ + 16) + 1024;
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = leff2_kernel<T, C, NPG, NC, NBUF, WPS, PW, CP>;
    static bool lds_done[64] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, lds_done, "leff2")) return rc;
    const long long M = (long long)p.B * p.H * p.W;
    char name[96] = "";
    if (timing_enabled()) snprintf(name, sizeof(name), "leff2_%s_c%d_np%d_nc%d %lldx%dx%d", TypeName<T>::s, C, PW * NPG, NC, M, C, 4 * C);
    {
        ScopedTimer tm(name, 2.0 * M * C * 4 * C + 18.0 * M * 4 * C, (double)M * 4 * C * SZ + (double)M * C * 8, st);
        // persistent launch: at most `resident` workgroups (what a CU holds of this variant x 256 CUs), every one walking ceil(n_tiles / grid) tiles;
        // the grid is a multiple of 8 (XCD-aware tile order) and divides the tiles as evenly as it can.  Measured per stage against one tile per
        // workgroup (profiles/r04_run6.txt, ms per step): C = 32 0.163 -> 0.129, C = 256 (dec1) 0.653 -> 0.621, C = 128 with 1024 tiles 0.339 -> 0.329,
        // but C = 64 0.263 -> 0.267 / 0.137 -> 0.137 and C = 128 with 4096 tiles 0.308 -> 0.332 (three workgroups per CU already cover each other's
        // prologue and epilogue there, and an even split leaves 688 of 768 slots filled) -- so the walk is used where it paid.
        // UF_VARIANT="persist=0" / "persist=1": never / wherever there are more tiles than resident workgroups.
        const int pe = variant("persist", -1);
        const bool pays = C <= 32 || C >= 256 || (C == 128 && p.n_tiles <= 1024);
        const bool persist = pe >= 0 ? pe != 0 : pays;
        constexpr int waves = PW * NPG + NC;
        constexpr int by_lds = (160 * 1024) / smem, by_waves = 32 / waves, by_regs = (WPS * 4) / waves > 0 ? (WPS * 4) / waves : 1;
        constexpr int per_cu = by_lds < by_waves ? (by_lds < by_regs ? by_lds : by_regs) : (by_waves < by_regs ? by_waves : by_regs);
        const int resident = 256 * (per_cu < 1 ? 1 : per_cu);
        int grid = p.n_tiles;
        if (persist && p.n_tiles > resident) {
            const int rounds = (p.n_tiles + resident - 1) / resident;
            grid = ((p.n_tiles + rounds - 1) / rounds + 7) / 8 * 8;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(waves * 64), smem, st, p);
    }
    return check_launch("leff2");
}

// Shape -> variant.  bf16: C <= 128: 8-wave workgroups small enough (51 KiB, <= 80 registers) for three per CU; C = 256: two per
// CU (3-deep DMA ring, 128 registers); C = 512: 4 stencil + 8 MFMA waves, one workgroup per CU (the consumers' accumulators and
// their W2 prefetch need 168 registers).  f32 (parity mode): one configuration per width.
// depth of the halo-tile DMA ring (NBUF) per width (compile-time knobs of the round-5 experiment).  Hypothesis: a workgroup has NBUF - 1 slots in
// flight while it convolves one, so an interval cannot be shorter than (loaded HBM latency) / (NBUF - 1) -- C = 512 with 3 slots: 63 us / 32 intervals
// = 2 us.  Measured (profiles/r05_run8_ring.txt): C = 512 with 3 / 5 / 8 slots 66.2 / 66.3 / 67.0 us, C = 256 (enc3 form) 3 / 6 slots 30.3 / 31.5,
// dec1 on one 16-wave workgroup per CU with 5 / 8 slots 100.9 / 96.2 against 83.3 for two 8-wave workgroups with 3: the ring is NOT the limit;
// the role stamps (profiles/r05_run9_leff2_stamps.txt) show the consumers' interval (2.2 K cycles of work for 0.5 K of MFMA at C = 512) as the
// longer chain.
#ifndef UF_NB512
#define UF_NB512 3
#endif
#ifndef UF_NB256
#define UF_NB256 3
#endif
#ifndef UF_NB256E
#define UF_NB256E 3
#endif
#ifndef UF_L256_BIG
#define UF_L256_BIG 0   // 1: dec1-sized grids on 8 + 8 waves, one workgroup per CU, NBUF = UF_NB256E
#endif
template <typename T, int C>
int launch_c(const Leff2Params& p, hipStream_t st) {
    const int tiles = p.B * (p.H / 8) * (p.W / 8);
    if constexpr (sizeof(T) == 2 && UF_MCONV != 0 && C >= 32) {
        // 8 producer waves per 64-channel group (two pixel tiles each) where the grid is at most one round of workgroups: a producer wave
        // is a serial chain of LDS round trips (taps -> fragments -> halo reads -> MFMA -> GELU -> operand tile: ~2 K cycles per interval
        // for ~250 instructions, scripts/ubench.py stamps2), so with nothing else resident on the CU shorter chains win -- enc2 (C = 128,
        // 1024 tiles) 0.322 -> 0.307 ms per step, enc3 (C = 256, 256 tiles) 0.247 -> 0.234; with several rounds of workgroups the wider
        // workgroups cost residency (dec1 0.631 -> 0.739, C <= 64 +4...9 %): profiles/r04_run2.txt.  UF_VARIANT="leff2=1" forces it, "leff2=2" never.
        const int ev = variant("leff2", 0);
        const bool force = ev == 1, never = ev == 2;
        // (Stencil jobs on the consumer waves -- template parameter CP -- measured -8 % in round 4, profiles/r04_run7.txt, and are not instantiated.)
        if constexpr (C == 128) { if (!never && (force || tiles <= 1024)) return launch_v<T, C, 1, 4, 2, 6, 8>(p, st); }
        else if constexpr (C == 256) { if (!never && (force || tiles <= 256 || UF_L256_BIG)) return launch_v<T, C, 1, 8, UF_NB256E, 4, 8>(p, st); }
        else if constexpr (C == 512) { if (force) return launch_v<T, C, 1, 8, 4, 4, 8>(p, st); }
        else { if (force) return launch_v<T, C, 1, 4, 2, 6, 8>(p, st); }
    }
    if constexpr (sizeof(T) == 2) {
        if constexpr (C <= 128) return launch_v<T, C, 1, 4, 2, 6>(p, st);
        else if constexpr (C == 256) return launch_v<T, C, 1, 4, UF_NB256, 4>(p, st);
        else return launch_v<T, C, 1, 8, UF_NB512, 3>(p, st);
    } else {
        (void)tiles;
        if constexpr (C <= 256) return launch_v<T, C, 1, 4, 2, 2>(p, st);
        else return launch_v<T, C, 1, 4, 2, 2>(p, st);
    }
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
unsigned long long* debug_get_tbuf();
}  // namespace uf

using namespace uf;

namespace uf {
int launch_leff2(const void* h1, const float* w9, const float* bdw, const void* W2, const float* b2, float* x, int ld, int B, int H, int W, int C,
                 uf_dtype dtype, const float* drop, hipStream_t st) {
    UF_REQUIRE(h1 && w9 && bdw && W2 && b2 && x, UF_ERR_NULL, "uf_dwconv_linear2_fwd: null pointer");
    UF_REQUIRE(B > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0, UF_ERR_SHAPE, "uf_dwconv_linear2_fwd: B=%d H=%d W=%d (multiples of 8)", B, H, W);
    UF_REQUIRE(ld >= C && ld % 4 == 0, UF_ERR_ALIGN, "uf_dwconv_linear2_fwd: ld=%d", ld);
    UF_REQUIRE(((uintptr_t)h1 % 16) == 0 && ((uintptr_t)W2 % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w9 % 16) == 0 &&
                   ((uintptr_t)bdw % 16) == 0 && ((uintptr_t)b2 % 16) == 0,
               UF_ERR_ALIGN, "uf_dwconv_linear2_fwd: operands must be 16-byte aligned");
    UF_REQUIRE((long long)H * W * 4LL * C * (long long)dtype_size(dtype) < 0xffffff00LL, UF_ERR_SHAPE,
               "uf_dwconv_linear2_fwd: one image of the hidden tensor must stay under 4 GiB (32-bit buffer offsets)");
    UF_REQUIRE((long long)B * H * W < 0x7fffffffLL / 4, UF_ERR_SHAPE, "uf_dwconv_linear2_fwd: too many tokens");
    Leff2Params p{};
    p.tbuf = debug_get_tbuf();
    p.h1 = h1; p.w9 = w9; p.bdw = bdw; p.W2 = W2; p.b2 = b2; p.x = x; p.ld = ld; p.B = B; p.H = H; p.W = W; p.drop = drop;
    p.n_tiles = B * (H / 8) * (W / 8);
    if (dtype == UF_BF16) return launch_t<bf16>(p, C, st);
    if (dtype == UF_F16) return launch_t<f16>(p, C, st);
    if (dtype == UF_F32) return launch_t<float>(p, C, st);
    set_error("uf_dwconv_linear2_fwd: dtype %d", (int)dtype);
    return UF_ERR_UNSUPPORTED;
}
}  // namespace uf

extern "C" int uf_dwconv_linear2_fwd(const void* h1, const float* w9, const float* bdw, const void* W2, const float* b2,
                                     float* x, int ld, int B, int H, int W, int C, uf_dtype dtype, void* stream) {
    return uf::launch_leff2(h1, w9, bdw, W2, b2, x, ld, B, H, W, C, dtype, nullptr, (hipStream_t)stream);
}
