// The whole attention half of a LeWin block in ONE kernel, one workgroup per 8x8 window
// (reference model.py:951-986):
//
//   x[win] += proj( softmax( (LN1(x)[win] + modulator) Wq^T * scale . ((..) Wk^T)^T + bias + mask ) (..) Wv^T )
//
// Phase 0  LN1 + roll/partition gather (+ modulator) -> Xn [64][C] in LDS (x read once).
// Phase 1  per (head, query-tile group) unit, one wave: q/k/v = Xn W^T on the MFMA with weight
//          fragments streamed L2 -> registers.  Q and K tiles are computed with the weight as the
//          MFMA A operand, V with the activation as A operand; with those two orientations the
//          ACCUMULATOR registers of q, k and v are already laid out exactly as the operand
//          fragments of S^T = K Q^T and O^T = V^T P^T (a lane holds 8 k-slots of one row; only the
//          pairing of slots between the two operands matters).  So q, k, v, the 64x64 scores and P
//          never leave registers: no LDS, no HBM.  Softmax: 16 in-lane values + 2 xor steps.
//          The head's output goes to the O tile [64][C] in LDS.
// Phase 2  proj: out = O Wp^T (+bias), window_reverse + roll back folded into the store index,
//          + residual, in place on the f32 stream.
// Phase 3  (optional, 2-byte operands) first linear of LeFF on the same 64 tokens while their new rows are still in
//          registers: LN2 (two-pass, cross-wave sums through LDS) -> Xn, then h1 = GELU(Xn W1^T + b1) in 64 x 64 units
//          with the weight ring / direct-store epilogue of ln_gemm (model.py:987, :657-658).  Saves the separate
//          ln_fc1 launch, its re-read of x and its LayerNorm phase.
// HBM traffic of the attention half: x once in, x once out, weights from L2 (+ h1 out when phase 3 runs).
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "uf_internal.h"

namespace uf {
namespace {

struct AttnBlkParams {
    float* x; int ld;
    float* xo; int ldo;                    // where phase 2 writes the new rows (= x, ld unless the caller wants them out of place)
    const float* gamma; const float* beta; const float* modulator;
    const void* Wqkv; const float* bqkv;   // T[3C][C], f32[3C]
    const float* rpb_tab;                  // f32[heads][15][15] compact Toeplitz rel-pos bias, x reversed: [dy+7][7-dx]
    const void* Wp; const float* bp;       // T[C][C], f32[C]
    const float* gamma2; const float* beta2;   // phase 3 (h1 != NULL): norm2, linear1 (fragment-major T[4C][C]), its bias
    const void* W1; const float* b1;
    void* h1;                              // T[B*H*W][4C] or NULL
    const float* drop;                     // training: per-image DropPath scale of this branch (bernoulli(keep)/keep, model.py:986) or NULL
    // TR = 1 (training forward that keeps what the backward reads): xn / o T[M][C] in window-row order, q / k T[nW][heads][64][32] (q times
    // head_dim^-0.5), v^T T[nW][heads][32][64] -- the layouts of uf_ln_qkv_fwd / uf_window_attention_fwd, read by uf_linear_wgrad and
    // uf_window_attention_bwd -- and z = LN2(x1) T[M][C] in token order; h1 then receives the PRE-activation of linear1 (a1)
    void* s_xn; void* s_q; void* s_k; void* s_vt; void* s_o; void* s_z;
    int n_windows, H, W, shift;
    float qscale;
    float qscale_plain;                    // head_dim^-0.5 without the log2(e) of the softmax domain: the q the backward reads
    unsigned long long* tbuf;   // optional phase timestamps (uf_debug_set_tbuf)
};

constexpr float LOG2E = 1.4426950408889634f;

// weight-fragment ring depths (k-steps of L2 -> register loads in flight) of the three GEMM phases at C >= 256
#ifndef UF_P1_OFFSET
#define UF_P1_OFFSET 0
#endif
#ifndef UF_FC1_OFFSET
#define UF_FC1_OFFSET 0
#endif
#ifndef UF_STAGGER
#define UF_STAGGER 0
#endif
#ifndef UF_SETPRIO
#define UF_SETPRIO 0
#endif
#define UF_PRIO_UP() do { if (UF_SETPRIO) __builtin_amdgcn_s_setprio(1); } while (0)
#define UF_PRIO_DN() do { if (UF_SETPRIO) __builtin_amdgcn_s_setprio(0); } while (0)
#ifndef UF_ABL
#define UF_ABL 0   // ablations for profiling (1: no x loads, 2: no LN math, 3: no modulator loads, 4: no h1 stores, 5: no row stores in phase 2, 6: no weight-fragment loads in any GEMM phase, 7: no LDS operand-fragment loads, 8: no GELU in phase 3, 9: no softmax arithmetic); 0 in every shipped build
#endif
// UF_NT (experiment, bit mask): non-temporal hints on the activation streams so that they do not evict the block's weights (4 MB at C = 512 = one
// XCD's whole L2) -- 1: h1 stores, 2: the new x rows of phase 2, 4: the x loads of phase 0
#ifndef UF_NT
#define UF_NT 0
#endif
#ifndef UF_LN_ROTATE
#define UF_LN_ROTATE 0
#endif
#ifndef UF_LN_WIDE
#define UF_LN_WIDE 1
#endif
#ifndef UF_QKV_RING
#define UF_QKV_RING 3
#endif
#ifndef UF_PROJ_RING
#define UF_PROJ_RING 3
#endif
#ifndef UF_FC1_RING
#define UF_FC1_RING 5
#endif
// UF_HOIST: the first weight fragments of a GEMM phase are requested BEFORE the phase in front of it (q/k/v weights before the LayerNorm
// of phase 0, the next unit's right behind the current unit's last MFMA, proj weights + residual rows in front of the barrier that ends
// phase 1, fc1 weights in front of LN2): at C <= 128 a phase is 2-4 k-steps long, so the L2 round trip of its first fragments was as
// long as the phase itself and nothing hid it; fc1 weights are also requested in front of the row stores of phase 2 (vector memory
// returns in order and the counter covers stores: fragments requested behind the store burst wait for its last ack).  Bit-identical
// results (same MFMAs on the same operands).  Bit mask of widths it is on for: 1: C <= 128, 2: C = 256, 4: C = 512.  Measured
// (profiles/r03_hoist2_ab.txt, four interleaved rounds): attn_block 3.98 -> 3.88 ms per step with every width on (7), the default.
#ifndef UF_HOIST
#define UF_HOIST 7
#endif
template <int C> constexpr bool hoist_on() { return (C <= 128 && (UF_HOIST & 1)) || (C == 256 && (UF_HOIST & 2)) || (C == 512 && (UF_HOIST & 4)); }

#ifndef UF_ATTN_LR_DEFAULT
#define UF_ATTN_LR_DEFAULT 0
#endif
template <typename T> struct FragFromAcc {   // primary: the 2-byte operand types
    static __device__ __forceinline__ void make(Frag<T>& f, f32x4 a, f32x4 b) {
        f.v = u32x4{pack2<T>(a[0], a[1]), pack2<T>(a[2], a[3]), pack2<T>(b[0], b[1]), pack2<T>(b[2], b[3])};
    }
};
template <> struct FragFromAcc<float> {
    static __device__ __forceinline__ void make(Frag<float>& f, f32x4 a, f32x4 b) { f.lo = a; f.hi = b; }
};

// weight / operand fragment loads of the GEMM phases, with the timing ablations 6 / 7 (registers filled from the lane id instead of memory)
template <typename T> __device__ __forceinline__ void wfrag_load(Frag<T>& f, const T* p) {
    if constexpr (UF_ABL == 6 && sizeof(T) == 2) { const unsigned v = 0x3c003c00u ^ (unsigned)(uintptr_t)p; f.v = u32x4{v, v, v, v}; }
    else load_frag(f, p);
}
template <typename T> __device__ __forceinline__ void afrag_load(Frag<T>& f, const T* p) {
    if constexpr (UF_ABL == 7 && sizeof(T) == 2) { const unsigned v = 0x3c003c00u ^ (unsigned)(uintptr_t)p; f.v = u32x4{v, v, v, v}; }
    else load_frag(f, p);
}

// A fragment made from accumulators is COMPUTED HERE (ST form): the IR-level sinking passes otherwise move the bias add + pack down to the first use
// across the sched_barriers -- the f32 accumulators (twice the registers) then stay alive through the next projection and spill under the 168-register bound
template <typename T> __device__ __forceinline__ void pin_frag(Frag<T>& f) {
    if constexpr (sizeof(T) == 2) asm volatile("" : "+v"(f.v));
}

template <int N> __device__ __forceinline__ float tree_sum(float* v) {   // balanced pairwise sum, N a power of two
    static_assert((N & (N - 1)) == 0, "power of two");
#pragma unroll
    for (int w = 1; w < N; w *= 2)
#pragma unroll
        for (int i = 0; i < N; i += 2 * w) v[i] += v[i + w];
    return v[0];
}

// h1[tok(row)][0..4C) = GELU(Xn[row][:] W1^T + b1) for the 64 rows of a window whose normalised operand tile Xn sits in
// LDS: the barrier-free 64 x 64 unit walk of ln_gemm (uf_lngemm.hip) -- fragment-major weights streamed L2 -> registers
// through a 3-deep ring pinned with sched_barrier, first k-steps of the next unit issued before the epilogue, and the
// permlane-widened direct 16-byte stores.  2-byte operand types only.
// Fc1Walk::first(): the weight fragments of the wave's first unit (callable before the operand tile exists: UF_HOIST);
// Fc1Walk::run(): the walk.
template <typename T, int C, int WAVES, int UW = 4, bool ACT = true>   // ACT = false: the pre-activation is stored (training: the backward needs GELU'(a1))
struct Fc1Walk {
    static_assert(UW == 4 || UW == 2, "units of 64 x 64 or 64 x 32");
    static constexpr int SZ = sizeof(T), KS = C / 32, RING = KS >= 8 ? UF_FC1_RING : (KS >= 4 ? 4 : 3), N4 = 4 * C, UNITS = N4 / (16 * UW);
    const T* wrow[UW];
    Frag<T> wf[RING][UW];
    const T* W1;
    int lane;
    __device__ __forceinline__ void wload(int ks, int slot) {
#pragma unroll
        for (int i = 0; i < UW; ++i) wfrag_load(wf[slot][i], wrow[i] + ks * 512);
    }
    __device__ __forceinline__ void unit_prefetch(int u) {
#pragma unroll
        for (int i = 0; i < UW; ++i) wrow[i] = W1 + ((size_t)(u * UW + i) * KS * 64 + lane) * 8;
#pragma unroll
        for (int s = 0; s < RING - 1; ++s)
            if (s < KS) wload(s, s);
    }
    __device__ __forceinline__ void first(const T* W1_, int wave, int lane_) {
        W1 = W1_; lane = lane_;
        if (wave < UNITS) unit_prefetch(wave);
    }
    __device__ __forceinline__ void run(const char* Xn, int SA, const float* b1, T* h1, const WinGeom& geo, int wave) {
    // RING k-steps of weight fragments in flight: one k-step is only 16 MFMAs (256 cycles) per wave, an L2 round trip under
    // load is 1-2 K cycles -- with 3 slots the walk stalled on every k-step (stamps: 3-4x the MFMA time); 16 registers a slot
    static_assert(SZ == 2, "direct-store epilogue packs pairs of a 2-byte type");
    const int fr = lane & 15, fg = lane >> 4;
    size_t rowoff[4];   // h1 row of this lane's token in each 16-row tile (window_reverse + roll back, as the residual rows)
#pragma unroll
    for (int j = 0; j < 4; ++j) rowoff[j] = (size_t)window_token(geo, j * 16 + fr) * N4;
    if (WAVES == 8 && UF_FC1_OFFSET > 0 && wave >= 4) __builtin_amdgcn_s_sleep(UF_FC1_OFFSET * (C / 256) > 127 ? 127 : UF_FC1_OFFSET * (C / 256));   // as in phase 1
#pragma unroll 1
    for (int u = wave; u < UNITS; u += WAVES) {
        const int nbase = u * 16 * UW;
        f32x4 acc[UW][4];
#pragma unroll
        for (int i = 0; i < UW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* arow = Xn + fr * SA + fg * 8 * SZ;
        Frag<T> af[2][4];
        auto aload = [&](int ks, int slot) {
#pragma unroll
            for (int j = 0; j < 4; ++j) afrag_load(af[slot][j], reinterpret_cast<const T*>(arow + j * 16 * SA + ks * 32 * SZ));
        };
        aload(0, 0);
        f32x4 bv[UW];   // the unit's bias: requested here, consumed after the k-loop (was an exposed L2 round trip per unit)
#pragma unroll
        for (int i = 0; i < UW; ++i) bv[i] = *reinterpret_cast<const f32x4*>(b1 + nbase + i * 16 + fg * 4);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + RING - 1 < KS) wload(ks + RING - 1, (ks + RING - 1) % RING);
            if (ks + 1 < KS) aload(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            UF_PRIO_UP();
#pragma unroll
            for (int i = 0; i < UW; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(acc[i][j], wf[ks % RING][i], af[ks & 1][j]);   // weight as A: lane = 4 channels of one token
            UF_PRIO_DN();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (u + WAVES < UNITS) unit_prefetch(u + WAVES);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int ip = 0; ip < UW; ip += 2) {
                f32x4 va = acc[ip][j] + bv[ip], vb = acc[ip + 1][j] + bv[ip + 1];
                if (UF_ABL != 8 && ACT) { gelu4<T>(va); gelu4<T>(vb); }
                const unsigned a0 = pack2<T>(va[0], va[1]), a1 = pack2<T>(va[2], va[3]);
                const unsigned c0 = pack2<T>(vb[0], vb[1]), c1 = pack2<T>(vb[2], vb[3]);
                const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(a0, c0, false, false);
                const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(a1, c1, false, false);
                const int n = nbase + (ip + (fg & 1)) * 16 + (fg >> 1) * 8;
                if (UF_ABL == 4) asm volatile("" :: "v"(s0[0]), "v"(s1[0]), "v"(s0[1]), "v"(s1[1]), "v"(h1 + rowoff[j] + n));   // ablation: no h1 stores
                else if (UF_NT & 1) __builtin_nontemporal_store(u32x4{s0[0], s1[0], s0[1], s1[1]}, reinterpret_cast<u32x4*>(h1 + rowoff[j] + n));
                else *reinterpret_cast<u32x4*>(h1 + rowoff[j] + n) = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
        }
    }
    }
};

struct NoWalk {   // f32 operands: phase 3 does not exist
    template <typename... A> __device__ __forceinline__ void first(A...) {}
    template <typename... A> __device__ __forceinline__ void run(A...) {}
};

// LR = 1 (round 4, 2-byte operand types at C <= 128): the low-register form.  The q / k / v projections of a unit run ONE AFTER THE OTHER (k, v,
// then q) through one flattened weight ring of two fragments per k-step instead of six, each projection's accumulators turn into operand
// fragments before the next one starts, and phase 3 walks 64 x 32 units: 109 / 168 / 220 registers at C = 32 / 64 / 128 become
// <= 96 / 128 / 168, i.e. 5 / 4 / 3 workgroups per CU instead of 4 / 3 / 2.  These widths are bound by how many independent windows a
// CU holds (a wave issues one instruction per ~5.3 cycles and waits on LDS / L2 round trips between its phases).  Same
// MFMAs on the same operands in the same order per accumulator: bit-identical results to LR = 0.
// LR = 3 (round 6, C = 256 with 4 waves): the SINGLE-OPERAND-TILE form.  The O tile of phase 1 overwrites the
// Xn tile instead of living beside it: a wave keeps the outputs of its two heads in 32 registers (packed to the operand type) until every wave of the
// workgroup has finished its projections -- the last reads of Xn -- and only then are they written, behind a barrier.  LDS 80.9 KB -> 47.1 KB = THREE
// workgroups per CU by LDS; for three by registers (<= 168) it is built on the low-register walk (LR = 1), normalises two rows per pass in phase 0 instead
// of four, fetches the residual rows in the epilogue instead of ahead of the k-loop and adds the bias table one diagonal at a time.  Three independent
// windows per CU = three waves per SIMD whose LayerNorm / softmax / GELU phases (VALU) and load / store bursts overlap each other's MFMAs.  Same MFMAs
// on the same operands in the same order per accumulator: bit-identical to LR = 0.
template <typename T, int C, int NT, int LR = 0, int TR = 0>
__global__ __launch_bounds__(NT, LR == 3 ? 3 : (LR == 1 ? (C <= 32 ? 5 : (C == 64 ? 4 : 3)) : ((sizeof(T) == 2 && C <= 32) ? 4 : ((sizeof(T) == 2 && C == 64) ? 3 : 2)))) void attn_block_kernel(const AttnBlkParams p) {
    constexpr bool ST = LR == 3;
    constexpr int SZ = sizeof(T);
    constexpr int WAVES = NT / 64;
    constexpr int HEADS = C / 32;
    constexpr int SA = C * SZ + 16;                 // LDS row stride of Xn and O
    constexpr int KS = C / 32;                      // k-steps of the projections
    // unit = (head, group of QT query tiles); heads >= 4: one unit per head
    constexpr int QT = HEADS >= 4 ? 4 : HEADS;      // query tiles per unit (4, 2 or 1)
    constexpr int UNITS = HEADS * (4 / QT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xn = smem;
    char* Os = ST ? smem : smem + 64 * SA;                        // ST: O overwrites Xn (behind a barrier, see phase 1)
    float* Tab = reinterpret_cast<float*>(smem + (ST ? 1 : 2) * 64 * SA);   // [HEADS][225] compact rel-pos bias
    float* Red = Tab + HEADS * 225;                               // [2][WAVES][64] LN2 partial sums (phase 3)
    float* Bq = Red + 2 * WAVES * 64;                             // [3C] q/k/v bias + [C] proj bias: read from LDS inside the unit walks, not from L2

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int bw = blockIdx.x;                      // window index (image-major, as window_partition)
    const WinGeom geo = window_geom(bw, p.H, p.W, p.shift);      // window coordinates once (scalar unit), rows by adds
    auto stamp = [&](int k) {
        if (p.tbuf && lane == 0 && (bw & 63) == 0) p.tbuf[((bw >> 6) * WAVES + wave) * 16 + k] = __builtin_readcyclecounter();
    };
    stamp(0);
    Census census; census.begin();
    if (UF_STAGGER > 0 && ((blockIdx.x >> 8) & 1)) {      // experiment: offset the second resident workgroup of a CU by UF_STAGGER x 8 K cycles
#pragma unroll
        for (int k = 0; k < UF_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);
    }

    // The small tables of the later phases (relative-position bias, q/k/v and proj biases) are REQUESTED here, all at once and
    // unconditionally (clamped index), and stored to LDS after phase 0.  They used to be copied by a load -> store loop behind
    // phase 0: 11 dependent L2 round trips per thread at C >= 256 -- 20 K of the 27 K cycles the stamps charged to "LN".
    constexpr int NTABV = HEADS * 225 + 4 * C, NTAB = (NTABV + NT - 1) / NT;
    float tabv[NTAB];
#pragma unroll
    for (int k = 0; k < NTAB; ++k) {
        int i = tid + k * NT;
        i = i < NTABV ? i : NTABV - 1;
        const float* src = i < HEADS * 225 ? p.rpb_tab + i : (i < HEADS * 225 + 3 * C ? p.bqkv + (i - HEADS * 225) : p.bp + (i - HEADS * 225 - 3 * C));
        tabv[k] = *src;
    }
    // q/k/v weight ring of phase 1 (k-steps of L2 -> register loads in flight) and the first unit's prologue, see UF_HOIST
    constexpr bool HOIST = hoist_on<C>() && SZ == 2;      // (the f32 parity variants have no registers to spare)
    constexpr int WR = (SZ == 2 && C >= 256) ? UF_QKV_RING : ((SZ == 2 && C >= 128) ? 3 : 2);
    const T* Wqkv = reinterpret_cast<const T*>(p.Wqkv);
    const T* wrow[6];
    Frag<T> wf[WR][LR ? 2 : 6];
    auto wload = [&](int ks, int slot) {
#pragma unroll
        for (int i = 0; i < 6; ++i) wfrag_load(wf[slot][i], wrow[i] + ks * 512);
    };
    // LR: step g of the flattened walk = (projection g / KS in the order k, v, q; k-step g % KS): two fragments
    auto wload2 = [&](int g, int slot) {
        const int pj = g / KS, ks = g - pj * KS, base = pj == 0 ? 2 : (pj == 1 ? 4 : 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) wfrag_load(wf[slot][i], wrow[base + i] + ks * 512);
    };
    auto unit_weights = [&](int u) {   // 16-row weight tiles of the unit's head in the fragment-major Wqkv (q tiles h*2+i, k tiles C/16+.., v tiles 2C/16+..) + ring prologue
        const int h = u / (4 / QT);
#pragma unroll
        for (int i = 0; i < 6; ++i) wrow[i] = Wqkv + ((size_t)((i >> 1) * (C / 16) + h * 2 + (i & 1)) * KS * 64 + lane) * 8;
#pragma unroll
        for (int pf = 0; pf < WR - 1; ++pf) {
            if constexpr (LR) { if (pf < 3 * KS) wload2(pf, pf); }
            else { if (pf < KS) wload(pf, pf); }
        }
    };
    if constexpr (HOIST) {
        if (wave < UNITS) unit_weights(wave);
    }
    // ---------------- phase 0: LN1 (+gather, +modulator) -> Xn --------------------------------------
    {
        // lanes per row: 16 channels (four 16-byte pieces) per lane where the row is long enough.  The per-ROW work -- two
        // cross-lane reductions, the square root and the division -- is what this phase spends its VALU time on (measured:
        // 27 K cycles at C = 256 with 64 lanes x 4 channels per row, against 4 K for the loads alone), and it is paid per lane:
        // 16 lanes x 16 channels cut the rows a thread owns from 16 to 4 and each reduction from 6 steps to 4.
        constexpr int LPR = UF_LN_WIDE ? ((C / 16) < 8 ? 8 : (C / 16)) : ((C / 4) < 64 ? (C / 4) : 64);
        constexpr int V4 = C / (4 * LPR);
        constexpr int RPP = NT / LPR;
        constexpr int NP = 64 / RPP;
        constexpr int U0 = (16 / V4) < NP ? (16 / V4) : NP;   // 16 x 16-byte loads in flight per thread
        constexpr int U = (ST && U0 > 2) ? 2 : U0;            // ST: half the rows per pass (x and modulator rows of a pass: 128 -> 64 registers)
        static_assert(NP >= 1 && NP % U == 0, "pass batching");
        const int sub = tid % LPR;
        // the order in which a workgroup walks its rows is rotated by the window index: all workgroups start together, and with
        // one fixed order the addresses requested at any moment differ only in the window bits (several address bits are the
        // same chip-wide -> the requests of a round pile up on a subset of the memory channels)
        static_assert((U & (U - 1)) == 0, "U is a power of two");
        const int rot = UF_LN_ROTATE ? bw : 0;
#pragma unroll 1
        for (int r0 = 0; r0 < 64; r0 += RPP * U) {
            f32x4 v[U][V4], md[U][V4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int src = window_token(geo, r0 + ((u + rot) & (U - 1)) * RPP + tid / LPR);
#pragma unroll
                for (int i = 0; i < V4; ++i) {
                    if (UF_ABL == 1) v[u][i] = f32x4{(float)src, (float)i, (float)sub, 1.0f};   // ablation: no x loads
                    else if (UF_NT & 4) v[u][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.x + (size_t)src * p.ld + (i * LPR + sub) * 4));
                    else v[u][i] = *reinterpret_cast<const f32x4*>(p.x + (size_t)src * p.ld + (i * LPR + sub) * 4);
                }
            }
            // the modulator rows of the same tokens, requested together with x: read inside the normalisation loop they were
            // 16 dependent L2 round trips per thread (ablation: 11 K of the 23 K cycles of this phase at C = 256, 14 K at C = 512)
            if (p.modulator && UF_ABL != 3) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int row = r0 + ((u + rot) & (U - 1)) * RPP + tid / LPR;
#pragma unroll
                    for (int i = 0; i < V4; ++i) md[u][i] = *reinterpret_cast<const f32x4*>(p.modulator + (size_t)row * C + (i * LPR + sub) * 4);
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int i = 0; i < V4; ++i) md[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            f32x4 gm[V4], bt[V4];
#pragma unroll
            for (int i = 0; i < V4; ++i) {
                gm[i] = *reinterpret_cast<const f32x4*>(p.gamma + (i * LPR + sub) * 4);
                bt[i] = *reinterpret_cast<const f32x4*>(p.beta + (i * LPR + sub) * 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = r0 + ((u + rot) & (U - 1)) * RPP + tid / LPR;
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
                float rstd;
                if constexpr (SZ == 2) rstd = __builtin_amdgcn_rsqf(sq * (1.0f / C) + 1e-5f);   // 1 ulp: far under the bf16 rounding of the result
                else rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
#pragma unroll
                for (int i = 0; i < V4; ++i) {
                    const int c = (i * LPR + sub) * 4;
                    f32x4 y = v[u][i] * rstd * gm[i] + bt[i] + md[u][i];      // + modulator (model.py:966-969), zeros in the encoder
                    if (UF_ABL == 2) y = v[u][i];                             // ablation: no LN math
                    store4(reinterpret_cast<T*>(Xn + row * SA) + c, y);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NTAB; ++k) {              // requested before phase 0 (see there); Tab | Bq are contiguous in LDS
        const int i = tid + k * NT;
        if (i < HEADS * 225) Tab[i] = tabv[k] * LOG2E;   // scores live in the log2 domain (see softmax)
        else if (i < HEADS * 225 + 4 * C) Bq[i - HEADS * 225] = tabv[k];
    }
    stamp(1);
    lds_barrier();
    stamp(2);
    // TR: an LDS operand tile [64][C] -> global rows, 16-byte pieces, consecutive threads on consecutive pieces of a row
    auto tile_out = [&](const char* tile, void* dst, bool token_order) {
        constexpr int PPR = C * SZ / 16;
#pragma unroll 4
        for (int i = tid; i < 64 * PPR; i += NT) {
            const int r = i / PPR, pc = i - r * PPR;
            const size_t row = token_order ? (size_t)window_token(geo, r) : (size_t)bw * 64 + r;
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(dst) + (row * C * SZ + pc * 16)) = *reinterpret_cast<const u32x4*>(tile + r * SA + pc * 16);
        }
    };
    if constexpr (TR) tile_out(Xn, p.s_xn, false);

    // SW-MSA mask predicate of this window (model.py:924-942), evaluated in registers
    const int nWc = p.W >> 3, nW = (p.H >> 3) * nWc;
    const int wi = bw % nW;
    const bool last_r = p.shift > 0 && (wi / nWc) == (p.H >> 3) - 1;
    const bool last_c = p.shift > 0 && (wi % nWc) == nWc - 1;

    // ---------------- phase 1: per-unit QKV projection + attention, all in registers -------------------
    // 8-wave workgroups put TWO of their waves on every SIMD, and the barriers keep them in lockstep: both fight for the matrix
    // pipe during the projections, then both leave it idle during the softmax.  Holding back the second half of the waves by about
    // one projection lets a SIMD run one wave's MFMAs beside the other's VALU work (UF_P1_OFFSET x 64 cycles, 0 = off).
    if (WAVES == 8 && UF_P1_OFFSET > 0 && wave >= 4) __builtin_amdgcn_s_sleep(UF_P1_OFFSET * (C / 256) > 127 ? 127 : UF_P1_OFFSET * (C / 256));
    constexpr int UPW = (UNITS + WAVES - 1) / WAVES;             // units per wave
    static_assert(!ST || (SZ == 2 && UNITS % WAVES == 0), "single-tile form: 2-byte operands, every wave owns the same number of heads");
    unsigned opk[ST ? UPW : 1][QT][4];                           // ST: the finished heads of this wave, packed, until Xn may be overwritten
    auto unit = [&](const int u, const int ui) __attribute__((always_inline)) {
        const int h = u / (4 / QT), q0 = (u % (4 / QT)) * QT;   // head, first query tile
        Frag<T> qf[QT], kf[4], vtf[2][2];
        if constexpr (LR) {
            // ---- low-register form: k, then v, then q; the ring runs across the three projections (3 KS steps) ----
            if constexpr (!HOIST) unit_weights(u);
            Frag<T> af[2][4];
            auto aload2 = [&](int g, int slot) {      // activation fragments of step g: all four row tiles (k, v) or the unit's query tiles (q)
                const int pj = g / KS, ks = g - pj * KS;
                if (pj < 2 || QT == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) afrag_load(af[slot][j], reinterpret_cast<const T*>(Xn + (j * 16 + fr) * SA + (ks * 32 + fg * 8) * SZ));
                } else {
#pragma unroll
                    for (int j = 0; j < QT; ++j) afrag_load(af[slot][j], reinterpret_cast<const T*>(Xn + ((q0 + j) * 16 + fr) * SA + (ks * 32 + fg * 8) * SZ));
                }
            };
            aload2(0, 0);
            const f32x4 bk0 = *reinterpret_cast<const f32x4*>(Bq + C + h * 32 + fg * 4), bk1 = *reinterpret_cast<const f32x4*>(Bq + C + h * 32 + 16 + fg * 4);
            {
                f32x4 ak[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) ak[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int g = ks;
                    if (g + WR - 1 < 3 * KS) wload2(g + WR - 1, (g + WR - 1) % WR);
                    aload2(g + 1, (g + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) mma16(ak[i][j], wf[g % WR][i], af[g & 1][j]);        // k: weight as A operand
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { FragFromAcc<T>::make(kf[j], ak[0][j] + bk0, ak[1][j] + bk1); if constexpr (ST) pin_frag(kf[j]); }
            }
            {
                f32x4 av[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int g = KS + ks;
                    if (g + WR - 1 < 3 * KS) wload2(g + WR - 1, (g + WR - 1) % WR);
                    aload2(g + 1, (g + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) mma16(av[i][j], af[g & 1][j], wf[g % WR][i]);        // v: activation as A operand
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float bv0 = Bq[2 * C + h * 32 + fr], bv1 = Bq[2 * C + h * 32 + 16 + fr];
#pragma unroll
                for (int sk = 0; sk < 2; ++sk) {
                    FragFromAcc<T>::make(vtf[0][sk], av[0][2 * sk] + bv0, av[0][2 * sk + 1] + bv0);
                    FragFromAcc<T>::make(vtf[1][sk], av[1][2 * sk] + bv1, av[1][2 * sk + 1] + bv1);
                    if constexpr (ST) { pin_frag(vtf[0][sk]); pin_frag(vtf[1][sk]); }
                }
            }
            {
                f32x4 aq[2][QT];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < QT; ++j) aq[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int g = 2 * KS + ks;
                    if (g + WR - 1 < 3 * KS) wload2(g + WR - 1, (g + WR - 1) % WR);
                    if (ks + 1 < KS) aload2(g + 1, (g + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < QT; ++j) mma16(aq[i][j], wf[g % WR][i], af[g & 1][j]);       // q: weight as A operand
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (u == wave) stamp(3);
                if constexpr (HOIST) {
                    if (u + WAVES < UNITS) unit_weights(u + WAVES);
                }
                const f32x4 bq0 = *reinterpret_cast<const f32x4*>(Bq + h * 32 + fg * 4), bq1 = *reinterpret_cast<const f32x4*>(Bq + h * 32 + 16 + fg * 4);
#pragma unroll
                for (int j = 0; j < QT; ++j) FragFromAcc<T>::make(qf[j], (aq[0][j] + bq0) * p.qscale, (aq[1][j] + bq1) * p.qscale);
            }
        } else {
        f32x4 aq[2][QT], ak[2][4], av[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < QT; ++j) aq[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) { ak[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; av[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        // weight fragments come from L2 (>= 500 cycles): ring of WR k-steps in flight where registers allow (declared in front of
        // phase 0); the activation fragments come from LDS, one step ahead is enough
        Frag<T> af[2][4];
        Frag<T> afq[2][QT < 4 ? QT : 1];   // query-tile fragments when q0 is a runtime value (static register indexing only)
        if constexpr (!HOIST) unit_weights(u);
        auto aload = [&](int ks, int slot) {
#pragma unroll
            for (int j = 0; j < 4; ++j) afrag_load(af[slot][j], reinterpret_cast<const T*>(Xn + (j * 16 + fr) * SA + (ks * 32 + fg * 8) * SZ));
            if constexpr (QT < 4) {
#pragma unroll
                for (int j = 0; j < QT; ++j)
                    afrag_load(afq[slot][j], reinterpret_cast<const T*>(Xn + ((q0 + j) * 16 + fr) * SA + (ks * 32 + fg * 8) * SZ));
            }
        };
        aload(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + WR - 1 < KS) wload(ks + WR - 1, (ks + WR - 1) % WR);
            if (ks + 1 < KS) aload(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int s = ks & 1, sw = ks % WR;
            UF_PRIO_UP();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < QT; ++j) {                                                 // q: weight as A operand
                    if constexpr (QT < 4) mma16(aq[i][j], wf[sw][i], afq[s][j]);
                    else mma16(aq[i][j], wf[sw][i], af[s][j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(ak[i][j], wf[sw][2 + i], af[s][j]);        // k: weight as A operand
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(av[i][j], af[s][j], wf[sw][4 + i]);        // v: activation as A operand
            }
            UF_PRIO_DN();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (u == wave) stamp(3);
        if constexpr (HOIST) {      // the next unit's first weight fragments fly under this unit's attention (the ring is free)
            if (u + WAVES < UNITS) unit_weights(u + WAVES);
        }
        // bias (+ scale on q, model.py:497), then the accumulators ARE the attention operands
        {
            const f32x4 bq0 = *reinterpret_cast<const f32x4*>(Bq + h * 32 + fg * 4), bq1 = *reinterpret_cast<const f32x4*>(Bq + h * 32 + 16 + fg * 4);
            const f32x4 bk0 = *reinterpret_cast<const f32x4*>(Bq + C + h * 32 + fg * 4), bk1 = *reinterpret_cast<const f32x4*>(Bq + C + h * 32 + 16 + fg * 4);
            const float bv0 = Bq[2 * C + h * 32 + fr], bv1 = Bq[2 * C + h * 32 + 16 + fr];
#pragma unroll
            for (int j = 0; j < QT; ++j) FragFromAcc<T>::make(qf[j], (aq[0][j] + bq0) * p.qscale, (aq[1][j] + bq1) * p.qscale);   // qscale carries log2(e)
#pragma unroll
            for (int j = 0; j < 4; ++j) FragFromAcc<T>::make(kf[j], ak[0][j] + bk0, ak[1][j] + bk1);
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                FragFromAcc<T>::make(vtf[0][sk], av[0][2 * sk] + bv0, av[0][2 * sk + 1] + bv0);
                FragFromAcc<T>::make(vtf[1][sk], av[1][2 * sk] + bv1, av[1][2 * sk + 1] + bv1);
            }
            if constexpr (TR && SZ == 2) {
                // what the backward reads, in the layouts of the three-kernel path: q = T((xn Wq^T + bq) head_dim^-0.5) (its own rounding: the
                // fragment above carries log2(e) as well), k and v^T straight from the operand fragments (8 bytes = 4 channels / 4 tokens per lane)
                const size_t wh = (size_t)bw * HEADS + h;
                T* qb = reinterpret_cast<T*>(p.s_q) + wh * 2048;
#pragma unroll
                for (int j = 0; j < QT; ++j) {
                    store4(qb + ((q0 + j) * 16 + fr) * 32 + fg * 4, (aq[0][j] + bq0) * p.qscale_plain);
                    store4(qb + ((q0 + j) * 16 + fr) * 32 + 16 + fg * 4, (aq[1][j] + bq1) * p.qscale_plain);
                }
                if (q0 == 0) {              // units of one head differ only in their query tiles: the first one writes the head's k and v
                    T* kb = reinterpret_cast<T*>(p.s_k) + wh * 2048;
                    T* vb = reinterpret_cast<T*>(p.s_vt) + wh * 2048;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        *reinterpret_cast<u32x2*>(kb + (j * 16 + fr) * 32 + fg * 4) = u32x2{kf[j].v[0], kf[j].v[1]};
                        *reinterpret_cast<u32x2*>(kb + (j * 16 + fr) * 32 + 16 + fg * 4) = u32x2{kf[j].v[2], kf[j].v[3]};
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int sk = 0; sk < 2; ++sk) {
                            *reinterpret_cast<u32x2*>(vb + (i * 16 + fr) * 64 + (2 * sk) * 16 + fg * 4) = u32x2{vtf[i][sk].v[0], vtf[i][sk].v[1]};
                            *reinterpret_cast<u32x2*>(vb + (i * 16 + fr) * 64 + (2 * sk + 1) * 16 + fg * 4) = u32x2{vtf[i][sk].v[2], vtf[i][sk].v[3]};
                        }
                }
            }
        }
        }
        // S^T = K Q^T : s[kt][j] -> lane: query (q0+j)*16+fr, keys 16kt+4fg+r
        f32x4 s[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int j = 0; j < QT; ++j) {
                s[kt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                mma16(s[kt][j], kf[kt], qf[j]);
            }
        if (u == wave) stamp(8);
        // relative-position bias (model.py:500-506).  bias[q][k] depends only on (yq-yk, xq-xk): for this
        // lane (query column fr, key group fg) the 4 keys of a tile are 4 consecutive dx, and over the
        // (query tile, key tile) pairs dy takes 7 values -> 7 x 4 table entries per unit, read from the
        // compact table in LDS instead of 16 KiB of dense bias per (window, head) from L2.
        // Softmax in the log2 domain: q was scaled by scale*log2(e) and the table by log2(e), so
        // exp(s - max) = exp2(s' - max') is one v_sub + v_exp per score (no multiply); the SW-MSA mask
        // (-100, model.py:924-942) becomes -100*log2(e) and is applied only in the windows that have one
        // (last window row / column of a shifted block): a wave-uniform branch, interior windows skip it.
        {
            const int dyc = (fr >> 3) - (fg >> 1);
            const int xb = 7 - (fr & 7) + 4 * (fg & 1);
            const float* th = Tab + h * 225 + xb;
            auto tb_row = [&](int d) {
                int row = 2 * (q0 + d - 3) + dyc + 7;      // dy + 7 for (query tile - key tile) = q0 + d - 3 ... (d = j - kt + 3)
                row = row < 0 ? 0 : (row > 14 ? 14 : row);  // rows outside [0,14] belong to unused (j,kt) pairs
                const float* tr = th + row * 15;
                return f32x4{tr[0], tr[1], tr[2], tr[3]};
            };
            if constexpr (ST) {           // one diagonal of (query tile, key tile) pairs at a time: 4 registers of table values live instead of 28
#pragma unroll
                for (int d = 0; d < 7; ++d) {
                    const f32x4 tbd = tb_row(d);
#pragma unroll
                    for (int j = 0; j < QT; ++j)
                        if (j + 3 - d >= 0 && j + 3 - d < 4) s[j + 3 - d][j] += tbd;
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                f32x4 tb[7];
#pragma unroll
                for (int d = 0; d < 7; ++d) tb[d] = tb_row(d);
#pragma unroll
                for (int j = 0; j < QT; ++j)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) s[kt][j] += tb[j - kt + 3];
            }
        }
        if (last_r || last_c) {
#pragma unroll
            for (int j = 0; j < QT; ++j) {
                const int qi = (q0 + j) * 16 + fr;
                const bool q_lo_y = (qi >> 3) >= 4, q_lo_x = (qi & 7) >= 4;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const int k0 = kt * 16 + fg * 4;
                    const bool dy = last_r && (((k0 >> 3) >= 4) != q_lo_y);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool k_lo_x = ((k0 & 7) + r) >= 4;
                        if (dy || (last_c && (k_lo_x != q_lo_x))) s[kt][j][r] += -100.0f * LOG2E;
                    }
                }
            }
        }
        float inv[QT];
#pragma unroll
        for (int j = 0; j < QT; ++j) {
            float mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][j][r]);
            mx = red_xor32<RedMax>(red_xor16<RedMax>(mx));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = UF_ABL == 9 ? s[kt][j][r] - mx : __builtin_amdgcn_exp2f(s[kt][j][r] - mx);
                    s[kt][j][r] = e;
                    sum += e;
                }
            sum = red_xor32<RedSum>(red_xor16<RedSum>(sum));
            inv[j] = 1.0f / sum;
        }
        if (u == wave) stamp(9);
        // O^T = V^T P^T ; o[dt][j]: lane query fr, d = 16dt+4fg+r
        f32x4 o[2][QT];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int j = 0; j < QT; ++j) o[dt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sk = 0; sk < 2; ++sk)
#pragma unroll
            for (int j = 0; j < QT; ++j) {
                Frag<T> pf;
                FragFromAcc<T>::make(pf, s[2 * sk][j], s[2 * sk + 1][j]);
                mma16(o[0][j], vtf[0][sk], pf);
                mma16(o[1][j], vtf[1][sk], pf);
            }
        if (u == wave) stamp(10);
        // head merge (model.py:519): O[token][h*32 + d]
#pragma unroll
        for (int j = 0; j < QT; ++j) {
            if constexpr (ST) {           // parked in registers: other waves still read Xn, which the O tile overwrites
                const f32x4 oa = o[0][j] * inv[j], ob = o[1][j] * inv[j];
                opk[ui][j][0] = pack2<T>(oa[0], oa[1]); opk[ui][j][1] = pack2<T>(oa[2], oa[3]);
                opk[ui][j][2] = pack2<T>(ob[0], ob[1]); opk[ui][j][3] = pack2<T>(ob[2], ob[3]);
                // pinned: left to itself the scheduler keeps the 8 f32 accumulators alive across the next unit and packs them at the end (32 spilled registers)
                asm volatile("" : "+v"(opk[ui][j][0]), "+v"(opk[ui][j][1]), "+v"(opk[ui][j][2]), "+v"(opk[ui][j][3]));
            } else {
                T* orow = reinterpret_cast<T*>(Os + ((q0 + j) * 16 + fr) * SA) + h * 32 + fg * 4;
                store4(orow, o[0][j] * inv[j]);
                store4(orow + 16, o[1][j] * inv[j]);
            }
        }
        if (u == wave) stamp(4);
    };
    if constexpr (ST) {
#pragma unroll
        for (int ui = 0; ui < UPW; ++ui) unit(wave + ui * WAVES, ui);
        lds_barrier();                                            // every wave has made its last read of Xn
#pragma unroll
        for (int ui = 0; ui < UPW; ++ui) {
            const int u = wave + ui * WAVES, h = u / (4 / QT), q0 = (u % (4 / QT)) * QT;
#pragma unroll
            for (int j = 0; j < QT; ++j) {
                char* orow = Os + ((q0 + j) * 16 + fr) * SA + (h * 32 + fg * 4) * SZ;
                *reinterpret_cast<u32x2*>(orow) = u32x2{opk[ui][j][0], opk[ui][j][1]};
                *reinterpret_cast<u32x2*>(orow + 16 * SZ) = u32x2{opk[ui][j][2], opk[ui][j][3]};
            }
        }
    } else {
#pragma unroll 1
        for (int u = wave; u < UNITS; u += WAVES) unit(u, 0);
    }
    stamp(5);
    std::conditional_t<SZ == 2, Fc1Walk<T, C, WAVES, LR ? 2 : 4, TR == 0>, NoWalk> fc1w;     // phase 3's weight ring: its first fragments are requested in front of LN2 (UF_HOIST)
    // ---------------- phase 2: proj + window_reverse + roll back + residual ---------------------------
    {
        constexpr int WN = (C / 16) < WAVES ? (C / 16) : WAVES, WM = WAVES / WN;
        constexpr int TNW = (C / 16) / WN, TMW = (4 / WM) > 0 ? (4 / WM) : 1;
        static_assert(WM <= 4, "more waves than 16-row tiles");
        const int wm = wave / WN, wn = wave % WN;
        const T* Wp = reinterpret_cast<const T*>(p.Wp);
        f32x4 acc[TNW][TMW];
#pragma unroll
        for (int i = 0; i < TNW; ++i)
#pragma unroll
            for (int j = 0; j < TMW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int PR = SZ == 2 ? (KS >= 8 ? UF_PROJ_RING : 3) : 2;   // weight ring depth (k-steps in flight)
        Frag<T> wf[PR][TNW], af[2][TMW];
        auto wload = [&](int ks, int slot) {
#pragma unroll
            for (int i = 0; i < TNW; ++i) wfrag_load(wf[slot][i], Wp + (((size_t)(wn * TNW + i) * KS + ks) * 64 + lane) * 8);
        };
        auto aload = [&](int ks, int slot) {
#pragma unroll
            for (int j = 0; j < TMW; ++j) afrag_load(af[slot][j], reinterpret_cast<const T*>(Os + ((wm * TMW + j) * 16 + fr) * SA + (ks * 32 + fg * 8) * SZ));
        };
        // the rows this wave updates: addresses now, and (2-byte operand types: registers allow it) the residual values and
        // the bias requested BEFORE the k-loop, so that their round trip (L2: the rows were read in phase 0) hides under it
        float* xrow[TMW];
        float* xorow[TMW];
#pragma unroll
        for (int j = 0; j < TMW; ++j) {
            const size_t tok = (size_t)window_token(geo, (wm * TMW + j) * 16 + fr);
            xrow[j] = p.x + tok * p.ld;
            xorow[j] = p.xo + tok * p.ldo;
        }
        constexpr bool PRE = SZ == 2 && !ST;      // ST: 64 registers it does not have
        f32x4 res[PRE ? TNW : 1][PRE ? TMW : 1];
        auto proj_requests = [&]() {       // what does not depend on the O tile: weight prologue, residual rows
#pragma unroll
            for (int pf = 0; pf < PR - 1; ++pf)
                if (pf < KS) wload(pf, pf);
            if constexpr (PRE) {
#pragma unroll
                for (int i = 0; i < TNW; ++i) {
                    const int n = (wn * TNW + i) * 16 + fg * 4;
#pragma unroll
                    for (int j = 0; j < TMW; ++j) res[i][j] = *reinterpret_cast<const f32x4*>(xrow[j] + n);
                }
            }
        };
        if constexpr (HOIST) proj_requests();      // in flight across the barrier (UF_HOIST): waves that finish phase 1 early wait there anyway
        lds_barrier();
        stamp(6);
        if constexpr (TR) tile_out(Os, p.s_o, false);
        if constexpr (!HOIST) proj_requests();
        aload(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PR - 1 < KS) wload(ks + PR - 1, (ks + PR - 1) % PR);
            if (ks + 1 < KS) aload(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            UF_PRIO_UP();
#pragma unroll
            for (int i = 0; i < TNW; ++i)
#pragma unroll
                for (int j = 0; j < TMW; ++j) mma16(acc[i][j], wf[ks % PR][i], af[ks & 1][j]);
            UF_PRIO_DN();
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(7);
        if constexpr (SZ == 2 && HOIST) {
            // phase 3's first weight fragments are requested BEFORE this phase's row stores: vector memory returns in order and the
            // counter covers stores too, so fragments requested behind the 64 x C x 4-byte store burst would wait for its last ack
            if (p.h1) fc1w.first(reinterpret_cast<const T*>(p.W1), wave, lane);
        }
        const float dscale = p.drop ? p.drop[geo.img] : 1.0f;   // DropPath: x + scale_b * branch (timm, train mode only)
#pragma unroll
        for (int j = 0; j < TMW; ++j) {
            float* xr = xrow[j];
            float* xw = xorow[j];
#pragma unroll
            for (int i = 0; i < TNW; ++i) {
                const int n = (wn * TNW + i) * 16 + fg * 4;
                const f32x4 b = *reinterpret_cast<const f32x4*>(Bq + 3 * C + n);
                if constexpr (PRE) acc[i][j] = res[i][j] + (acc[i][j] + b) * dscale;   // the block's new rows stay in registers
                else acc[i][j] = *reinterpret_cast<const f32x4*>(xr + n) + (acc[i][j] + b) * dscale;
                if (UF_ABL == 5) asm volatile("" :: "v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]), "v"(xw + n));   // ablation: no row stores
                else if (UF_NT & 2) __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4*>(xw + n));
                else *reinterpret_cast<f32x4*>(xw + n) = acc[i][j];
            }
        }
        if constexpr (SZ == 2) {
            if (p.h1) {
                // ---- LN2 of the new rows (model.py:987): two-pass mean / variance; a token's C channels are spread over
                // the 4 lane groups of a wave (xor 16, 32) and the WN waves of its row group (LDS).  All sums are balanced
                // binary trees over the 16-channel tiles, so the result does not depend on how many waves share a row
                // (the 4- and 8-wave variants of one C must agree bit for bit: batch-size independence).
                float mean[TMW], rstd[TMW];
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                    for (int j = 0; j < TMW; ++j) {
                        float part[TNW];
#pragma unroll
                        for (int i = 0; i < TNW; ++i) {
                            if (pass == 0) part[i] = (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
                            else {
                                const f32x4 d = acc[i][j] - mean[j];
                                part[i] = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                            }
                            part[i] = red_xor32<RedSum>(red_xor16<RedSum>(part[i]));   // the tile's 16 channels
                        }
                        const float sacc = tree_sum<TNW>(part);
                        if (fg == 0) Red[(pass * WAVES + wave) * 64 + (wm * TMW + j) * 16 + fr] = sacc;
                    }
                    lds_barrier();
#pragma unroll
                    for (int j = 0; j < TMW; ++j) {
                        float wsum[WN];
#pragma unroll
                        for (int w2 = 0; w2 < WN; ++w2) wsum[w2] = Red[(pass * WAVES + wm * WN + w2) * 64 + (wm * TMW + j) * 16 + fr];
                        const float tot = tree_sum<WN>(wsum);
                        if (pass == 0) mean[j] = tot * (1.0f / C);
                        else rstd[j] = 1.0f / sqrtf(tot * (1.0f / C) + 1e-5f);
                    }
                }
#pragma unroll
                for (int i = 0; i < TNW; ++i) {
                    const int n = (wn * TNW + i) * 16 + fg * 4;
                    const f32x4 g2 = *reinterpret_cast<const f32x4*>(p.gamma2 + n), b2 = *reinterpret_cast<const f32x4*>(p.beta2 + n);
#pragma unroll
                    for (int j = 0; j < TMW; ++j)
                        store4(reinterpret_cast<T*>(Xn + ((wm * TMW + j) * 16 + fr) * SA) + n, (acc[i][j] - mean[j]) * rstd[j] * g2 + b2);
                }
            }
        }
    }
    if constexpr (SZ == 2) {
        if (p.h1) {
            lds_barrier();
            stamp(11);
            if constexpr (TR) tile_out(Xn, p.s_z, true);
            if constexpr (!HOIST) fc1w.first(reinterpret_cast<const T*>(p.W1), wave, lane);
            fc1w.run(Xn, SA, p.b1, reinterpret_cast<T*>(p.h1), geo, wave);
            stamp(12);
        }
    }
    census.end(p.tbuf, bw);
}

template <typename T, int C, int NT, int LR = 0, int TR = 0>
int launch_one(const AttnBlkParams& p, hipStream_t st) {
    constexpr int smem = (LR == 3 ? 1 : 2) * 64 * (C * (int)sizeof(T) + 16) + (C / 32) * 225 * 4 + 2 * (NT / 64) * 64 * 4 + 4 * C * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = attn_block_kernel<T, C, NT, LR, TR>;
    static bool lds_done[64] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, lds_done, "attn_block")) return rc;
    char name[96] = "";
    if (timing_enabled())
        snprintf(name, sizeof(name), "attn_block%s_%s_c%d_nt%d %dx%d", TR ? "_train" : (p.h1 ? "_fc1" : ""), TypeName<T>::s, C, NT, p.n_windows * 64, C);
    const double M = (double)p.n_windows * 64;
    const double fc1_flops = p.h1 ? 2.0 * M * C * 4.0 * C : 0.0, fc1_bytes = p.h1 ? M * 4.0 * C * sizeof(T) + 4.0 * C * C * sizeof(T) : 0.0;
    {
        const double side_bytes = TR ? 6.0 * M * C * sizeof(T) : 0.0;     // training form: xn, q, k, v^T, o, z stored beside x1 and a1
        ScopedTimer tm(name, 2.0 * M * C * (4.0 * C + 128.0) + fc1_flops, M * C * 8.0 + 4.0 * C * C * sizeof(T) + fc1_bytes + side_bytes, st);
        hipLaunchKernelGGL(kern, dim3(p.n_windows), dim3(NT), smem, st, p);
    }
    return check_launch("attn_block");
}

}  // namespace

// development aid (uf_debug_set_tbuf): one pointer, read by every launch -- atomic so that a thread switching it on or off
// never tears a concurrent launch's read
static std::atomic<unsigned long long*> g_tbuf{nullptr};
void debug_set_tbuf(void* p) { g_tbuf.store((unsigned long long*)p, std::memory_order_release); }
unsigned long long* debug_get_tbuf() { return g_tbuf.load(std::memory_order_acquire); }

// true when the fused kernel covers (dtype, C, head_dim); otherwise the caller uses the 3-kernel path
bool attn_block_supported(const uf_block_params* bp, const float* user_mask, uf_dtype dtype, int C, int heads) {
    // needs the compact (Toeplitz) bias table -- any index buffer built like the reference's (model.py:467-477) has
    // one -- and no caller-supplied mask; everything else takes the 3-kernel path
    if (!bp->rpb_tab || user_mask) return false;
    if (heads <= 0 || C != heads * 32) return false;
    if (dtype_half(dtype)) return C == 32 || C == 64 || C == 128 || C == 256 || C == 512;
    return C == 32 || C == 64 || C == 128 || C == 256;   // f32: two [64][C] tiles must fit LDS
}

int launch_attn_block(const uf_block_params* bp, float* x, int ld, int B, int H, int W, int C, uf_dtype dtype, void* h1_out, hipStream_t st, const float* drop,
                      float* xo, int ldo) {
    AttnBlkParams p{};
    p.drop = drop;
    p.x = x; p.ld = ld;
    p.xo = xo ? xo : x; p.ldo = xo ? ldo : ld; p.gamma = bp->norm1_w; p.beta = bp->norm1_b; p.modulator = bp->modulator;
    p.Wqkv = bp->wqkv_fm; p.bqkv = bp->bqkv; p.rpb_tab = bp->rpb_tab;
    p.Wp = bp->wproj_fm; p.bp = bp->bproj;
    p.gamma2 = bp->norm2_w; p.beta2 = bp->norm2_b; p.W1 = bp->w1_fm; p.b1 = bp->b1;
    p.h1 = dtype_half(dtype) ? h1_out : nullptr;   // phase 3 exists for 2-byte operands only
    p.n_windows = B * (H / 8) * (W / 8); p.H = H; p.W = W; p.shift = bp->shift;
    p.qscale = (float)(1.0 / sqrt(32.0)) * LOG2E;   // q = q * scale (model.py:497), times log2(e): softmax via exp2
    p.tbuf = debug_get_tbuf();

    // Form by shape; UF_VARIANT="attn=k" forces one (A/B runs, bit-identity test): 0 the first form, 1 the low-register form at C <= 128 with the tighter
    // register bound (one more workgroup per CU), 2 the same code at the occupancy of the first form, 3 the single-operand-tile form at C = 256.
    // Defaults from the same-box A/B of the bit-identical forms (profiles/r04_run4.txt, ms per step: first / 1 / 2):
    // C = 32 (enc0) 0.187 / 0.212 / 0.172 -> 2; C = 128 with >= 4096 windows (dec2) 0.409 / 0.383 / 0.406 -> 1; everything else stays
    // on the first form (C = 128 with 1024 windows 0.445 / 0.445 / 0.467, C = 64 0.361 / 0.373 / 0.366 and 0.177 / 0.179 / 0.183).
    // The single-tile form (3) is NOT a default: bit-identical, three workgroups per CU, 128 vs 136 us on the isolated dec1 launch but 123 vs 117 us
    // inside the model and the MFMA pipe busy 26.5 % instead of 27.7 % (profiles/r06_run3_stages.txt, r06_run4_ab.txt, r06_pmc?_st?_attn_leff.csv).
    const int forced = variant("attn", -1);
    const int lr = (forced >= 0 && forced <= 2) ? forced : (forced == 3 ? 0 : (C == 32 ? 2 : ((C == 128 && p.n_windows >= 4096) ? 1 : 0)));
    const bool st256 = forced == 3;
#define UF_AB(TT, CV, NTV) return launch_one<TT, CV, NTV>(p, st)
#define UF_AB_HALF(TT)                                                                                                              \
        switch (C) {                                                                                                                    \
            case 32: if (lr == 1) return launch_one<TT, 32, 256, 1>(p, st); if (lr == 2) return launch_one<TT, 32, 256, 2>(p, st); UF_AB(TT, 32, 256);      \
            case 64: if (lr == 1) return launch_one<TT, 64, 256, 1>(p, st); if (lr == 2) return launch_one<TT, 64, 256, 2>(p, st); UF_AB(TT, 64, 256);      \
            case 128: if (lr == 1) return launch_one<TT, 128, 256, 1>(p, st); if (lr == 2) return launch_one<TT, 128, 256, 2>(p, st); UF_AB(TT, 128, 256);                                                                                              \
            case 256:                                                                                                                   \
                if (p.n_windows <= 256) UF_AB(TT, 256, 512);   /* one workgroup per CU at most: 8 waves (one head each) instead of 4 */ \
                if (st256) return launch_one<TT, 256, 256, 3>(p, st);   /* single-operand-tile form: three workgroups per CU */       \
                UF_AB(TT, 256, 256);                                                                                                    \
            case 512: UF_AB(TT, 512, 512);                                                                                              \
        }
    if (dtype == UF_BF16) {
        UF_AB_HALF(bf16)
    } else if (dtype == UF_F16) {
        UF_AB_HALF(f16)
    } else {
        switch (C) {
            case 32: UF_AB(float, 32, 256);
            case 64: UF_AB(float, 64, 256);
            case 128: UF_AB(float, 128, 256);
            case 256: UF_AB(float, 256, 256);
        }
    }
#undef UF_AB_HALF
#undef UF_AB
    set_error("attn_block: unsupported C=%d for dtype %d", C, (int)dtype);
    return UF_ERR_UNSUPPORTED;
}


// Training forward of the attention half + linear1: the fused kernel with side stores of every operand the
// backward reads, x1 out of place (the block's input stays for the LayerNorm backward).  2-byte operand types, head_dim 32.
int launch_attn_block_train(const uf_block_params* bp, const float* x, int ld, float* x1, int ld1, int B, int H, int W, int C, uf_dtype dtype, const float* drop,
                            void* xn, void* q, void* k, void* vt, void* o, void* z, void* a1, hipStream_t st) {
    AttnBlkParams p{};
    p.drop = drop;
    p.x = const_cast<float*>(x); p.ld = ld;            // read only: xo != x
    p.xo = x1; p.ldo = ld1; p.gamma = bp->norm1_w; p.beta = bp->norm1_b; p.modulator = bp->modulator;
    p.Wqkv = bp->wqkv_fm; p.bqkv = bp->bqkv; p.rpb_tab = bp->rpb_tab;
    p.Wp = bp->wproj_fm; p.bp = bp->bproj;
    p.gamma2 = bp->norm2_w; p.beta2 = bp->norm2_b; p.W1 = bp->w1_fm; p.b1 = bp->b1;
    p.h1 = a1;
    p.s_xn = xn; p.s_q = q; p.s_k = k; p.s_vt = vt; p.s_o = o; p.s_z = z;
    p.n_windows = B * (H / 8) * (W / 8); p.H = H; p.W = W; p.shift = bp->shift;
    p.qscale_plain = (float)(1.0 / sqrt(32.0));
    p.qscale = p.qscale_plain * LOG2E;
    p.tbuf = debug_get_tbuf();
#define UF_ABT(TT)                                                                                          \
        switch (C) {                                                                                        \
            case 32: return launch_one<TT, 32, 256, 0, 1>(p, st);                                           \
            case 64: return launch_one<TT, 64, 256, 0, 1>(p, st);                                           \
            case 128: return launch_one<TT, 128, 256, 0, 1>(p, st);                                         \
            case 256:                                                                                       \
                if (p.n_windows <= 256) return launch_one<TT, 256, 512, 0, 1>(p, st);                       \
                return launch_one<TT, 256, 256, 0, 1>(p, st);                                               \
            case 512: return launch_one<TT, 512, 512, 0, 1>(p, st);                                         \
        }
    if (dtype == UF_BF16) {
        UF_ABT(bf16)
    } else if (dtype == UF_F16) {
        UF_ABT(f16)
    }
#undef UF_ABT
    set_error("attn_block (training form): unsupported C=%d for dtype %d (bf16 / f16 operands, C = 32 ... 512)", C, (int)dtype);
    return UF_ERR_UNSUPPORTED;
}

}  // namespace uf
