// The steps either side of the hot path, all HBM-bound streaming kernels:
//   f-2  training-step tail: Charbonnier loss + its gradient in one pass (losses.py:41-52, train/train_denoise.py:164,181),
//        multi-tensor AdamW with decoupled weight decay (train/train_denoise.py:77, torch.optim.AdamW semantics);
//   f-3  evaluation metrics: per-image clamped MSE for myPSNR / batch_PSNR (utils/image_utils.py:40-51) and SSIM with the
//        11x11 Gaussian window (utils/caculate_psnr_ssim.py:35-81);
//   f-1  arbitrary-resolution wrapper: expand2square pad + mask (test/test_sidd.py:79-92) and masked_select crop + clamp
//        (test/test_sidd.py:106-109) as two copy kernels;
//   f-4  input pipeline: random crop + one of the 8 rot90/flip transforms (dataset/dataset_denoise.py:54-70,
//        utils/dataset_utils.py:5-33) from uint8 or f32 frames straight to f32 patches, and MixUp
//        (utils/dataset_utils.py:37-53).
// Every reduction is two-stage through the caller's workspace in a fixed order: bit-reproducible, no atomics.
#include "uf_internal.h"

namespace uf {
namespace {

constexpr int RB = 256;                 // threads per reduction block

__device__ __forceinline__ double block_sum(double v, double* sh) {
    // wave tree on the VALU (64 lanes), then the 4 waves through LDS: a fixed order
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int w = 0; w < RB / 64; ++w) t += sh[w];
    return t;   // valid in thread 0
}

// ---- Charbonnier ------------------------------------------------------------------------------------------------------
// loss = mean(sqrt(d^2 + eps^2)), d = y - target; dy = gscale * d / sqrt(d^2 + eps^2) / n       (losses.py:47-52)
__global__ __launch_bounds__(RB) void charbonnier_kernel(const float* __restrict__ y, const float* __restrict__ tgt, float* __restrict__ dy,
                                                         double* __restrict__ partial, long long n4, long long n, float eps2, float gs) {
    __shared__ double sh[RB / 64];
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * RB;
    for (long long i = (long long)blockIdx.x * RB + threadIdx.x; i < n4; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4*>(y)[i], b = reinterpret_cast<const f32x4*>(tgt)[i];
        f32x4 g;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = a[k] - b[k];
            const float r = sqrtf(d * d + eps2);
            s += r;
            g[k] = gs * d / r;
        }
        acc += (double)s;
        if (dy) reinterpret_cast<f32x4*>(dy)[i] = g;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)       // tail (n % 4 elements), at most 3
        for (long long i = n4 * 4; i < n; ++i) {
            const float d = y[i] - tgt[i];
            const float r = sqrtf(d * d + eps2);
            acc += (double)r;
            if (dy) dy[i] = gs * d / r;
        }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
// one workgroup of RB threads: thread t adds partials t, t + RB, ... in index order, block_sum adds the threads in a fixed order (bit-reproducible).  Round 6: it was
// ONE thread walking all partials -- 124 us of dependent loads between the forward and the backward of every training step (profiles/r06_train_serial_kernel_stats.csv)
__global__ __launch_bounds__(RB) void finalize_mean_kernel(const double* __restrict__ partial, int nb, double inv_n, float* __restrict__ out) {
    __shared__ double sh[RB / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += RB) acc += partial[i];
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = (float)(t * inv_n);
}

// ---- multi-tensor AdamW ---------------------------------------------------------------------------------------------
constexpr int AW_MAX = 40;               // tensors per launch (kernel arguments by value: no device-side table to own)
constexpr int AW_CHUNK = 8192;           // elements per workgroup
struct AdamWArgs {
    float* p[AW_MAX]; const float* g[AW_MAX]; float* m[AW_MAX]; float* v[AW_MAX];
    long long n[AW_MAX];
    int first_chunk[AW_MAX + 1];         // prefix sums of ceil(n / AW_CHUNK)
    int count;
    float decay, omb1, omb2, b2, step, bc2s, eps, gs;   // host doubles rounded once, as torch's Python scalars are: 1 - lr*wd, 1 - b1, 1 - b2, lr / (1 - b1^t), sqrt(1 - b2^t)
    // dynamic loss scale on the device (uf_adamw_step_scaled): state = {scale, 1 / scale, found_inf, growth tracker, good steps}; NULL = plain step.
    // With it the gradient factor is gs * state[1], the update is skipped when state[2] != 0, and the bias corrections use step state[4] + 1.
    const float* scaler;
    double lr, beta1, beta2;
};
// torch.optim.AdamW (decoupled weight decay), single-tensor formulas, f32 state:
//   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2 v + (1-b2) g g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// in torch's operation order (_single_tensor_adamw: mul_, lerp_, mul_.addcmul_, sqrt / bias_correction2_sqrt + eps, addcdiv_)
__global__ __launch_bounds__(256) void adamw_kernel(const AdamWArgs a) {
    int t = 0;
    const int c = blockIdx.x;
#pragma unroll 1
    while (t + 1 < a.count && c >= a.first_chunk[t + 1]) ++t;     // <= 40 scalar steps
    const long long base = (long long)(c - a.first_chunk[t]) * AW_CHUNK;
    const long long n = a.n[t];
    float* __restrict__ p = a.p[t]; const float* __restrict__ g = a.g[t]; float* __restrict__ m = a.m[t]; float* __restrict__ v = a.v[t];
    float gs = a.gs, stepf = a.step, bc2s = a.bc2s;
    if (a.scaler) {                                     // GradScaler semantics (train/train_denoise.py:180-184): unscale, skip on inf / nan
        if (a.scaler[2] != 0.0f) return;                // found_inf: the whole step is skipped (wave-uniform)
        gs *= a.scaler[1];
        const double t = (double)a.scaler[4] + 1.0;     // optimizer steps actually taken so far + 1
        stepf = (float)(a.lr / (1.0 - pow(a.beta1, t)));
        bc2s = (float)sqrt(1.0 - pow(a.beta2, t));
    }
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= gs;
        pp *= a.decay;
        mm = mm + a.omb1 * (gg - mm);
        vv = vv * a.b2 + a.omb2 * (gg * gg);
        pp -= stepf * (mm / (sqrtf(vv) / bc2s + a.eps));
    };
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (vec) {
#pragma unroll 2
        for (int k = threadIdx.x * 4; k < AW_CHUNK; k += 256 * 4) {
            const long long i = base + k;
            if (i + 4 <= n) {
                f32x4 pp = *reinterpret_cast<f32x4*>(p + i), mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
                const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = pp[e], me = mm[e], ve = vv[e];
                    upd(pe, gg[e], me, ve);
                    pp[e] = pe; mm[e] = me; vv[e] = ve;
                }
                *reinterpret_cast<f32x4*>(p + i) = pp; *reinterpret_cast<f32x4*>(m + i) = mm; *reinterpret_cast<f32x4*>(v + i) = vv;
            } else {
                for (long long j = i; j < n && j < i + 4; ++j) upd(p[j], g[j], m[j], v[j]);
            }
        }
    } else {
        for (int k = threadIdx.x; k < AW_CHUNK; k += 256) {
            const long long i = base + k;
            if (i < n) upd(p[i], g[i], m[i], v[i]);
        }
    }
}

// ---- PSNR: per-image sum of squared differences of the clamped images -------------------------------------------------
__global__ __launch_bounds__(RB) void sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ partial,
                                                    long long per_img, int blocks_per_img, int clamp01) {
    __shared__ double sh[RB / 64];
    const int img = blockIdx.x / blocks_per_img, blk = blockIdx.x - img * blocks_per_img;
    const float* pa = a + (size_t)img * per_img; const float* pb = b + (size_t)img * per_img;
    double acc = 0.0;
    for (long long i = (long long)blk * RB + threadIdx.x; i < per_img; i += (long long)blocks_per_img * RB) {
        float x = pa[i], y = pb[i];
        if (clamp01) { x = fminf(fmaxf(x, 0.f), 1.f); y = fminf(fmaxf(y, 0.f), 1.f); }
        const float d = x - y;
        acc += (double)(d * d);
    }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
__global__ void finalize_per_image_kernel(const double* __restrict__ partial, int blocks_per_img, int n_img, double inv, float* __restrict__ out) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= n_img) return;
    double t = 0.0;
    for (int i = 0; i < blocks_per_img; ++i) t += partial[(size_t)img * blocks_per_img + i];
    out[img] = (float)(t * inv);
}

// ---- SSIM (utils/caculate_psnr_ssim.py:35-56): uint8-quantised images in [0,255], 11x11 Gaussian sigma 1.5, "valid" region ----
// one thread per output pixel of the (H-10) x (W-10) map of one (image, channel) plane; a 16x16 output tile reads a 26x26 input
// tile through LDS.  Quantisation follows calculate_ssim: (img * 255).round() as uint8 (the conversion wraps out-of-range values, see below).
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ partial, int H, int W,
                                                   int tiles_x, int tiles_y) {
    __shared__ float sa[26][27], sb[26][27];
    __shared__ double sh[RB / 64];
    __shared__ double c_gauss11[11];     // cv2.getGaussianKernel(11, 1.5): exp(-(i-5)^2 / (2 sigma^2)), normalised
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < 11; ++i) { c_gauss11[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += c_gauss11[i]; }
        for (int i = 0; i < 11; ++i) c_gauss11[i] /= s;
    }
    const int plane = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x - plane * (tiles_x * tiles_y);
    const int ty0 = (tr / tiles_x) * 16, tx0 = (tr % tiles_x) * 16;
    const float* pa = a + (size_t)plane * H * W; const float* pb = b + (size_t)plane * H * W;
    for (int i = threadIdx.x; i < 26 * 26; i += 256) {
        const int yy = i / 26, xx = i - yy * 26;
        const int gy = ty0 + yy, gx = tx0 + xx;
        float va = 0.f, vb = 0.f;
        if (gy < H && gx < W) {
            // (img * 255.0).round().astype(np.uint8), utils/caculate_psnr_ssim.py:59-62: a float -> uint8 conversion, which keeps the low 8 bits of the
            // rounded integer (values outside [0, 1] WRAP, they are not clamped; the reference's callers pass clamped restorations, so it only matters
            // for parity on out-of-range inputs -- VERDICT r03 "weak" 13)
            // A float -> int conversion is undefined for NaN / inf / |v| >= 2^31: those pixels quantise to 0, as the oracle's int64 path
            // gives for non-finite values; everything a restoration can produce is far inside the range.
            auto quant = [](float v) {
                v = rintf(v * 255.0f);
                return (v == v && fabsf(v) < 2147483520.0f) ? (float)((int)v & 255) : 0.0f;
            };
            va = quant(pa[(size_t)gy * W + gx]);
            vb = quant(pb[(size_t)gy * W + gx]);
        }
        sa[yy][xx] = va; sb[yy][xx] = vb;
    }
    __syncthreads();
    const int oy = threadIdx.x >> 4, ox = threadIdx.x & 15;
    double val = 0.0;
    if (ty0 + oy < H - 10 && tx0 + ox < W - 10) {
        double mu1 = 0, mu2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int ky = 0; ky < 11; ++ky) {
            double r1 = 0, r2 = 0, r11 = 0, r22 = 0, r12 = 0;
            for (int kx = 0; kx < 11; ++kx) {
                const double w = c_gauss11[kx], x = sa[oy + ky][ox + kx], y = sb[oy + ky][ox + kx];
                r1 += w * x; r2 += w * y; r11 += w * x * x; r22 += w * y * y; r12 += w * x * y;
            }
            const double w = c_gauss11[ky];
            mu1 += w * r1; mu2 += w * r2; s11 += w * r11; s22 += w * r22; s12 += w * r12;
        }
        const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
        const double m11 = mu1 * mu1, m22 = mu2 * mu2, m12 = mu1 * mu2;
        val = ((2 * m12 + C1) * (2 * (s12 - m12) + C2)) / ((m11 + m22 + C1) * ((s11 - m11) + (s22 - m22) + C2));
    }
    const double t = block_sum(val, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// ---- expand2square / crop + clamp ------------------------------------------------------------------------------------
// canvas (B,C,X,X) = 0 except the image at (y0,x0); mask (B,1,X,X) = 1 over the image        (test/test_sidd.py:79-92)
__global__ __launch_bounds__(256) void pad_canvas_kernel(const float* __restrict__ img, float* __restrict__ canvas, float* __restrict__ mask, int BC, int C, int h, int w,
                                                         int X, int y0, int x0) {
    const long long n = (long long)BC * X * X;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int xx = (int)(i % X), yy = (int)((i / X) % X), pc = (int)(i / ((long long)X * X));
        const bool in = yy >= y0 && yy < y0 + h && xx >= x0 && xx < x0 + w;
        canvas[i] = in ? img[((size_t)pc * h + (yy - y0)) * w + (xx - x0)] : 0.f;
        if (mask && pc % C == 0) mask[((size_t)(pc / C) * X + yy) * X + xx] = in ? 1.f : 0.f;
    }
}
// out (B,C,h,w) = clamp(canvas[:, :, y0:y0+h, x0:x0+w], 0, 1)                               (test/test_sidd.py:108-109)
__global__ __launch_bounds__(256) void crop_clamp_kernel(const float* __restrict__ canvas, float* __restrict__ out, int BC, int h, int w, int X, int y0, int x0, int clamp01) {
    const long long n = (long long)BC * h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int xx = (int)(i % w), yy = (int)((i / w) % h), pc = (int)(i / ((long long)w * h));
        float v = canvas[((size_t)pc * X + y0 + yy) * X + x0 + xx];
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        out[i] = v;
    }
}

// ---- training patches: crop + rot90/flip --------------------------------------------------------------------------------
// out[b][c][i][j] = T_k(src[idx[b]][c][r0:r0+ps][c0:c0+ps])[i][j], k = one of Augment_RGB_torch.transform0..7
// (utils/dataset_utils.py:8-33: rot90 by k&3 quarter turns in dims [-1,-2], then flip(-2) when k >= 4).  With R = rot90(x, k,
// dims=[-1,-2]) (which equals rot90 by -k in the usual [-2,-1] convention):
//   k=0 R[i][j] = x[i][j];  k=1 R[i][j] = x[n-1-j][i];  k=2 R[i][j] = x[n-1-i][n-1-j];  k=3 R[i][j] = x[j][n-1-i];
// and the flip maps i -> n-1-i before that.  src is uint8 (0..255, divided by 255 like load_img, utils/image_utils.py:31-35)
// or f32; layout (N,H,W,3) "hwc" (what cv2 / PIL hand over) or (N,3,H,W).
template <typename S>
__global__ __launch_bounds__(256) void crop_aug_kernel(const S* __restrict__ src, float* __restrict__ out, const int* __restrict__ meta /* [B][4]: idx, r0, c0, k */,
                                                       int N, int B, int H, int W, int ps, int hwc) {
    const long long n = (long long)B * 3 * ps * ps;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        const int j = (int)(t % ps), i = (int)((t / ps) % ps), c = (int)((t / ((long long)ps * ps)) % 3), b = (int)(t / ((long long)3 * ps * ps));
        // device-side metadata is clamped into the frame stack: a bad index or crop origin (user-supplied meta) reads a valid
        // pixel instead of memory outside the allocation (the host checks N, H, W, ps; the entries themselves live on the device)
        int idx = meta[b * 4], r0 = meta[b * 4 + 1], c0 = meta[b * 4 + 2];
        const int k = meta[b * 4 + 3] & 7;
        idx = idx < 0 ? 0 : (idx >= N ? N - 1 : idx);
        r0 = r0 < 0 ? 0 : (r0 > H - ps ? H - ps : r0);
        c0 = c0 < 0 ? 0 : (c0 > W - ps ? W - ps : c0);
        const int ii = (k & 4) ? ps - 1 - i : i;     // flip(-2) is applied last: undo it first
        int sy, sx;
        switch (k & 3) {
            case 0: sy = ii; sx = j; break;
            case 1: sy = ps - 1 - j; sx = ii; break;
            case 2: sy = ps - 1 - ii; sx = ps - 1 - j; break;
            default: sy = j; sx = ps - 1 - ii; break;
        }
        const size_t y = r0 + sy, x = c0 + sx;
        const size_t o = hwc ? (((size_t)idx * H + y) * W + x) * 3 + c : (((size_t)idx * 3 + c) * H + y) * W + x;
        float v;
        if constexpr (sizeof(S) == 1) v = (float)src[o] / 255.0f; else v = (float)src[o];
        out[t] = v;
    }
}
// MixUp (utils/dataset_utils.py:44-53): out[b] = lam[b] * x[b] + (1 - lam[b]) * x[perm[b]]
__global__ __launch_bounds__(256) void mixup_kernel(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ lam, const int* __restrict__ perm,
                                                    int B, long long per) {
    const long long n = (long long)B * per;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        const int b = (int)(t / per);
        const long long r = t - (long long)b * per;
        const float l = lam[b];
        out[t] = l * x[t] + (1.0f - l) * x[(size_t)perm[b] * per + r];
    }
}

int grid_for(long long n, int per_block) {
    long long g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));   // grid-stride beyond 2048 workgroups (8 per CU)
}

}  // namespace
}  // namespace uf

using namespace uf;

extern "C" size_t uf_charbonnier_workspace_bytes(long long n) {
    return (size_t)grid_for((n + 3) / 4, RB) * sizeof(double);
}
extern "C" int uf_charbonnier_fwd_bwd(const float* y, const float* target, float* dy, float* loss, long long n, float eps, float grad_scale,
                                      void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(y && target && loss && ws, UF_ERR_NULL, "uf_charbonnier_fwd_bwd: null pointer");
    UF_REQUIRE(n > 0, UF_ERR_SHAPE, "uf_charbonnier_fwd_bwd: n=%lld", n);
    UF_REQUIRE(((uintptr_t)y % 16) == 0 && ((uintptr_t)target % 16) == 0 && (!dy || ((uintptr_t)dy % 16) == 0) && ((uintptr_t)ws % 8) == 0, UF_ERR_ALIGN,
               "uf_charbonnier_fwd_bwd: y, target, dy must be 16-byte aligned");
    const int nb = grid_for((n + 3) / 4, RB);
    UF_REQUIRE(ws_bytes >= (size_t)nb * sizeof(double), UF_ERR_WORKSPACE, "uf_charbonnier_fwd_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    {
        ScopedTimer tm("charbonnier", 6.0 * n, (dy ? 12.0 : 8.0) * n, st);
        hipLaunchKernelGGL(charbonnier_kernel, dim3(nb), dim3(RB), 0, st, y, target, dy, (double*)ws, n / 4, n, eps * eps, grad_scale / (float)n);
    }
    hipLaunchKernelGGL(finalize_mean_kernel, dim3(1), dim3(RB), 0, st, (const double*)ws, nb, 1.0 / (double)n, loss);
    return check_launch("charbonnier");
}

// found_inf (multi-tensor, 40 tensors per launch): state[2] = 1 if any gradient element is inf or nan (torch's _amp_foreach_non_finite_check_and_unscale_
// without the unscale, which uf_adamw_step_scaled folds into its gradient factor)
struct FoundInfArgs { const float* g[AW_MAX]; long long n[AW_MAX]; int first_chunk[AW_MAX + 1]; int count; float* state; };
__global__ __launch_bounds__(256) void found_inf_kernel(const FoundInfArgs a) {
    int t = 0;
    const int c = blockIdx.x;
#pragma unroll 1
    while (t + 1 < a.count && c >= a.first_chunk[t + 1]) ++t;
    const long long base = (long long)(c - a.first_chunk[t]) * AW_CHUNK, n = a.n[t];
    const float* __restrict__ g = a.g[t];
    bool bad = false;
    for (int k = threadIdx.x; k < AW_CHUNK; k += 256) {
        const long long i = base + k;
        if (i < n) { const float v = g[i]; bad = bad || !(fabsf(v) <= 3.4028234e38f); }      // false for inf and for nan
    }
    if (__syncthreads_or(bad ? 1 : 0) && threadIdx.x == 0) a.state[2] = 1.0f;                 // racing writers store the same value
}
// GradScaler.update(): found_inf -> scale *= backoff, tracker = 0; else ++good steps, ++tracker, scale *= growth every `interval` clean steps
__global__ void scaler_update_kernel(float* state, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float scale = state[0], tracker = state[3];
    if (state[2] != 0.0f) { scale *= backoff; tracker = 0.0f; }
    else {
        state[4] += 1.0f;
        tracker += 1.0f;
        if (tracker >= (float)interval) { scale *= growth; tracker = 0.0f; }
    }
    state[0] = scale; state[1] = 1.0f / scale; state[2] = 0.0f; state[3] = tracker;
}

static int adamw_any(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const long long* numel,
                     int n_tensors, double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale, const float* scaler, void* stream) {
    UF_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel, UF_ERR_NULL, "uf_adamw_step: null pointer");
    UF_REQUIRE(n_tensors >= 0 && step >= 1, UF_ERR_SHAPE, "uf_adamw_step: n_tensors=%d step=%d (step counts from 1)", n_tensors, step);
    hipStream_t st = (hipStream_t)stream;
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    int i = 0;
    while (i < n_tensors) {
        AdamWArgs a{};
        a.decay = (float)(1.0 - lr * weight_decay); a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
        a.b2 = (float)beta2; a.step = (float)(lr / bc1); a.bc2s = (float)sqrt(bc2); a.eps = (float)eps; a.gs = (float)grad_scale;
        a.scaler = scaler; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2;
        int c = 0, chunks = 0;
        double elems = 0;
        for (; i < n_tensors && c < AW_MAX; ++i) {
            if (numel[i] <= 0) continue;
            UF_REQUIRE(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i], UF_ERR_NULL, "uf_adamw_step: tensor %d has a null pointer", i);
            const long long nch = (numel[i] + AW_CHUNK - 1) / AW_CHUNK;
            UF_REQUIRE(chunks + nch < 0x7fffffffLL, UF_ERR_SHAPE, "uf_adamw_step: too many elements in one launch");
            a.p[c] = params[i]; a.g[c] = grads[i]; a.m[c] = exp_avg[i]; a.v[c] = exp_avg_sq[i]; a.n[c] = numel[i];
            a.first_chunk[c] = chunks;
            chunks += (int)nch; elems += (double)numel[i];
            ++c;
        }
        if (c == 0) break;
        a.first_chunk[c] = chunks; a.count = c;
        {
            ScopedTimer tm("adamw", 12.0 * elems, 28.0 * elems, st);
            hipLaunchKernelGGL(adamw_kernel, dim3(chunks), dim3(256), 0, st, a);
        }
        if (int rc = check_launch("adamw")) return rc;
    }
    return UF_OK;
}

extern "C" int uf_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const long long* numel,
                             int n_tensors, double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream) {
    return adamw_any(params, grads, exp_avg, exp_avg_sq, numel, n_tensors, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, stream);
}

extern "C" int uf_adamw_step_scaled(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const long long* numel,
                                    int n_tensors, double lr, double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                                    const float* scaler_state, void* stream) {
    UF_REQUIRE(scaler_state, UF_ERR_NULL, "uf_adamw_step_scaled: null scaler state");
    return adamw_any(params, grads, exp_avg, exp_avg_sq, numel, n_tensors, lr, beta1, beta2, eps, weight_decay, 1, grad_scale, scaler_state, stream);
}

extern "C" int uf_grad_scaler_check(const float* const* grads, const long long* numel, int n_tensors, float* scaler_state, void* stream) {
    UF_REQUIRE(grads && numel && scaler_state, UF_ERR_NULL, "uf_grad_scaler_check: null pointer");
    UF_REQUIRE(n_tensors >= 0, UF_ERR_SHAPE, "uf_grad_scaler_check: n_tensors=%d", n_tensors);
    hipStream_t st = (hipStream_t)stream;
    int i = 0;
    while (i < n_tensors) {
        FoundInfArgs a{};
        a.state = scaler_state;
        int c = 0, chunks = 0;
        for (; i < n_tensors && c < AW_MAX; ++i) {
            if (numel[i] <= 0) continue;
            UF_REQUIRE(grads[i], UF_ERR_NULL, "uf_grad_scaler_check: tensor %d is null", i);
            const long long nch = (numel[i] + AW_CHUNK - 1) / AW_CHUNK;
            UF_REQUIRE(chunks + nch < 0x7fffffffLL, UF_ERR_SHAPE, "uf_grad_scaler_check: too many elements in one launch");
            a.g[c] = grads[i]; a.n[c] = numel[i]; a.first_chunk[c] = chunks;
            chunks += (int)nch;
            ++c;
        }
        if (c == 0) break;
        a.first_chunk[c] = chunks; a.count = c;
        hipLaunchKernelGGL(found_inf_kernel, dim3(chunks), dim3(256), 0, st, a);
        if (int rc = check_launch("grad_scaler_check")) return rc;
    }
    return UF_OK;
}

extern "C" int uf_grad_scaler_update(float* scaler_state, double growth_factor, double backoff_factor, int growth_interval, void* stream) {
    UF_REQUIRE(scaler_state, UF_ERR_NULL, "uf_grad_scaler_update: null pointer");
    UF_REQUIRE(growth_factor > 1.0 && backoff_factor > 0.0 && backoff_factor < 1.0 && growth_interval >= 1, UF_ERR_SHAPE,
               "uf_grad_scaler_update: growth %g (> 1), backoff %g (in (0, 1)), interval %d (>= 1)", growth_factor, backoff_factor, growth_interval);
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler_state, (float)growth_factor, (float)backoff_factor, growth_interval);
    return check_launch("grad_scaler_update");
}

extern "C" size_t uf_image_metric_workspace_bytes(int n_images, int C, int H, int W) {
    const long long per = (long long)C * H * W;
    const int bpi = grid_for(per, RB * 16);
    const long long ssim_blocks = (long long)n_images * C * ((H - 10 + 15) / 16 > 0 ? (H - 10 + 15) / 16 : 1) * ((W - 10 + 15) / 16 > 0 ? (W - 10 + 15) / 16 : 1);
    const long long nb = (long long)n_images * bpi > ssim_blocks ? (long long)n_images * bpi : ssim_blocks;
    return (size_t)nb * sizeof(double);
}
extern "C" int uf_batch_mse(const float* a, const float* b, float* mse_per_image, int n_images, int C, int H, int W, int clamp01, void* ws, size_t ws_bytes,
                            void* stream) {
    UF_REQUIRE(a && b && mse_per_image && ws, UF_ERR_NULL, "uf_batch_mse: null pointer");
    UF_REQUIRE(n_images > 0 && C > 0 && H > 0 && W > 0, UF_ERR_SHAPE, "uf_batch_mse: shape");
    const long long per = (long long)C * H * W;
    const int bpi = grid_for(per, RB * 16);
    UF_REQUIRE(ws_bytes >= (size_t)n_images * bpi * sizeof(double) && ((uintptr_t)ws % 8) == 0, UF_ERR_WORKSPACE, "uf_batch_mse: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    {
        ScopedTimer tm("batch_mse", 3.0 * per * n_images, 8.0 * per * n_images, st);
        hipLaunchKernelGGL(sqdiff_kernel, dim3(n_images * bpi), dim3(RB), 0, st, a, b, (double*)ws, per, bpi, clamp01);
    }
    hipLaunchKernelGGL(finalize_per_image_kernel, dim3((n_images + 63) / 64), dim3(64), 0, st, (const double*)ws, bpi, n_images, 1.0 / (double)per, mse_per_image);
    return check_launch("batch_mse");
}
extern "C" int uf_batch_ssim(const float* a, const float* b, float* ssim_per_image, int n_images, int C, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    UF_REQUIRE(a && b && ssim_per_image && ws, UF_ERR_NULL, "uf_batch_ssim: null pointer");
    UF_REQUIRE(n_images > 0 && C > 0 && H > 10 && W > 10, UF_ERR_SHAPE, "uf_batch_ssim: needs H, W > 10 (11x11 window, valid region)");
    const int tx = (W - 10 + 15) / 16, ty = (H - 10 + 15) / 16;
    const long long nb = (long long)n_images * C * tx * ty;
    UF_REQUIRE(nb < 0x7fffffffLL && ws_bytes >= (size_t)nb * sizeof(double) && ((uintptr_t)ws % 8) == 0, UF_ERR_WORKSPACE, "uf_batch_ssim: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    {
        ScopedTimer tm("batch_ssim", 1250.0 * n_images * C * H * W, 8.0 * n_images * C * H * W, st);
        hipLaunchKernelGGL(ssim_kernel, dim3((unsigned)nb), dim3(256), 0, st, a, b, (double*)ws, H, W, tx, ty);
    }
    hipLaunchKernelGGL(finalize_per_image_kernel, dim3((n_images + 63) / 64), dim3(64), 0, st, (const double*)ws, C * tx * ty, n_images,
                       1.0 / ((double)C * (H - 10) * (W - 10)), ssim_per_image);
    return check_launch("batch_ssim");
}

extern "C" int uf_expand2square(const float* img, float* canvas, float* mask, int B, int C, int h, int w, int X, void* stream) {
    UF_REQUIRE(img && canvas, UF_ERR_NULL, "uf_expand2square: null pointer");
    UF_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && X >= h && X >= w, UF_ERR_SHAPE, "uf_expand2square: B=%d C=%d h=%d w=%d X=%d", B, C, h, w, X);
    const long long n = (long long)B * C * X * X;
    hipStream_t st = (hipStream_t)stream;
    {
        ScopedTimer tm("expand2square", 0.0, 4.0 * n + 4.0 * B * C * h * w, st);
        hipLaunchKernelGGL(pad_canvas_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, img, canvas, mask, B * C, C, h, w, X, (X - h) / 2, (X - w) / 2);
    }
    return check_launch("expand2square");
}
extern "C" int uf_crop_clamp(const float* canvas, float* out, int B, int C, int h, int w, int X, int clamp01, void* stream) {
    UF_REQUIRE(canvas && out, UF_ERR_NULL, "uf_crop_clamp: null pointer");
    UF_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && X >= h && X >= w, UF_ERR_SHAPE, "uf_crop_clamp: B=%d C=%d h=%d w=%d X=%d", B, C, h, w, X);
    const long long n = (long long)B * C * h * w;
    hipStream_t st = (hipStream_t)stream;
    {
        ScopedTimer tm("crop_clamp", 0.0, 8.0 * n, st);
        hipLaunchKernelGGL(crop_clamp_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, canvas, out, B * C, h, w, X, (X - h) / 2, (X - w) / 2, clamp01);
    }
    return check_launch("crop_clamp");
}

extern "C" int uf_crop_augment(const void* src, int src_is_u8, int src_hwc, float* out, const int* meta, int B, int N, int H, int W, int ps, void* stream) {
    UF_REQUIRE(src && out && meta, UF_ERR_NULL, "uf_crop_augment: null pointer");
    UF_REQUIRE(B > 0 && N > 0 && ps > 0 && ps <= H && ps <= W, UF_ERR_SHAPE, "uf_crop_augment: B=%d N=%d H=%d W=%d ps=%d", B, N, H, W, ps);
    const long long n = (long long)B * 3 * ps * ps;
    hipStream_t st = (hipStream_t)stream;
    {
        ScopedTimer tm("crop_augment", 0.0, (src_is_u8 ? 5.0 : 8.0) * n, st);
        if (src_is_u8) hipLaunchKernelGGL(crop_aug_kernel<unsigned char>, dim3(grid_for(n, 1024)), dim3(256), 0, st, (const unsigned char*)src, out, meta, N, B, H, W, ps, src_hwc);
        else hipLaunchKernelGGL(crop_aug_kernel<float>, dim3(grid_for(n, 1024)), dim3(256), 0, st, (const float*)src, out, meta, N, B, H, W, ps, src_hwc);
    }
    return check_launch("crop_augment");
}
extern "C" int uf_mixup(const float* x, float* out, const float* lam, const int* perm, int B, long long per_sample, void* stream) {
    UF_REQUIRE(x && out && lam && perm && x != out, UF_ERR_NULL, "uf_mixup: null pointer (out must not alias x: samples read their partner)");
    UF_REQUIRE(B > 0 && per_sample > 0, UF_ERR_SHAPE, "uf_mixup: shape");
    const long long n = (long long)B * per_sample;
    hipStream_t st = (hipStream_t)stream;
    {
        ScopedTimer tm("mixup", 3.0 * n, 12.0 * n, st);
        hipLaunchKernelGGL(mixup_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, x, out, lam, perm, B, per_sample);
    }
    return check_launch("mixup");
}
