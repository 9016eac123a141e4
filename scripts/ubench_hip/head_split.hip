// OutputProj (conv3x3, 64 -> 3 channels, + global residual; reference model.py:828-836, :1305) on the matrix pipe with SPLIT operands -- round 6 prototype,
// measured inside the library and removed again (profiles/r06_run31_head.txt, r06_run31_ab.txt).  Kept as a record of the kernel; it is not built.
//
// The f32 forms of the library (uf_elementwise.hip: output_proj_kernel / output_proj2_kernel) are bound by their own instruction stream: 432 packed FMAs and 18
// LDS vector reads per thread and strip, 139-143 us at 16 x 256 x 256 against a 49 us HBM floor.  Here D[n][pixel] = sum_k W[n][k] X[pixel][k], k = (tap, channel)
// (K = 576 = 18 k-steps of 32), n = the 3 output channels padded to the 16 rows of a 16x16x32 MFMA, and BOTH operands carried as hi + lo parts of the operand
// type (hi = T(v), lo = T(v - hi)): three MFMAs per k-step (hi hi, hi lo, lo hi).  Products of 2-byte operands are exact in the f32 accumulator, so the result
// keeps 16 (bf16) / 22 (f16) bits of every operand: measured 2.6e-5 / 3.8e-6 max-abs against the f32 form.
// A workgroup stages the 6 x 34-pixel halo tile of a 4 x 32-pixel output tile as two T tiles (pixel pitch 160 bytes: 16 neighbouring pixels of a fragment read hit
// 16 different bank groups), wave w owns output row w, the 36 weight fragments are built once per workgroup and live in registers, workgroups walk the tiles.
//
// Result: 120 us against 139 us (batch 16), 229 against 284 (batch 32) -- the tile staging (6.4 dependent rounds of two 32-byte loads per thread, two tiles per
// CU in flight) now bounds it -- and NO difference in the bench line (2519 / 2503 / 2512 against 2540 / 2503 / 2483 img/s); it also made the fused forward differ
// from the block-by-block forward (which runs the f32 head) by 2.6e-5, which tests/test_gpu_model.py::test_checkpoint_forms_and_blockwise_path rightly rejects.
//
// template <typename T>
// __global__ __launch_bounds__(256, 2) void output_proj_mfma_kernel(const float* x, int ld_x, const float* w /* [3][9][64] */, const float* bias, const float* img,
//                                                                   float* out, int B, int H, int W, int add_img, int tiles_x, int tiles_y, int n_tiles) {
//     constexpr int C2 = 64, TWP = 32, PW = TWP + 2, PH = 6, PP = 160, TILE_B = PH * PW * PP, KSN = 18;
//     extern __shared__ char smem[];  char* Hi = smem; char* Lo = smem + TILE_B;
//     const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
//     Frag<T> wh[KSN], wl[KSN];                                   // A operand: row n = fr (rows 3..15 zero), k slots 8 fg ..
//     for (int ks = 0; ks < KSN; ++ks) {
//         const int tap = ks >> 1, ch = (ks & 1) * 32 + fg * 8;
//         float v[8], r[8];  load8(v, w + ((fr < 3 ? fr : 0) * 9 + tap) * C2 + ch);  if (fr >= 3) zero8(v);
//         wh[ks].v = pack8<T>(v);  unpack8<T>(wh[ks].v, r);  for (e) r[e] = v[e] - r[e];  wl[ks].v = pack8<T>(r);
//     }
//     for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
//         const int bt = xcd_tile(t, n_tiles), b = bt / (tiles_x * tiles_y), tr = bt % (tiles_x * tiles_y), y0 = (tr / tiles_x) * 4, x0 = (tr % tiles_x) * TWP;
//         __syncthreads();
//         for (int q = tid; q < PH * PW * 8; q += 256) {          // 8 channels of one halo pixel: f32 -> hi, lo
//             const int pix = q >> 3, cc = q & 7, pr = pix / PW, pc = pix % PW, iy = y0 - 1 + pr, ix = x0 - 1 + pc;
//             float v[8], r[8];  load8(v, x + ((b * H + clamp(iy)) * W + clamp(ix)) * ld_x + cc * 8);  if (outside) zero8(v);
//             const u32x4 h = pack8<T>(v);  unpack8<T>(h, r);  for (e) r[e] = v[e] - r[e];
//             store16(Hi + pix * PP + cc * 16, h);  store16(Lo + pix * PP + cc * 16, pack8<T>(r));
//         }
//         __syncthreads();
//         f32x4 acc[2] = {};
//         const int lbase = (wave * PW + fr) * PP + fg * 16;
//         for (int ks = 0; ks < KSN; ++ks) {
//             const int tap = ks >> 1, ky = tap / 3, kx = tap % 3, off = lbase + (ky * PW + kx) * PP + (ks & 1) * 64;
//             for (int j = 0; j < 2; ++j) {
//                 Frag<T> ah, al;  load_frag(ah, Hi + off + j * 16 * PP);  load_frag(al, Lo + off + j * 16 * PP);
//                 mma16(acc[j], wh[ks], ah);  mma16(acc[j], wh[ks], al);  mma16(acc[j], wl[ks], ah);
//             }
//         }
//         if (fg == 0 && y0 + wave < H)                            // lane group 0 holds rows n = 0..3 of its pixel
//             for (int j = 0; j < 2; ++j) if (x0 + 16 * j + fr < W) for (int c = 0; c < 3; ++c) out[..] = acc[j][c] + bias[c] (+ img[..]);
//     }
// }
