// The first convolution of the networks: Conv2d(Cin in {1..3}, Cout, 3, padding=1) on the fp32 NCHW image
// (models/unetvgg.py:72 Conv2d(3,64); models/unetplusplus.py:45 conv0_0.conv1 = Conv2d(1,64)), straight from the image.
//
//   forward :  y[p][co] = bias[co] + sum_{k} patch[p][k] * w[co][k],   k = (ky*3+kx)*Cin + c < 9*Cin <= 32
//   weights :  dw[co][k] = sum_p dy[p][co] * patch[p][k]
//
// K = 9*Cin <= 27 fits ONE 32-deep MFMA step, so the layer is bound by the 64-channel activation it writes (forward) or
// reads (weight gradient), not by arithmetic.  Neither kernel materialises the patch image: the 3-channel halo tile of the
// image is staged in LDS as fp32 and every lane gathers the eight patch values of its MFMA operand from it.
#include "common.h"

namespace segk {

constexpr int kFirstThreads = 256;
constexpr int kFTH = 8, kFTW = 64;                 // workgroup tile: 8 rows x 64 columns, one 8 x 16 sub-tile per wavefront
constexpr int kFLW = kFTW + 2 + 2;                 // LDS row stride in floats (66 used)
constexpr int kFMaxCin = 3;

// row r of a 32-row group of the MFMA output holds channel 8*((r&15)>>2) + 4*(r>>4) + (r&3): a lane then owns eight
// consecutive channels across two 16-row tiles and stores 16 bytes per pixel
__host__ __device__ inline int first_row_channel(int row) { return (row & ~31) + 8 * ((row & 15) >> 2) + 4 * ((row >> 4) & 1) + (row & 3); }

__device__ __forceinline__ float first_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

struct FirstArgs {
    const float* x;        // [N, Cin, H, W] fp32
    const float* w;        // [Cout, Cin, 3, 3] fp32
    const float* bias;     // [Cout] or null
    bf16* y;               // NHWC, ld elements per pixel
    int64_t ldy;
    float* stat_part;      // optional [N * tiles * 4][2][Cout]
    int N, Cin, H, W, Cout, tiles_x, tiles_y, relu;
    // recomputing modes (the convolution is followed by BatchNorm + ReLU and its output is never stored):
    const float* nstats = nullptr;   // [Cout][2] mean, 1/std
    const float* gamma = nullptr;
    const float* beta = nullptr;
    const float* red = nullptr;      // [Cout][2] the norm backward's two sums
    float inv_m = 0.f;               // 1 / (N*H*W)
    const bf16* g = nullptr;         // gradient of relu(norm(conv(x))), NHWC
    int64_t ldg = 0;
};

// What the kernel does with the convolution's (fp32, never rounded) output z = conv(x) + bias:
//   kStore      store z (and, optionally, the per-wave statistics of the stored values)
//   kStatsOnly  per-wave sum and sum of squares of z, nothing stored
//   kNormRelu   store relu(norm(z))
//   kBwdSums    with g = dL/d relu(norm(z)): per-wave sums of g*[pre > 0] and g*[pre > 0]*zhat
//   kBwdDz      store dL/dz = gamma*rstd*(g*[pre > 0] - s1/M - zhat*s2/M)
// The last four recompute z from the image (a few MFMAs per tile) instead of reading a stored copy: the image is
// 1-3 channels, z is 64.
enum FirstMode { kStore = 0, kStatsOnly = 1, kNormRelu = 2, kBwdSums = 3, kBwdDz = 4 };

// NCT = Cout / 16 output-channel tiles (2 or 4); CIN input channels.  A workgroup walks kTilesPerWg vertically adjacent
// 8 x 64 tiles: the weight fragments and the per-lane gather offsets are set up once, and the next tile's image values
// are fetched into registers while the current tile is multiplied and stored.
constexpr int kTilesPerWg = 6;

template <int NCT, int CIN, int MODE = kStore>
__global__ __launch_bounds__(kFirstThreads, 2) void conv_first_fwd_kernel(const FirstArgs a) {
    constexpr int PER_C = (kFTH + 2) * (kFTW + 2), NVAL = CIN * PER_C;
    constexpr int NLD = (NVAL + kFirstThreads - 1) / kFirstThreads;       // staged values per thread
    constexpr int kZeroWord = kFMaxCin * (kFTH + 2) * kFLW;
    constexpr int KMAX = 9 * CIN;
    __shared__ float img[kFMaxCin * (kFTH + 2) * kFLW + 4];     // + one zero word that padded K indices read
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int ygroups = (a.tiles_y + kTilesPerWg - 1) / kTilesPerWg;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int yg = b % ygroups;
    const int n = b / ygroups;
    const int x0 = tx * kFTW;
    const int ty_lo = yg * kTilesPerWg, ty_hi = min(ty_lo + kTilesPerWg, a.tiles_y);

    // this thread's share of a halo tile, value i: channel << 20 | halo row << 12 | halo column (or -1: no value)
    int src[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int v = i * kFirstThreads + tid;
        const int c = v / PER_C, r = (v - c * PER_C) / (kFTW + 2), xx = v - c * PER_C - r * (kFTW + 2);
        src[i] = v < NVAL ? ((c << 20) | (r << 12) | xx) : -1;
    }
    auto fetch = [&](int ty, float (&val)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = src[i] >> 20, r = (src[i] >> 12) & 255, xx = src[i] & 4095;
            const int gy = ty * kFTH - 1 + r, gx = x0 - 1 + xx;
            const bool ok = src[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            val[i] = ok ? a.x[(((int64_t)n * CIN + c) * a.H + gy) * a.W + gx] : 0.f;
        }
    };
    auto deposit = [&](const float (&val)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = src[i] >> 20, r = (src[i] >> 12) & 255, xx = src[i] & 4095;
            if (src[i] >= 0) img[(c * (kFTH + 2) + r) * kFLW + xx] = val[i];
        }
    };

    float cur[NLD];
    fetch(ty_lo, cur);

    // ---- weight fragments (A operand: rows = output channels in MFMA order, K = patch index), once per workgroup ----
    bf16x8 wa[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
        const int co = first_row_channel(t * 16 + l15);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * g + e;
            float v = 0.f;
            if (k < KMAX) {
                const int tap = k / CIN, c = k - tap * CIN;
                v = a.w[((int64_t)co * CIN + c) * 9 + tap];
            }
            wa[t][e] = (bf16)v;
        }
    }
    // ---- per-lane LDS word offsets of the eight patch values (K = 8g .. 8g+7) of pixel (row 0, column 16*wave + l15) ----
    int off[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * g + e;
        if (k < KMAX) {
            const int tap = k / CIN, c = k - tap * CIN;
            off[e] = (c * (kFTH + 2) + tap / 3) * kFLW + tap % 3 + 16 * wave + l15;
        } else {
            off[e] = kZeroWord;
        }
    }
    float bv[NCT / 2][8];
#pragma unroll
    for (int k2 = 0; k2 < NCT / 2; ++k2) {
#pragma unroll
        for (int r = 0; r < 8; ++r) bv[k2][r] = 0.f;
        if (a.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.bias + 32 * k2 + 8 * g), b1 = *reinterpret_cast<const float4*>(a.bias + 32 * k2 + 8 * g + 4);
            bv[k2][0] = b0.x; bv[k2][1] = b0.y; bv[k2][2] = b0.z; bv[k2][3] = b0.w; bv[k2][4] = b1.x; bv[k2][5] = b1.y; bv[k2][6] = b1.z; bv[k2][7] = b1.w;
        }
    }
    if (tid == 0) img[kZeroWord] = 0.f;
    // per-channel constants of the norm for this lane's channels, written exactly as the streaming norm kernels write them
    constexpr bool kNorm = MODE >= kNormRelu;
    constexpr bool kBwd = MODE == kBwdSums || MODE == kBwdDz;
    // The backward modes keep as few constants live as the arithmetic allows (the registers go to the gradient vectors that
    // are requested ahead of the matrix steps): the sums are taken of g and g*z and turned into sum g*zhat when the partial
    // row is written; dz = sc*(g - m1 - zhat*m2) is evaluated as sc*g - n_a - n_b*z with n_b = sc*m2*rstd, n_a = sc*m1 - n_b*mean.
    float n_sc[kNorm ? NCT / 2 : 1][8], n_sh[kNorm ? NCT / 2 : 1][8];
    float n_a[MODE == kBwdDz ? NCT / 2 : 1][8], n_b[MODE == kBwdDz ? NCT / 2 : 1][8];
    if constexpr (kNorm) {
#pragma unroll
        for (int k2 = 0; k2 < NCT / 2; ++k2)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int c = 32 * k2 + 8 * g + r;
                const float mean = a.nstats[2 * c], rstd = a.nstats[2 * c + 1];
                n_sc[k2][r] = rstd * (a.gamma ? a.gamma[c] : 1.f);
                n_sh[k2][r] = (a.beta ? a.beta[c] : 0.f) - mean * n_sc[k2][r];
                if constexpr (MODE == kBwdDz) {
                    n_b[k2][r] = n_sc[k2][r] * (a.red[2 * c + 1] * a.inv_m) * rstd;
                    n_a[k2][r] = n_sc[k2][r] * (a.red[2 * c] * a.inv_m) - n_b[k2][r] * mean;
                }
            }
    }

    const int xx = x0 + 16 * wave + l15;
    for (int ty = ty_lo; ty < ty_hi; ++ty) {
        __syncthreads();                       // everyone is done reading the previous tile
        deposit(cur);
        __syncthreads();
        if (ty + 1 < ty_hi) fetch(ty + 1, cur);   // in flight behind this tile's arithmetic and stores

        // ---- two half-tiles of four rows each: MFMAs, then bias, ReLU, 16-byte stores and the per-wave channel
        //      statistics of what was stored (kept in registers across the halves) ----
        const int y0 = ty * kFTH;
        float ssum[NCT / 2][8], ssq[NCT / 2][8];
#pragma unroll
        for (int k2 = 0; k2 < NCT / 2; ++k2)
#pragma unroll
            for (int r = 0; r < 8; ++r) ssum[k2][r] = ssq[k2][r] = 0.f;
        // rows per pass: four, or two where the backward modes' per-channel constants need the registers
        constexpr int RP = kBwd ? 2 : kFTH / 2;
#pragma unroll 1
        for (int half = 0; half < kFTH / RP; ++half) {
            f32x4 acc[NCT][RP];
#pragma unroll
            for (int t = 0; t < NCT; ++t)
#pragma unroll
                for (int j = 0; j < RP; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // backward modes: this pass's gradient vectors are requested before the gathers and matrix steps that they wait behind
            struct alignas(16) G8 { bf16 e[8]; };
            G8 gq[kBwd ? RP : 1][kBwd ? NCT / 2 : 1];
            if constexpr (kBwd) {
#pragma unroll
                for (int j = 0; j < RP; ++j) {
                    const int yy = ty * kFTH + half * RP + j;
                    if (yy >= a.H || xx >= a.W) continue;
                    const bf16* gp = a.g + (((int64_t)n * a.H + yy) * a.W + xx) * a.ldg + 8 * g;
#pragma unroll
                    for (int k2 = 0; k2 < NCT / 2; ++k2) gq[j][k2] = *reinterpret_cast<const G8*>(gp + 32 * k2);
                }
            }
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int row = half * RP + j;
                bf16x8 pb;
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[e] = (bf16)img[off[e] + (off[e] == kZeroWord ? 0 : row * kFLW)];
#pragma unroll
                for (int t = 0; t < NCT; ++t) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[t], pb, acc[t][j], 0, 0, 0);
            }
            // row by row, both 32-channel halves of a pixel back to back: the 128 bytes of a pixel leave together
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int yy = y0 + half * RP + j;
                if (yy >= a.H || xx >= a.W) continue;
                const int64_t pix = ((int64_t)n * a.H + yy) * a.W + xx;
                bf16* dst = a.y + pix * a.ldy + 8 * g;
#pragma unroll
                for (int k2 = 0; k2 < NCT / 2; ++k2) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = acc[2 * k2][j][e] + bv[k2][e]; v[4 + e] = acc[2 * k2 + 1][j][e] + bv[k2][4 + e]; }
                    if constexpr (MODE == kStore) {
                        if (a.relu) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
                        }
                        Elem<bf16>::store8(dst + 32 * k2, v);
                        if (a.stat_part) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                const float q = (float)(bf16)v[r];
                                ssum[k2][r] += q;
                                ssq[k2][r] += q * q;
                            }
                        }
                    } else if constexpr (MODE == kStatsOnly) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) { ssum[k2][r] += v[r]; ssq[k2][r] += v[r] * v[r]; }
                    } else if constexpr (MODE == kNormRelu) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r] * n_sc[k2][r] + n_sh[k2][r], 0.f);
                        Elem<bf16>::store8(dst + 32 * k2, v);
                    } else {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const float pre = v[r] * n_sc[k2][r] + n_sh[k2][r];
                            const float gg = pre > 0.f ? (float)gq[j][k2].e[r] : 0.f;
                            if constexpr (MODE == kBwdSums) { ssum[k2][r] += gg; ssq[k2][r] += gg * v[r]; }
                            else v[r] = n_sc[k2][r] * gg - n_a[k2][r] - n_b[k2][r] * v[r];
                        }
                        if constexpr (MODE == kBwdDz) Elem<bf16>::store8(dst + 32 * k2, v);
                    }
                }
            }
        }
        if (a.stat_part && (MODE == kStore || MODE == kStatsOnly || MODE == kBwdSums)) {
#pragma unroll
            for (int k2 = 0; k2 < NCT / 2; ++k2) {
                const int co = 32 * k2 + 8 * g;
#pragma unroll
                for (int r = 0; r < 8; ++r) { ssum[k2][r] = first_row16_sum(ssum[k2][r]); ssq[k2][r] = first_row16_sum(ssq[k2][r]); }
                if constexpr (MODE == kBwdSums) {      // (sum g, sum g*z) -> (sum g, sum g*zhat)
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float mean = a.nstats[2 * (co + r)], rstd = a.nstats[2 * (co + r) + 1];
                        ssq[k2][r] = rstd * (ssq[k2][r] - mean * ssum[k2][r]);
                    }
                }
                if (l15 == 0) {
                    const int64_t row = (((int64_t)n * a.tiles_y + ty) * a.tiles_x + tx) * 4 + wave;
                    float* dst = a.stat_part + row * 2 * a.Cout + co;
                    *reinterpret_cast<float4*>(dst) = make_float4(ssum[k2][0], ssum[k2][1], ssum[k2][2], ssum[k2][3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(ssum[k2][4], ssum[k2][5], ssum[k2][6], ssum[k2][7]);
                    *reinterpret_cast<float4*>(dst + a.Cout) = make_float4(ssq[k2][0], ssq[k2][1], ssq[k2][2], ssq[k2][3]);
                    *reinterpret_cast<float4*>(dst + a.Cout + 4) = make_float4(ssq[k2][4], ssq[k2][5], ssq[k2][6], ssq[k2][7]);
                }
            }
        }
    }
}

template <int NCT, int MODE = kStore>
static void launch_first(const FirstArgs& a, unsigned grid, hipStream_t st) {
    if (a.Cin == 1) hipLaunchKernelGGL((conv_first_fwd_kernel<NCT, 1, MODE>), dim3(grid), dim3(kFirstThreads), 0, st, a);
    else if (a.Cin == 2) hipLaunchKernelGGL((conv_first_fwd_kernel<NCT, 2, MODE>), dim3(grid), dim3(kFirstThreads), 0, st, a);
    else hipLaunchKernelGGL((conv_first_fwd_kernel<NCT, 3, MODE>), dim3(grid), dim3(kFirstThreads), 0, st, a);
}

// ------------------------------------------------------------------------------------------------
// weight gradient of the first convolution: dw[co][k] = sum_p dy[p][co] * patch[p][k]  (K = pixels)
// ------------------------------------------------------------------------------------------------
// A workgroup streams 8 x 32-pixel tiles of dy (64 channels = one 128-byte row per pixel) through two LDS stages by
// DMA; both MFMA operands are needed pixel-major: dy through the transposing 8-byte LDS read, the patch values gathered
// from the fp32 halo tile of the image.  A K-step is 32 pixels = two rows of 16; K index 8g + 4h + q <-> pixel (row h,
// column 4g + q) on both operands.  Wave (strip, half) owns columns 16*strip .. +15 and rows 4*half .. +3 of the tile;
// all four waves accumulate the whole [32 k][64 co] block, summed through LDS at the end into one partial slab per
// workgroup; a second kernel adds the slabs in fixed order (deterministic) and writes OIHW.
constexpr int kGTW = 32, kGTH = 8;                              // dy tile of the weight-gradient kernel
constexpr int kGLW = kGTW + 2 + 2;                              // LDS row stride of the image halo tile (floats)
constexpr int kGTileBytes = kGTW * kGTH * 128;                  // 32 KB of dy per tile

struct FirstWgradArgs {
    const float* x;        // [N, Cin, H, W]
    const bf16* dy;        // NHWC, 64 channels used, ld elements per pixel
    int64_t ld;
    float* slab;           // [workgroups][64][32]
    int N, Cin, H, W, tiles_x, tiles_y, ntiles;
    // recomputing modes: dy is g = dL/d relu(norm(z)) with z = conv(x, w) + bias recomputed per tile
    const float* w = nullptr;        // [64, Cin, 3, 3]
    const float* bias = nullptr;
    const float* nstats = nullptr;   // [64][2] mean, 1/std
    const float* gamma = nullptr;
    const float* beta = nullptr;
    const float* red = nullptr;      // [64][2] the norm backward's two sums (kWgDz)
    float inv_m = 0.f;
    float* part = nullptr;           // kWgSums: [workgroups * 4][2][64] partial rows of (sum g', sum g' * zhat)
};

// kWgPlain: dy is the gradient of the convolution's output.  kWgSums / kWgDz: dy is the gradient g of relu(norm(z)); the
// workgroup recomputes z for its tile from the image halo it holds anyway (16 matrix steps per wave) and either reduces
// the norm backward's two sums over its tiles (kWgSums, no weight gradient) or turns g into dL/dz IN the LDS stage,
// where the weight gradient's transposing reads pick it up (kWgDz): dL/dz never exists in HBM.
enum FirstWgMode { kWgPlain = 0, kWgSums = 1, kWgDz = 2 };

__device__ const uint4 kFirstZero = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void first_glds16(const char* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_addr)
                 : "memory");
}

template <int CIN, int MODE = kWgPlain>
__global__ __launch_bounds__(kFirstThreads, 2) void conv_first_wgrad_kernel(const FirstWgradArgs a) {
    constexpr int KMAX = 9 * CIN;
    constexpr int PER_C = (kGTH + 2) * (kGTW + 2), NVAL = CIN * PER_C;
    constexpr int NLD = (NVAL + kFirstThreads - 1) / kFirstThreads;
    constexpr int IMG_WORDS = kFMaxCin * (kGTH + 2) * kGLW;
    constexpr int DL = kGTileBytes / 16 / kFirstThreads;        // DMA transfers per thread per tile (8)
    // one LDS object: [dy stage 0 | dy stage 1 | image halo tile + zero word]; the stages double as the reduction buffer
    __shared__ __attribute__((aligned(16))) char smem[2 * kGTileBytes + (IMG_WORDS + 4) * 4];
    float* const img = reinterpret_cast<float*>(smem + 2 * kGTileBytes);
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int strip = wave & 1, half = wave >> 1;

    // image halo values of this thread: channel << 20 | row << 12 | column, or -1
    int src[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int v = i * kFirstThreads + tid;
        const int c = v / PER_C, r = (v - c * PER_C) / (kGTW + 2), xx = v - c * PER_C - r * (kGTW + 2);
        src[i] = v < NVAL ? ((c << 20) | (r << 12) | xx) : -1;
    }
    // dy slots of this thread: slot q = k * 256 + tid <-> (tile pixel q >> 3, stored slot q & 7), which receives the
    // channel slot (q & 7) ^ swz(column): the DMA destination is linear in the lane index, so the swizzle sits in the source
    auto swz = [](int col) { return ((col >> 1) & 3) << 1; };
    struct Tile { int n, y0, x0; };
    auto decode = [&](int t) {
        Tile r;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        r.x0 = tx * kGTW; r.y0 = (t % a.tiles_y) * kGTH; r.n = t / a.tiles_y;
        return r;
    };
    auto stage_dy = [&](const Tile& t, int st) {
#pragma unroll
        for (int k = 0; k < DL; ++k) {
            const int q = k * kFirstThreads + tid;
            const int pt = q >> 3, r = pt / kGTW, cx = pt - r * kGTW, sl = (q & 7) ^ swz(cx);
            const int gy = t.y0 + r, gx = t.x0 + cx;
            const bool ok = gy < a.H && gx < a.W;
            const char* p = ok ? reinterpret_cast<const char*>(a.dy + (((int64_t)t.n * a.H + gy) * a.W + gx) * a.ld + sl * 8)
                               : reinterpret_cast<const char*>(&kFirstZero);
            first_glds16(p, lds0 + st * kGTileBytes + (k * kFirstThreads + wave * 64) * 16);
        }
    };
    auto fetch = [&](const Tile& t, float (&val)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = src[i] >> 20, r = (src[i] >> 12) & 255, xx = src[i] & 4095;
            const int gy = t.y0 - 1 + r, gx = t.x0 - 1 + xx;
            const bool ok = src[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            val[i] = ok ? a.x[(((int64_t)t.n * CIN + c) * a.H + gy) * a.W + gx] : 0.f;
        }
    };
    auto deposit = [&](const float (&val)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = src[i] >> 20, r = (src[i] >> 12) & 255, xx = src[i] & 4095;
            if (src[i] >= 0) img[(c * (kGTH + 2) + r) * kGLW + xx] = val[i];
        }
    };

    // patch operand (A: rows = k): word offset of (row 0 of this wave's half, column 16*strip + 4g) for k = 16*kt + l15
    int abase[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int k = 16 * kt + l15;
        if (k < KMAX) {
            const int tap = k / CIN, c = k - tap * CIN;
            abase[kt] = (c * (kGTH + 2) + 4 * half + tap / 3) * kGLW + tap % 3 + 16 * strip + 4 * g;
        } else {
            abase[kt] = -1;
        }
    }
    // dy operand (B: columns = co) through the transposing read: lane 4q + p of a 16-lane group supplies pixel
    // (row h, column 4g + q), 8-byte unit ct*4 + p; byte offset within a stage for row 0 of this wave's half
    unsigned badr[4];
    {
        const int q = l15 >> 2, p = l15 & 3, col = 16 * strip + 4 * g + q;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int u = ct * 4 + p;
            badr[ct] = lds0 + ((4 * half) * kGTW + col) * 128 + (((u >> 1) ^ swz(col)) << 4) + (u & 1) * 8;
        }
    }
    typedef __attribute__((address_space(3))) char* lds_cptr;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

    f32x4 acc[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[kt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- recomputing modes: forward operands of this wave's 4 x 16 pixels, and the norm's constants of this lane's channels ----
    constexpr bool kRe = MODE != kWgPlain;
    bf16x8 wa[kRe ? 4 : 1];                 // weights, rows = output channels in MFMA order, K = patch index 8g .. 8g+7
    int foff[kRe ? 8 : 1];                  // image words of the patch values of pixel (row 4*half, column 16*strip + l15)
    float n_sc[kRe ? 2 : 1][8], n_sh[kRe ? 2 : 1][8];       // pre = acc * sc + sh (the convolution's bias folded into sh)
    float n_a[MODE == kWgDz ? 2 : 1][8], n_b[MODE == kWgDz ? 2 : 1][8];   // dz = sc*g' - n_a - n_b*acc
    float s1[MODE == kWgSums ? 2 : 1][8], s2[MODE == kWgSums ? 2 : 1][8];
    unsigned gadr[kRe ? 2 : 1];             // LDS byte offset (stage 0) of this lane's 16-byte channel slot, row 4*half, k2 = 0, 1
    if constexpr (kRe) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int co = first_row_channel(t * 16 + l15);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * g + e;
                float v = 0.f;
                if (k < KMAX) {
                    const int tap = k / CIN, c = k - tap * CIN;
                    v = a.w[((int64_t)co * CIN + c) * 9 + tap];
                }
                wa[t][e] = (bf16)v;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * g + e;
            if (k < KMAX) {
                const int tap = k / CIN, c = k - tap * CIN;
                foff[e] = (c * (kGTH + 2) + 4 * half + tap / 3) * kGLW + tap % 3 + 16 * strip + l15;
            } else {
                foff[e] = -1;
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int c = 32 * k2 + 8 * g + r;
                const float mean = a.nstats[2 * c], rstd = a.nstats[2 * c + 1], bs = a.bias ? a.bias[c] : 0.f;
                const float sc = rstd * (a.gamma ? a.gamma[c] : 1.f);
                n_sc[k2][r] = sc;
                n_sh[k2][r] = (a.beta ? a.beta[c] : 0.f) - mean * sc + bs * sc;
                if constexpr (MODE == kWgDz) {
                    // zhat = (acc + bias - mean) * rstd ;  dz = sc*(g' - m1 - zhat*m2)
                    const float m1 = a.red[2 * c] * a.inv_m, m2 = a.red[2 * c + 1] * a.inv_m;
                    n_b[k2][r] = sc * m2 * rstd;
                    n_a[k2][r] = sc * m1 + n_b[k2][r] * (bs - mean);
                }
                if constexpr (MODE == kWgSums) s1[k2][r] = s2[k2][r] = 0.f;
            }
            const int col = 16 * strip + l15;
            gadr[k2] = ((4 * half) * kGTW + col) * 128 + (((4 * k2 + g) ^ swz(col)) << 4);
        }
    }

    const int first = blockIdx.x, stride = gridDim.x;
    float cur[NLD];
    if (first < a.ntiles) {
        const Tile t0 = decode(first);
        stage_dy(t0, 0);
        fetch(t0, cur);
    }
    if (tid == 0) img[IMG_WORDS] = 0.f;
    int st = 0;
    for (int t = first; t < a.ntiles; t += stride, st ^= 1) {
        __syncthreads();                                  // everyone has finished the previous tile (image tile, other stage)
        deposit(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's dy has landed (this wave's transfers)
        __syncthreads();
        if (t + stride < a.ntiles) {
            const Tile tn = decode(t + stride);
            stage_dy(tn, st ^ 1);
            fetch(tn, cur);
        }
        if constexpr (kRe) {
            // z for this wave's 4 rows x 16 columns, then this lane's g vectors from the stage: sums, or dz written back in place
            const Tile tc = decode(t);
            char* const stage = smem + st * kGTileBytes;
            const bool col_ok = tc.x0 + 16 * strip + l15 < a.W;
#pragma unroll 1
            for (int row = 0; row < 4; ++row) {         // one row at a time: the registers go to the constants and the sums
                constexpr int j = 0;
                f32x4 z[4][1];
                {
                    bf16x8 pb;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pb[e] = (bf16)img[foff[e] >= 0 ? foff[e] + row * kGLW : IMG_WORDS];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        z[tt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tt], pb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                }
                {
                    const bool ok = col_ok && tc.y0 + 4 * half + row < a.H;
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        struct alignas(16) G8 { bf16 e[8]; };
                        G8* slot = reinterpret_cast<G8*>(stage + gadr[k2] + row * kGTW * 128);
                        const G8 gv = *slot;
                        G8 dz;
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const float zz = r < 4 ? z[2 * k2][j][r & 3] : z[2 * k2 + 1][j][r & 3];
                            const float pre = zz * n_sc[k2][r] + n_sh[k2][r];
                            const float gg = pre > 0.f ? (float)gv.e[r] : 0.f;
                            if constexpr (MODE == kWgSums) { s1[k2][r] += gg; s2[k2][r] += gg * zz; }
                            else dz.e[r] = (bf16)(ok ? n_sc[k2][r] * gg - n_a[k2][r] - n_b[k2][r] * zz : 0.f);
                        }
                        if constexpr (MODE == kWgDz) *slot = dz;
                    }
                }
            }
            if constexpr (MODE == kWgDz) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave reads back only what it wrote itself
        }
        if constexpr (MODE != kWgSums)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // patch fragments: K = 8g + 4h + q  <->  pixel (row 2*ks + h, column 4g + q) of this wave's strip
            bf16x8 fa[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        fa[kt][4 * h + q] = (bf16)img[abase[kt] >= 0 ? abase[kt] + (2 * ks + h) * kGLW + q : IMG_WORDS];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const unsigned adr = badr[ct] + st * kGTileBytes + (2 * ks) * kGTW * 128;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)((lds_cptr)(size_t)adr));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)((lds_cptr)(size_t)adr + kGTW * 128));
                s16x8 fb;
                fb[0] = lo[0]; fb[1] = lo[1]; fb[2] = lo[2]; fb[3] = lo[3]; fb[4] = hi[0]; fb[5] = hi[1]; fb[6] = hi[2]; fb[7] = hi[3];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    acc[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kt], __builtin_bit_cast(bf16x8, fb), acc[kt][ct], 0, 0, 0);
            }
        }
    }
    if constexpr (MODE == kWgSums) {
        // one partial row per wave: (sum g', sum g'*zhat) with zhat = (acc + bias - mean) * rstd applied to the two raw sums
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int co = 32 * k2 + 8 * g;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                s1[k2][r] = first_row16_sum(s1[k2][r]);
                s2[k2][r] = first_row16_sum(s2[k2][r]);
                const float mean = a.nstats[2 * (co + r)], rstd = a.nstats[2 * (co + r) + 1], bs = a.bias ? a.bias[co + r] : 0.f;
                s2[k2][r] = rstd * (s2[k2][r] + (bs - mean) * s1[k2][r]);
            }
            if (l15 == 0) {
                float* dst = a.part + ((int64_t)(blockIdx.x * 4 + wave) * 2) * 64 + co;
                *reinterpret_cast<float4*>(dst) = make_float4(s1[k2][0], s1[k2][1], s1[k2][2], s1[k2][3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(s1[k2][4], s1[k2][5], s1[k2][6], s1[k2][7]);
                *reinterpret_cast<float4*>(dst + 64) = make_float4(s2[k2][0], s2[k2][1], s2[k2][2], s2[k2][3]);
                *reinterpret_cast<float4*>(dst + 64 + 4) = make_float4(s2[k2][4], s2[k2][5], s2[k2][6], s2[k2][7]);
            }
        }
        return;
    }
    // ---- sum the four waves' blocks through LDS (the dy stages are free now), one slab per workgroup ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [4 waves][64 co][32 k]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            *reinterpret_cast<float4*>(&red[(wave * 64 + ct * 16 + l15) * 32 + kt * 16 + 4 * g]) =
                make_float4(acc[kt][ct][0], acc[kt][ct][1], acc[kt][ct][2], acc[kt][ct][3]);
    __syncthreads();
    float* out = a.slab + (int64_t)blockIdx.x * 64 * 32;
    for (int i = tid; i < 64 * 32; i += kFirstThreads) out[i] = (red[i] + red[2048 + i]) + (red[4096 + i] + red[6144 + i]);
}

// dw[co][c][tap] (OIHW) (+)= sum over the workgroups' slabs [s][co][k], k = tap * Cin + c, in fixed order
__global__ __launch_bounds__(kFirstThreads) void conv_first_wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, int Cin,
                                                                                int Cout, float* __restrict__ dw, int accumulate) {
    __shared__ float part[8][32];
    const int co = blockIdx.x, k = threadIdx.x & 31, p = threadIdx.x >> 5;
    float s = 0.f;
    for (int i = p; i < nslab; i += 8) s += slab[((int64_t)i * 64 + co) * 32 + k];
    part[p][k] = s;
    __syncthreads();
    if (p == 0 && k < 9 * Cin && co < Cout) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += part[j][k];
        const int tap = k / Cin, c = k - tap * Cin;
        float* d = dw + ((int64_t)co * Cin + c) * 9 + tap;
        *d = accumulate ? *d + v : v;
    }
}

}  // namespace segk

using namespace segk;

extern "C" int64_t segk_conv_first_stat_rows(int64_t N, int64_t H, int64_t W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return N * ((H + kFTH - 1) / kFTH) * ((W + kFTW - 1) / kFTW) * 4;
}

extern "C" int segk_conv_first_fwd(const float* x, int64_t N, int64_t Cin, int64_t H, int64_t W, const float* w, const float* bias,
                                   int64_t Cout, void* y, int y_dtype, int64_t ldy, float* stat_partials, int flags, int dev,
                                   void* stream) {
    SEGK_REQUIRE(x && w && y, SEGK_EINVAL, "conv_first_fwd: null pointer");
    SEGK_REQUIRE(y_dtype == SEGK_BF16, SEGK_EINVAL, "conv_first_fwd: bf16 output only (fp32 runs as a 1x1 convolution over segk_im2col3x3)");
    SEGK_REQUIRE(Cin >= 1 && Cin <= kFMaxCin, SEGK_ESHAPE, "conv_first_fwd: %lld input channels (1..3)", (long long)Cin);
    SEGK_REQUIRE(Cout == 32 || Cout == 64, SEGK_ESHAPE, "conv_first_fwd: %lld output channels (32 or 64)", (long long)Cout);
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0 && N * H * W < (1ll << 31), SEGK_ESHAPE, "conv_first_fwd: bad N/H/W");
    SEGK_REQUIRE(ldy >= Cout && aligned16(y) && (ldy * 2) % 16 == 0, SEGK_EALIGN, "conv_first_fwd: output pointer/ld misaligned or ld < Cout");
    SEGK_REQUIRE(!bias || aligned16(bias), SEGK_EALIGN, "conv_first_fwd: bias must be 16-byte aligned");
    SEGK_REQUIRE(!stat_partials || aligned16(stat_partials), SEGK_EALIGN, "conv_first_fwd: statistics partials must be 16-byte aligned");
    SEGK_USE_DEVICE(dev);
    FirstArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = (bf16*)y; a.ldy = ldy; a.stat_part = stat_partials;
    a.N = (int)N; a.Cin = (int)Cin; a.H = (int)H; a.W = (int)W; a.Cout = (int)Cout;
    a.tiles_x = (int)((W + kFTW - 1) / kFTW); a.tiles_y = (int)((H + kFTH - 1) / kFTH);
    a.relu = flags & 1;
    const int64_t grid = N * a.tiles_x * ((a.tiles_y + kTilesPerWg - 1) / kTilesPerWg);
    SEGK_REQUIRE(grid < (1ll << 31), SEGK_ESHAPE, "conv_first_fwd: grid too large");
    if (Cout == 64) launch_first<4>(a, (unsigned)grid, (hipStream_t)stream);
    else launch_first<2>(a, (unsigned)grid, (hipStream_t)stream);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// The first convolution followed by BatchNorm2d + ReLU (models/unetvgg.py:72 conv1.0-2; models/unetplusplus.py:45 conv0_0.conv1/bn1)
// without ever storing the convolution's output: mode 1 = statistics partials of z = conv(x) + bias ([segk_conv_first_stat_rows][2][64],
// for segk_norm_stats_finalize), 2 = out <- relu(norm(z)), 3 = partials of the norm backward's two sums from g = dL/d relu(norm(z))
// (for segk_norm_bwd_sums_finalize), 4 = out <- dL/dz.  Every mode recomputes z from the 1-3 channel image.
extern "C" int segk_conv_first_norm(int mode, const float* x, int64_t N, int64_t Cin, int64_t H, int64_t W, const float* w,
                                    const float* bias, int64_t Cout, const float* stats, const float* gamma, const float* beta,
                                    const float* red, const void* g, int64_t ldg, void* out, int64_t ld_out, float* partials, int dev,
                                    void* stream) {
    SEGK_REQUIRE(mode >= kStatsOnly && mode <= kBwdDz, SEGK_EINVAL, "conv_first_norm: mode %d (1..4)", mode);
    SEGK_REQUIRE(x && w, SEGK_EINVAL, "conv_first_norm: null pointer");
    SEGK_REQUIRE(Cin >= 1 && Cin <= kFMaxCin, SEGK_ESHAPE, "conv_first_norm: %lld input channels (1..3)", (long long)Cin);
    SEGK_REQUIRE(Cout == 64, SEGK_ESHAPE, "conv_first_norm: %lld output channels (64)", (long long)Cout);
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0 && N * H * W < (1ll << 31), SEGK_ESHAPE, "conv_first_norm: bad N/H/W");
    const bool sums = mode == kStatsOnly || mode == kBwdSums, store = mode == kNormRelu || mode == kBwdDz;
    SEGK_REQUIRE(!sums || (partials && aligned16(partials)), SEGK_EINVAL, "conv_first_norm: modes 1 and 3 need a 16-byte aligned partials buffer");
    SEGK_REQUIRE(!store || (out && ld_out >= Cout && aligned16(out) && (ld_out * 2) % 16 == 0), SEGK_EALIGN,
                 "conv_first_norm: modes 2 and 4 need an aligned bf16 output with ld >= Cout");
    SEGK_REQUIRE(mode == kStatsOnly || stats, SEGK_EINVAL, "conv_first_norm: modes 2-4 need the statistics vector");
    SEGK_REQUIRE(mode < kBwdSums || (g && ldg >= Cout && aligned16(g) && (ldg * 2) % 16 == 0), SEGK_EALIGN,
                 "conv_first_norm: modes 3 and 4 need an aligned bf16 gradient with ld >= Cout");
    SEGK_REQUIRE(mode != kBwdDz || red, SEGK_EINVAL, "conv_first_norm: mode 4 needs the reduced sums");
    SEGK_REQUIRE(!bias || aligned16(bias), SEGK_EALIGN, "conv_first_norm: bias must be 16-byte aligned");
    SEGK_USE_DEVICE(dev);
    FirstArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = (bf16*)out; a.ldy = ld_out; a.stat_part = sums ? partials : nullptr;
    a.N = (int)N; a.Cin = (int)Cin; a.H = (int)H; a.W = (int)W; a.Cout = (int)Cout;
    a.tiles_x = (int)((W + kFTW - 1) / kFTW); a.tiles_y = (int)((H + kFTH - 1) / kFTH);
    a.relu = 0;
    a.nstats = stats; a.gamma = gamma; a.beta = beta; a.red = red; a.inv_m = 1.f / (float)(N * H * W);
    a.g = (const bf16*)g; a.ldg = ldg;
    const int64_t grid = N * a.tiles_x * ((a.tiles_y + kTilesPerWg - 1) / kTilesPerWg);
    SEGK_REQUIRE(grid < (1ll << 31), SEGK_ESHAPE, "conv_first_norm: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (mode == kStatsOnly) launch_first<4, kStatsOnly>(a, (unsigned)grid, st);
    else if (mode == kNormRelu) launch_first<4, kNormRelu>(a, (unsigned)grid, st);
    else if (mode == kBwdSums) launch_first<4, kBwdSums>(a, (unsigned)grid, st);
    else launch_first<4, kBwdDz>(a, (unsigned)grid, st);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

static int first_wgrad_grid(int64_t ntiles) { return (int)(ntiles < 512 ? ntiles : 512); }

extern "C" size_t segk_conv_first_wgrad_workspace(int64_t N, int64_t H, int64_t W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t ntiles = N * ((H + kGTH - 1) / kGTH) * ((W + kGTW - 1) / kGTW);
    return (size_t)first_wgrad_grid(ntiles) * 64 * 32 * sizeof(float);
}

// Backward of Conv2d(Cin, 64) -> BatchNorm2d (batch statistics) -> ReLU from g = dL/d(output), the image and the parameters:
// two streaming passes over g.  Pass 1 recomputes z per tile and reduces the norm's two sums (-> dgamma, dbeta); pass 2
// recomputes z again, turns g into dL/dz inside the LDS stage and multiplies it into the weight gradient.  Neither z nor
// dL/dz is ever written to memory.
static size_t first_norm_bwd_sections(int64_t N, int64_t H, int64_t W, size_t* part_off, size_t* red_off, size_t* fin_off) {
    const int64_t ntiles = N * ((H + kGTH - 1) / kGTH) * ((W + kGTW - 1) / kGTW);
    const size_t grid = (size_t)first_wgrad_grid(ntiles);
    size_t off = align_up(grid * 64 * 32 * sizeof(float), 256);
    *part_off = off; off += align_up(grid * 4 * 2 * 64 * sizeof(float), 256);
    *red_off = off; off += 256 * sizeof(float);
    *fin_off = off; off += align_up(segk_norm_stats_finalize_workspace(1, (int64_t)grid * 4, 64), 256);
    return off;
}

extern "C" size_t segk_conv_first_norm_bwd_workspace(int64_t N, int64_t H, int64_t W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    size_t a, b, c;
    return first_norm_bwd_sections(N, H, W, &a, &b, &c);
}

extern "C" int segk_conv_first_norm_bwd(const float* x, int64_t N, int64_t Cin, int64_t H, int64_t W, const float* w, const float* bias,
                                        const float* stats, const float* gamma, const float* beta, const void* g, int64_t ldg,
                                        float* dw, float* dgamma, float* dbeta, int flags, void* workspace, size_t workspace_bytes,
                                        int dev, void* stream) {
    SEGK_REQUIRE(x && w && stats && g && dw && workspace, SEGK_EINVAL, "conv_first_norm_bwd: null pointer");
    SEGK_REQUIRE(Cin >= 1 && Cin <= kFMaxCin, SEGK_ESHAPE, "conv_first_norm_bwd: %lld input channels (1..3)", (long long)Cin);
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0 && N * H * W < (1ll << 31), SEGK_ESHAPE, "conv_first_norm_bwd: bad N/H/W");
    SEGK_REQUIRE(ldg >= 64 && aligned16(g) && (ldg * 2) % 16 == 0, SEGK_EALIGN, "conv_first_norm_bwd: gradient pointer/ld misaligned or ld < 64");
    SEGK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), SEGK_EINVAL, "conv_first_norm_bwd: dgamma/dbeta come in pairs");
    SEGK_REQUIRE(!bias || aligned16(bias), SEGK_EALIGN, "conv_first_norm_bwd: bias must be 16-byte aligned");
    size_t part_off, red_off, fin_off;
    const size_t need = first_norm_bwd_sections(N, H, W, &part_off, &red_off, &fin_off);
    SEGK_REQUIRE(workspace_bytes >= need, SEGK_EWORKSPACE, "conv_first_norm_bwd: workspace too small");
    SEGK_USE_DEVICE(dev);
    FirstWgradArgs a;
    a.x = x; a.dy = (const bf16*)g; a.ld = ldg; a.slab = (float*)workspace;
    a.N = (int)N; a.Cin = (int)Cin; a.H = (int)H; a.W = (int)W;
    a.tiles_x = (int)((W + kGTW - 1) / kGTW); a.tiles_y = (int)((H + kGTH - 1) / kGTH);
    a.ntiles = (int)(N * a.tiles_x * a.tiles_y);
    a.w = w; a.bias = bias; a.nstats = stats; a.gamma = gamma; a.beta = beta;
    a.part = (float*)((char*)workspace + part_off);
    float* red = (float*)((char*)workspace + red_off);
    a.red = red; a.inv_m = 1.f / (float)(N * H * W);
    const int grid = first_wgrad_grid(a.ntiles);
    hipStream_t st = (hipStream_t)stream;
    // flags bit2: the norm's half only (first pass + reduction: dgamma, dbeta, the two sums left in the workspace);
    // flags bit3: the convolution's half only (second pass, from the sums a bit2 call left in the same workspace)
    const bool norm_half = !(flags & 8), conv_half = !(flags & 4);
    SEGK_REQUIRE(norm_half || conv_half, SEGK_EINVAL, "conv_first_norm_bwd: flags bit2 and bit3 exclude each other");
    if (norm_half) {
        if (Cin == 1) hipLaunchKernelGGL((conv_first_wgrad_kernel<1, kWgSums>), dim3(grid), dim3(kFirstThreads), 0, st, a);
        else if (Cin == 2) hipLaunchKernelGGL((conv_first_wgrad_kernel<2, kWgSums>), dim3(grid), dim3(kFirstThreads), 0, st, a);
        else hipLaunchKernelGGL((conv_first_wgrad_kernel<3, kWgSums>), dim3(grid), dim3(kFirstThreads), 0, st, a);
        SEGK_LAUNCH_CHECK();
        if (int rc = segk_norm_bwd_sums_finalize(a.part, (int64_t)grid * 4, 64, red, dgamma, dbeta, flags & 2, (char*)workspace + fin_off,
                                                 need - fin_off, dev, stream))
            return rc;
    }
    if (!conv_half) return SEGK_OK;
    if (Cin == 1) hipLaunchKernelGGL((conv_first_wgrad_kernel<1, kWgDz>), dim3(grid), dim3(kFirstThreads), 0, st, a);
    else if (Cin == 2) hipLaunchKernelGGL((conv_first_wgrad_kernel<2, kWgDz>), dim3(grid), dim3(kFirstThreads), 0, st, a);
    else hipLaunchKernelGGL((conv_first_wgrad_kernel<3, kWgDz>), dim3(grid), dim3(kFirstThreads), 0, st, a);
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv_first_wgrad_reduce_kernel, dim3(64), dim3(kFirstThreads), 0, st, (const float*)workspace, grid, (int)Cin, 64, dw,
                       (flags >> 1) & 1);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_conv_first_wgrad(const float* x, int64_t N, int64_t Cin, int64_t H, int64_t W, const void* dy, int dy_dtype,
                                     int64_t ld_dy, int64_t Cout, float* dw, int flags, void* workspace, size_t workspace_bytes,
                                     int dev, void* stream) {
    SEGK_REQUIRE(x && dy && dw && workspace, SEGK_EINVAL, "conv_first_wgrad: null pointer");
    SEGK_REQUIRE(dy_dtype == SEGK_BF16, SEGK_EINVAL, "conv_first_wgrad: bf16 gradients only");
    SEGK_REQUIRE(Cin >= 1 && Cin <= kFMaxCin, SEGK_ESHAPE, "conv_first_wgrad: %lld input channels (1..3)", (long long)Cin);
    SEGK_REQUIRE(Cout == 64, SEGK_ESHAPE, "conv_first_wgrad: %lld output channels (64)", (long long)Cout);
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0 && N * H * W < (1ll << 31), SEGK_ESHAPE, "conv_first_wgrad: bad N/H/W");
    SEGK_REQUIRE(ld_dy >= Cout && aligned16(dy) && (ld_dy * 2) % 16 == 0, SEGK_EALIGN, "conv_first_wgrad: dy pointer/ld misaligned or ld < Cout");
    SEGK_REQUIRE(workspace_bytes >= segk_conv_first_wgrad_workspace(N, H, W), SEGK_EWORKSPACE, "conv_first_wgrad: workspace too small");
    SEGK_USE_DEVICE(dev);
    FirstWgradArgs a;
    a.x = x; a.dy = (const bf16*)dy; a.ld = ld_dy; a.slab = (float*)workspace;
    a.N = (int)N; a.Cin = (int)Cin; a.H = (int)H; a.W = (int)W;
    a.tiles_x = (int)((W + kGTW - 1) / kGTW); a.tiles_y = (int)((H + kGTH - 1) / kGTH);
    a.ntiles = (int)(N * a.tiles_x * a.tiles_y);
    const int grid = first_wgrad_grid(a.ntiles);
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 1) hipLaunchKernelGGL((conv_first_wgrad_kernel<1>), dim3(grid), dim3(kFirstThreads), 0, st, a);
    else if (Cin == 2) hipLaunchKernelGGL((conv_first_wgrad_kernel<2>), dim3(grid), dim3(kFirstThreads), 0, st, a);
    else hipLaunchKernelGGL((conv_first_wgrad_kernel<3>), dim3(grid), dim3(kFirstThreads), 0, st, a);
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv_first_wgrad_reduce_kernel, dim3(64), dim3(kFirstThreads), 0, st, (const float*)workspace, grid, (int)Cin,
                       (int)Cout, dw, (flags >> 1) & 1);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}
