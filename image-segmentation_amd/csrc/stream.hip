// HBM-bound streaming kernels on NHWC activations: layout conversion, BatchNorm/InstanceNorm
// statistics / apply / backward, ReLU masks, 2x2 max pool, bilinear x2 up-sampling (fused with
// InstanceNorm-apply + ReLU), column sums, the 1x1 classification head and the Adam update.
// Every thread moves 8 consecutive channels (16 B of bf16 / 32 B of fp32) of one pixel, so a
// wavefront touches whole 128-B lines; reductions are two-stage with a fixed-order finalize.
#include "common.h"

namespace segk {

constexpr int kThreads = 256;

// Tensors far larger than the memory-side cache (256 MB) are swept with non-temporal accesses and four independent
// 16-byte vectors in flight per lane: measured on this part, a 2-reads-1-write sweep of 604 MB buffers runs at 4.5-5.4 TB/s
// with plain one-vector-at-a-time accesses and at 5.9-6.3 TB/s this way.  Smaller tensors keep plain accesses: their
// consumer finds them in the memory-side cache, and non-temporal accesses cost 15-25 % there.
constexpr int64_t kBigTensorBytes = 200ll << 20;
#ifndef SEGK_BWD_U
#define SEGK_BWD_U 2
#endif
constexpr int kBwdRowsInFlight = SEGK_BWD_U;   // row groups per trip in the backward passes of big tensors
template <typename T, bool NT> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]) {
    if constexpr (NT) Elem<T>::load8s(p, v);
    else Elem<T>::load8(p, v);
}
template <typename T, bool NT> __device__ __forceinline__ void st8(T* p, const float (&v)[8]) {
    if constexpr (NT) Elem<T>::store8s(p, v);
    else Elem<T>::store8(p, v);
}

static inline int grid_for(int64_t items, int cap = 8192) {
    int64_t g = (items + kThreads - 1) / kThreads;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------ layout conversion
template <typename T>
__global__ __launch_bounds__(kThreads) void nchw_to_nhwc_kernel(const float* __restrict__ src, int64_t N, int Cin,
                                                                int64_t HW, T* __restrict__ dst, int Cout, int64_t ld) {
    const int cv = Cout / 8;
    const int64_t items = N * HW * cv;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kThreads) {
        const int v = (int)(i % cv);
        const int64_t p = i / cv, n = p / HW, hw = p % HW;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = v * 8 + j;
            o[j] = c < Cin ? src[(n * Cin + c) * HW + hw] : 0.f;
        }
        Elem<T>::store8(dst + p * ld + v * 8, o);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void im2col3x3_kernel(const float* __restrict__ src, int64_t N, int Cin, int H,
                                                             int W, T* __restrict__ dst, int Cout, int64_t ld) {
    const int cv = Cout / 8;
    const int64_t HW = (int64_t)H * W, items = N * HW * cv;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kThreads) {
        const int v = (int)(i % cv);
        const int64_t p = i / cv, n = p / HW;
        const int hw = (int)(p % HW), y = hw / W, x = hw % W;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = v * 8 + j;
            float val = 0.f;
            if (k < 9 * Cin) {
                const int tap = k / Cin, c = k % Cin;
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) val = src[(n * Cin + c) * HW + (int64_t)yy * W + xx];
            }
            o[j] = val;
        }
        Elem<T>::store8(dst + p * ld + v * 8, o);
    }
}

// Same patch image, one thread per pixel, for the shapes the networks use (1-3 input channels, 32 stored columns): the
// column -> (tap, channel) mapping is resolved at compile time and neighbouring lanes read neighbouring pixels of one plane.
template <typename T, int CIN>
__global__ __launch_bounds__(kThreads) void im2col3x3_pixel_kernel(const float* __restrict__ src, int64_t N, int H, int W,
                                                                   T* __restrict__ dst, int64_t ld) {
    const int64_t HW = (int64_t)H * W, items = N * HW;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kThreads) {
        const int64_t n = i / HW;
        const int hw = (int)(i % HW), y = hw / W, x = hw % W;
        const float* img = src + n * CIN * HW;
        float o[4][8];
#pragma unroll
        for (int k = 9 * CIN; k < 32; ++k) o[k / 8][k % 8] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
#pragma unroll
            for (int c = 0; c < CIN; ++c) o[(tap * CIN + c) / 8][(tap * CIN + c) % 8] = ok ? img[c * HW + (int64_t)yy * W + xx] : 0.f;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) Elem<T>::store8(dst + i * ld + v * 8, o[v]);
    }
}

// ------------------------------------------------------------------------------ statistics
// Block (b, g) sums its pixel range of group g for all channels.  partial layout:
//   ws[((g*bpg + b)*2 + k)*C + c],  k = 0: sum, 1: sum of squares (or the two backward sums)
// MODE 0 sums (x - K) and (x - K)^2 with K = the group's first row (kept in kshift[g*C + c]): the variance then comes
// from numbers of the size of the spread, not of the mean, so E[d^2] - E[d]^2 does not cancel when |mean| >> sigma
// (few-sample BatchNorms of un-normalised features).
template <typename T, int MODE>  // MODE 0: shifted sum, shifted sum of squares ; MODE 1: column sum only
__global__ __launch_bounds__(kThreads) void stats_partial_kernel(const T* __restrict__ x, int64_t ld, int64_t M, int C,
                                                                 int bpg, float* __restrict__ ws, float* __restrict__ kshift) {
    __shared__ float red[kThreads][17];
    const int cv = C / 8, rows = kThreads / cv;
    const int v = threadIdx.x % cv, r = threadIdx.x / cv;
    const int g = blockIdx.y, b = blockIdx.x;
    const int64_t per = (M + bpg - 1) / bpg;
    const int64_t lo = (int64_t)b * per, hi = (lo + per < M) ? lo + per : M;
    const T* base = x + (int64_t)g * M * ld + v * 8;
    float s[8], q[8], kk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = kk[j] = 0.f;
    if (MODE == 0 && r < rows) {
        Elem<T>::load8(base, kk);
        if (b == 0 && r == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) kshift[(int64_t)g * C + v * 8 + j] = kk[j];
        }
    }
    for (int64_t p = lo + r; p < hi && r < rows; p += rows) {   // r >= rows: spare threads when C/8 does not divide the block
        float a[8];
        Elem<T>::load8(base + p * ld, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = a[j] - kk[j];
            s[j] += d;
            if (MODE == 0) q[j] += d * d;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = s[j]; red[threadIdx.x][8 + j] = q[j]; }
    __syncthreads();
    // thread t < C*2 produces channel t%C of sum kind t/C, adding the rows in fixed order
    for (int t = threadIdx.x; t < 2 * C; t += kThreads) {
        const int k = t / C, c = t % C;
        if (MODE == 1 && k == 1) continue;
        float acc = 0.f;
        for (int rr = 0; rr < rows; ++rr) acc += red[rr * cv + c / 8][k * 8 + (c & 7)];
        ws[(((int64_t)g * bpg + b) * 2 + k) * C + c] = acc;
    }
}

// Fixed-order sum of per-block partials by one wavefront: lanes take the partials round-robin, a butterfly adds them.
__device__ __forceinline__ double wave_sum_partials(const float* __restrict__ ws, int64_t base, int64_t stride, int bpg) {
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int b = lane; b < bpg; b += 64) s += (double)ws[base + (int64_t)b * stride];
    return wave_sum_d(s);
}

// Fixed-order sum of the per-block partials of one (group, kind, channel) by a whole workgroup: thread t adds
// partials t, t+256, ... in double, a butterfly adds the lanes of each wavefront and wave 0's lane 0 adds the four
// wave sums -- the same order every run (deterministic), and a latency of one round of loads instead of bpg/64.
__device__ __forceinline__ double block_sum_partials(const float* __restrict__ ws, int64_t base, int64_t stride, int bpg,
                                                     double* smem4) {
    double s = 0.0;
    for (int b = threadIdx.x; b < bpg; b += kThreads) s += (double)ws[base + (int64_t)b * stride];
    s = wave_sum_d(s);
    __syncthreads();                       // smem4 may still be read from a previous call
    if ((threadIdx.x & 63) == 0) smem4[threadIdx.x >> 6] = s;
    __syncthreads();
    return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

// mean / rstd per (group, channel) + optional running-buffer update (BatchNorm, groups == 1).
// One workgroup per (channel, group).
__global__ __launch_bounds__(kThreads) void stats_finalize_kernel(const float* __restrict__ ws, int bpg, int64_t M, int C,
                                                                  float eps, float* __restrict__ stats,
                                                                  float* __restrict__ rmean, float* __restrict__ rvar,
                                                                  float momentum, const float* __restrict__ kshift,
                                                                  int64_t row_stride) {
    __shared__ double sm[4];
    const int g = blockIdx.y, c = blockIdx.x;
    const int64_t base = (int64_t)g * bpg * row_stride + c;   // row_stride: floats between consecutive partial rows (2*C when dense)
    const double s = block_sum_partials(ws, base, row_stride, bpg, sm);
    const double q = block_sum_partials(ws, base + C, row_stride, bpg, sm);
    if (threadIdx.x != 0) return;
    const double dm = s / (double)M;                       // mean of (x - K)
    const double mean = (kshift ? (double)kshift[(int64_t)g * C + c] : 0.0) + dm;
    double var = q / (double)M - dm * dm;
    if (var < 0.0) var = 0.0;
    stats[((int64_t)g * C + c) * 2 + 0] = (float)mean;
    stats[((int64_t)g * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean && g == 0) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mean);
        rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unbiased);
    }
}

__global__ __launch_bounds__(kThreads) void eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                              int C, float eps, float* __restrict__ stats) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= C) return;
    stats[c * 2 + 0] = rmean[c];
    stats[c * 2 + 1] = 1.f / sqrtf(rvar[c] + eps);
}

__global__ __launch_bounds__(kThreads) void colsum_finalize_kernel(const float* __restrict__ ws, int bpg, int C,
                                                                   float* __restrict__ out, int accumulate) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    const double s = block_sum_partials(ws, c, 2 * (int64_t)C, bpg, sm);
    if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// ------------------------------------------------------------------------------ normalise (+ReLU)
template <typename T, bool BIG>
__global__ __launch_bounds__(kThreads) void norm_apply_kernel(const T* __restrict__ x, int64_t ldx, int64_t M, int C,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int relu,
                                                              T* __restrict__ y, int64_t ldy) {
    const int cv = C / 8, rows = kThreads / cv;
    const int v = threadIdx.x % cv, r = threadIdx.x / cv;
    const int g = blockIdx.y;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        const float mean = stats[((int64_t)g * C + c) * 2], rstd = stats[((int64_t)g * C + c) * 2 + 1];
        sc[j] = rstd * (gamma ? gamma[c] : 1.f);
        sh[j] = (beta ? beta[c] : 0.f) - mean * sc[j];
    }
    const T* xb = x + (int64_t)g * M * ldx + v * 8;
    T* yb = y + (int64_t)g * M * ldy + v * 8;
    if (r >= rows) return;
    if constexpr (BIG) {
        // a block sweeps 4*rows consecutive pixels per trip: four vectors in flight per lane, 4 x 4 KB contiguous per block
        const int64_t chunk = (int64_t)rows * 4;
        for (int64_t base = (int64_t)blockIdx.x * chunk + r; base < M; base += (int64_t)gridDim.x * chunk) {
            float a[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + u * rows < M) ld8<T, true>(xb + (base + u * rows) * ldx, a[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (base + u * rows >= M) continue;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a[u][j] = a[u][j] * sc[j] + sh[j];
                    if (relu) a[u][j] = fmaxf(a[u][j], 0.f);
                }
                st8<T, true>(yb + (base + u * rows) * ldy, a[u]);
            }
        }
        return;
    }
    // four rows per trip, all loads issued before the first use: more bytes in flight per wave
    const int64_t step = (int64_t)gridDim.x * rows;
    int64_t p = (int64_t)blockIdx.x * rows + r;
    for (; p + 3 * step < M; p += 4 * step) {
        float a[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) ld8<T, BIG>(xb + (p + u * step) * ldx, a[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[u][j] = a[u][j] * sc[j] + sh[j];
                if (relu) a[u][j] = fmaxf(a[u][j], 0.f);
            }
            st8<T, BIG>(yb + (p + u * step) * ldy, a[u]);
        }
    }
    for (; p < M; p += step) {
        float a[8];
        ld8<T, BIG>(xb + p * ldx, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = a[j] * sc[j] + sh[j];
            if (relu) a[j] = fmaxf(a[j], 0.f);
        }
        st8<T, BIG>(yb + p * ldy, a);
    }
}

// backward, pass 1: per block  s1 = sum g, s2 = sum g*xhat   (g = dy masked by the ReLU of the forward)
template <typename T, bool BIG>
__global__ __launch_bounds__(kThreads) void norm_bwd_partial_kernel(const T* __restrict__ dy, int64_t ld_dy,
                                                                    const T* __restrict__ x, int64_t ldx, int64_t M, int C,
                                                                    const float* __restrict__ stats,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int relu, int bpg,
                                                                    float* __restrict__ ws) {
    __shared__ float red[kThreads][17];
    const int cv = C / 8, rows = kThreads / cv;
    const int v = threadIdx.x % cv, r = threadIdx.x / cv;
    const int g = blockIdx.y, b = blockIdx.x;
    float mean[8], rstd[8], ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        mean[j] = stats[((int64_t)g * C + c) * 2];
        rstd[j] = stats[((int64_t)g * C + c) * 2 + 1];
        ga[j] = gamma ? gamma[c] : 1.f;
        be[j] = beta ? beta[c] : 0.f;
    }
    const int64_t per = (M + bpg - 1) / bpg;
    const int64_t lo = (int64_t)b * per, hi = (lo + per < M) ? lo + per : M;
    const T* xb = x + (int64_t)g * M * ldx + v * 8;
    const T* db = dy + (int64_t)g * M * ld_dy + v * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    auto add_row = [&](const float (&a)[8], const float (&d)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sc = rstd[j] * ga[j];
            const float pre = a[j] * sc + (be[j] - mean[j] * sc);   // same expression as the forward
            const float xh = (a[j] - mean[j]) * rstd[j];
            const float gg = (relu && !(pre > 0.f)) ? 0.f : d[j];
            s1[j] += gg;
            s2[j] += gg * xh;
        }
    };
    int64_t p = lo + r;
    if constexpr (BIG) {
        // U consecutive row groups per trip: 2*U vectors in flight per lane
        constexpr int U = kBwdRowsInFlight;
        for (; p + (U - 1) * rows < hi && r < rows; p += U * rows) {
            float a[U][8], d[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ld8<T, true>(xb + (p + u * rows) * ldx, a[u]);
                ld8<T, true>(db + (p + u * rows) * ld_dy, d[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) add_row(a[u], d[u]);
        }
    }
    for (; p < hi && r < rows; p += rows) {   // r >= rows: spare threads when C/8 does not divide the block
        float a[8], d[8];
        ld8<T, BIG>(xb + p * ldx, a);
        ld8<T, BIG>(db + p * ld_dy, d);
        add_row(a, d);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = s1[j]; red[threadIdx.x][8 + j] = s2[j]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += kThreads) {
        const int k = t / C, c = t % C;
        float acc = 0.f;
        for (int rr = 0; rr < rows; ++rr) acc += red[rr * cv + c / 8][k * 8 + (c & 7)];
        ws[(((int64_t)g * bpg + b) * 2 + k) * C + c] = acc;
    }
}

// pass 1b: red[(g*C+c)*2 + {0,1}] = {s1, s2}; dgamma/dbeta.  One workgroup per (channel, group).
__global__ __launch_bounds__(kThreads) void norm_bwd_finalize_kernel(const float* __restrict__ ws, int bpg, int C,
                                                                     float* __restrict__ red, float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta, int accumulate, int64_t row_stride) {
    __shared__ double sm[4];
    const int g = blockIdx.y, c = blockIdx.x;
    const int64_t base = (int64_t)g * bpg * row_stride + c;   // row_stride: floats between consecutive partial rows (2*C when dense)
    const double s1 = block_sum_partials(ws, base, row_stride, bpg, sm);
    const double s2 = block_sum_partials(ws, base + C, row_stride, bpg, sm);
    if (threadIdx.x != 0) return;
    red[((int64_t)g * C + c) * 2 + 0] = (float)s1;
    red[((int64_t)g * C + c) * 2 + 1] = (float)s2;
    if (dgamma && g == 0) {
        dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
        dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    }
}

// pass 2: dx = gamma*rstd*(g - s1/M - xhat*s2/M)     (frozen: dx = gamma*rstd*g)
template <typename T, bool BIG>
__global__ __launch_bounds__(kThreads) void norm_bwd_dx_kernel(const T* __restrict__ dy, int64_t ld_dy,
                                                               const T* __restrict__ x, int64_t ldx, int64_t M, int C,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int relu, int frozen,
                                                               const float* __restrict__ red, T* __restrict__ dx,
                                                               int64_t ld_dx, int add_dx) {
    const int cv = C / 8, rows = kThreads / cv;
    const int v = threadIdx.x % cv, r = threadIdx.x / cv;
    const int g = blockIdx.y;
    float mean[8], rstd[8], ga[8], be[8], m1[8], m2[8];
    const float invM = 1.f / (float)M;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        mean[j] = stats[((int64_t)g * C + c) * 2];
        rstd[j] = stats[((int64_t)g * C + c) * 2 + 1];
        ga[j] = gamma ? gamma[c] : 1.f;
        be[j] = beta ? beta[c] : 0.f;
        m1[j] = frozen ? 0.f : red[((int64_t)g * C + c) * 2] * invM;
        m2[j] = frozen ? 0.f : red[((int64_t)g * C + c) * 2 + 1] * invM;
    }
    const T* xb = x + (int64_t)g * M * ldx + v * 8;
    const T* db = dy + (int64_t)g * M * ld_dy + v * 8;
    T* ob = dx + (int64_t)g * M * ld_dx + v * 8;
    auto row = [&](const float (&a)[8], float (&d)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sc = rstd[j] * ga[j];
            const float pre = a[j] * sc + (be[j] - mean[j] * sc);
            const float xh = (a[j] - mean[j]) * rstd[j];
            const float gg = (relu && !(pre > 0.f)) ? 0.f : d[j];
            d[j] = sc * (gg - m1[j] - xh * m2[j]);
        }
    };
    if constexpr (BIG) {
        if (r >= rows) return;
        // a block sweeps U*rows consecutive pixels per trip: 2*U (3*U with add_dx) vectors in flight per lane
        constexpr int U = kBwdRowsInFlight;
        const int64_t chunk = (int64_t)rows * U;
        for (int64_t base = (int64_t)blockIdx.x * chunk + r; base < M; base += (int64_t)gridDim.x * chunk) {
            float a[U][8], d[U][8], o[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (base + u * rows >= M) continue;
                ld8<T, true>(xb + (base + u * rows) * ldx, a[u]);
                ld8<T, true>(db + (base + u * rows) * ld_dy, d[u]);
                if (add_dx) ld8<T, true>(ob + (base + u * rows) * ld_dx, o[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (base + u * rows >= M) continue;
                row(a[u], d[u]);
                if (add_dx) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[u][j] += o[u][j];
                }
                st8<T, true>(ob + (base + u * rows) * ld_dx, d[u]);
            }
        }
        return;
    }
    for (int64_t p = (int64_t)blockIdx.x * rows + r; p < M && r < rows; p += (int64_t)gridDim.x * rows) {
        float a[8], d[8];
        ld8<T, BIG>(xb + p * ldx, a);
        ld8<T, BIG>(db + p * ld_dy, d);
        row(a, d);
        if (add_dx) {
            float o[8];
            ld8<T, BIG>(ob + p * ld_dx, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] += o[j];
        }
        st8<T, BIG>(ob + p * ld_dx, d);
    }
}

// ------------------------------------------------------------------------------ norm + ReLU + 2x2 max pool, fused
// The last BatchNorm of an encoder stage feeds both the skip connection and the stage's max pool.  One thread owns a
// 2x2 window (8 channels): the forward writes the four activations and their maximum in one pass; the backward routes
// the pooled gradient to the window's first maximum (scan order, compared at storage precision exactly as a separate
// pool kernel would see the stored activations), adds the skip gradient and continues with the norm backward, so the
// full-resolution gradient is never written and re-read in between.
struct Window { int64_t p00; int64_t q; };   // first pixel of the window, window index (= pooled pixel)
__device__ __forceinline__ int64_t window_pixel(int64_t q, int H, int W) {
    const int Wo = W / 2, Ho = H / 2;
    const int ox = (int)(q % Wo);
    const int64_t t = q / Wo;
    const int oy = (int)(t % Ho);
    const int64_t n = t / Ho;
    return (n * H + 2 * oy) * W + 2 * ox;
}

template <typename T, bool BIG>
__global__ __launch_bounds__(kThreads) void norm_pool_apply_kernel(const T* __restrict__ x, int64_t ldx, int64_t Q, int H, int W,
                                                                   int C, const float* __restrict__ stats,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, T* __restrict__ y,
                                                                   int64_t ldy, T* __restrict__ pool, int64_t ldp) {
    const int cv = C / 8, rows = kThreads / cv;
    const int v = threadIdx.x % cv, r = threadIdx.x / cv;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        const float mean = stats[c * 2], rstd = stats[c * 2 + 1];
        sc[j] = rstd * (gamma ? gamma[c] : 1.f);
        sh[j] = (beta ? beta[c] : 0.f) - mean * sc[j];
    }
    for (int64_t q = (int64_t)blockIdx.x * rows + r; q < Q && r < rows; q += (int64_t)gridDim.x * rows) {
        const int64_t p00 = window_pixel(q, H, W);
        const int64_t off[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
        float a[4][8], m[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) ld8<T, BIG>(x + off[k] * ldx + v * 8, a[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[k][j] = fmaxf(a[k][j] * sc[j] + sh[j], 0.f);
                m[j] = k == 0 ? a[k][j] : fmaxf(m[j], a[k][j]);
            }
            st8<T, BIG>(y + off[k] * ldy + v * 8, a[k]);
        }
        st8<T, BIG>(pool + q * ldp + v * 8, m);
    }
}

// gradient reaching the norm output at the four pixels of window q (masked by the ReLU), and the normalised inputs
template <typename T, bool BIG>
__device__ __forceinline__ void window_grad(const T* __restrict__ da, int64_t ld_da, const T* __restrict__ dp, int64_t ld_dp,
                                            const T* __restrict__ x, int64_t ldx, int64_t q, const int64_t (&off)[4], int v,
                                            const float (&mean)[8], const float (&rstd)[8], const float (&ga)[8],
                                            const float (&be)[8], float (&g)[4][8], float (&xh)[4][8]) {
    float pre[4][8], gp[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) ld8<T, BIG>(x + off[k] * ldx + v * 8, pre[k]);
    ld8<T, BIG>(dp + q * ld_dp + v * 8, gp);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (da) ld8<T, BIG>(da + off[k] * ld_da + v * 8, g[k]);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[k][j] = 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float sc = rstd[j] * ga[j], sh = be[j] - mean[j] * sc;
        float act[4];
        int arg = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xh[k][j] = (pre[k][j] - mean[j]) * rstd[j];
            pre[k][j] = pre[k][j] * sc + sh;                                   // same expression as the forward
            act[k] = Elem<T>::to_f(Elem<T>::from_f(fmaxf(pre[k][j], 0.f)));   // the stored activation
            if (k > 0 && act[k] > act[arg]) arg = k;                           // first maximum in scan order wins
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = g[k][j] + (k == arg ? gp[j] : 0.f);
            g[k][j] = pre[k][j] > 0.f ? t : 0.f;
        }
    }
}

template <typename T, bool BIG>
__global__ __launch_bounds__(kThreads) void norm_pool_bwd_partial_kernel(const T* __restrict__ da, int64_t ld_da,
                                                                         const T* __restrict__ dp, int64_t ld_dp,
                                                                         const T* __restrict__ x, int64_t ldx, int64_t Q, int H,
                                                                         int W, int C, const float* __restrict__ stats,
                                                                         const float* __restrict__ gamma,
                                                                         const float* __restrict__ beta, int bpg,
                                                                         float* __restrict__ ws) {
    __shared__ float red[kThreads][17];
    const int cv = C / 8, rows = kThreads / cv;
    const int v = threadIdx.x % cv, r = threadIdx.x / cv;
    const int b = blockIdx.x;
    float mean[8], rstd[8], ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        mean[j] = stats[c * 2];
        rstd[j] = stats[c * 2 + 1];
        ga[j] = gamma ? gamma[c] : 1.f;
        be[j] = beta ? beta[c] : 0.f;
    }
    const int64_t per = (Q + bpg - 1) / bpg;
    const int64_t lo = (int64_t)b * per, hi = (lo + per < Q) ? lo + per : Q;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    for (int64_t q = lo + r; q < hi && r < rows; q += rows) {
        const int64_t p00 = window_pixel(q, H, W);
        const int64_t off[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
        float g[4][8], xh[4][8];
        window_grad<T, BIG>(da, ld_da, dp, ld_dp, x, ldx, q, off, v, mean, rstd, ga, be, g, xh);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += g[k][j]; s2[j] += g[k][j] * xh[k][j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = s1[j]; red[threadIdx.x][8 + j] = s2[j]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += kThreads) {
        const int k = t / C, c = t % C;
        float acc = 0.f;
        for (int rr = 0; rr < rows; ++rr) acc += red[rr * cv + c / 8][k * 8 + (c & 7)];
        ws[((int64_t)b * 2 + k) * C + c] = acc;
    }
}

template <typename T, bool BIG>
__global__ __launch_bounds__(kThreads) void norm_pool_bwd_dx_kernel(const T* __restrict__ da, int64_t ld_da,
                                                                    const T* __restrict__ dp, int64_t ld_dp,
                                                                    const T* __restrict__ x, int64_t ldx, int64_t Q, int H, int W,
                                                                    int C, int64_t M, const float* __restrict__ stats,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int frozen,
                                                                    const float* __restrict__ red, T* __restrict__ dx,
                                                                    int64_t ld_dx) {
    const int cv = C / 8, rows = kThreads / cv;
    const int v = threadIdx.x % cv, r = threadIdx.x / cv;
    float mean[8], rstd[8], ga[8], be[8], m1[8], m2[8];
    const float invM = 1.f / (float)M;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        mean[j] = stats[c * 2];
        rstd[j] = stats[c * 2 + 1];
        ga[j] = gamma ? gamma[c] : 1.f;
        be[j] = beta ? beta[c] : 0.f;
        m1[j] = frozen ? 0.f : red[c * 2] * invM;
        m2[j] = frozen ? 0.f : red[c * 2 + 1] * invM;
    }
    for (int64_t q = (int64_t)blockIdx.x * rows + r; q < Q && r < rows; q += (int64_t)gridDim.x * rows) {
        const int64_t p00 = window_pixel(q, H, W);
        const int64_t off[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
        float g[4][8], xh[4][8];
        window_grad<T, BIG>(da, ld_da, dp, ld_dp, x, ldx, q, off, v, mean, rstd, ga, be, g, xh);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[k][j] = rstd[j] * ga[j] * (g[k][j] - m1[j] - xh[k][j] * m2[j]);
            st8<T, BIG>(dx + off[k] * ld_dx + v * 8, g[k]);
        }
    }
}

// ------------------------------------------------------------------------------ 2x2 max pool
template <typename T>
__global__ __launch_bounds__(kThreads) void maxpool_fwd_kernel(const T* __restrict__ x, int64_t ldx, int64_t N, int H, int W,
                                                               int C, T* __restrict__ y, int64_t ldy) {
    const int cv = C / 8, Ho = H / 2, Wo = W / 2;
    const int64_t items = N * Ho * Wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kThreads) {
        const int v = (int)(i % cv);
        int64_t p = i / cv;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int64_t n = p / Ho;
        const T* s = x + ((n * H + 2 * oy) * W + 2 * ox) * ldx + v * 8;
        float a[8], b[8], c[8], d[8];
        Elem<T>::load8(s, a);
        Elem<T>::load8(s + ldx, b);
        Elem<T>::load8(s + (int64_t)W * ldx, c);
        Elem<T>::load8(s + (int64_t)W * ldx + ldx, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(c[j], d[j]));
        Elem<T>::store8(y + ((n * Ho + oy) * Wo + ox) * ldy + v * 8, a);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void maxpool_bwd_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ dy,
                                                               int64_t ld_dy, int64_t N, int H, int W, int C,
                                                               T* __restrict__ dx, int64_t ld_dx, int accumulate) {
    const int cv = C / 8, Ho = H / 2, Wo = W / 2;
    const int64_t items = N * Ho * Wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kThreads) {
        const int v = (int)(i % cv);
        int64_t p = i / cv;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int64_t n = p / Ho;
        const int64_t o00 = ((n * H + 2 * oy) * W + 2 * ox);
        const int64_t off[4] = {o00, o00 + 1, o00 + W, o00 + W + 1};
        float in[4][8], g[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) Elem<T>::load8(x + off[k] * ldx + v * 8, in[k]);
        Elem<T>::load8(dy + ((n * Ho + oy) * Wo + ox) * ld_dy + v * 8, g);
        int arg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float best = in[0][j];
            arg[j] = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (in[k][j] > best) { best = in[k][j]; arg[j] = k; }   // first maximum in scan order wins
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o[8];
            if (accumulate) Elem<T>::load8(dx + off[k] * ld_dx + v * 8, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float r = arg[j] == k ? g[j] : 0.f;
                o[j] = accumulate ? o[j] + r : r;
            }
            Elem<T>::store8(dx + off[k] * ld_dx + v * 8, o);
        }
    }
}

// ------------------------------------------------------------------------------ bilinear x2
// Work decomposition of the resampling kernels: a block owns a TY x 8 patch of pixels times a group of
// CG channel vectors (TY*8*CG = 256 threads), so that the overlapping neighbourhoods of adjacent pixels are served
// by the block's own L1 instead of being fetched once per XCD (consecutive blocks run on different XCDs, whose L2s
// do not share lines).
struct PatchItem { int64_t n; int h, w, v; bool ok; };
// Blocks are dealt to the 8 XCDs round-robin and every XCD has its own L2: renumber so that each XCD walks one contiguous
// eighth of the block list, and patches whose neighbourhoods overlap meet in one L2.
__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned b, unsigned G) {
    const unsigned base = G >> 3, rem = G & 7, x = b & 7;
    return x * base + (x < rem ? x : rem) + (b >> 3);
}
__host__ __device__ inline int patch_cg(int cv) { return cv % 8 == 0 ? 8 : (cv % 4 == 0 ? 4 : (cv % 2 == 0 ? 2 : 1)); }
static inline int64_t patch_blocks(int64_t N, int64_t H, int64_t W, int64_t C) {
    const int cv = (int)(C / 8), cg = patch_cg(cv), ty = kThreads / (8 * cg);
    return N * ((H + ty - 1) / ty) * ((W + 7) / 8) * (cv / cg);
}
__device__ __forceinline__ PatchItem patch_item(int64_t block, int tid, int H, int W, int cv) {
    const int cg = patch_cg(cv), ty = kThreads / (8 * cg);
    const int ncg = cv / cg, tiles_x = (W + 7) / 8, tiles_y = (H + ty - 1) / ty;
    const int cgi = (int)(block % ncg); block /= ncg;
    const int tx = (int)(block % tiles_x); block /= tiles_x;
    const int tyi = (int)(block % tiles_y);
    PatchItem it;
    it.n = block / tiles_y;
    it.v = cgi * cg + tid % cg;
    it.w = tx * 8 + (tid / cg) % 8;
    it.h = tyi * ty + tid / (cg * 8);
    it.ok = it.h < H && it.w < W;
    return it;
}


struct Lerp { int i0, i1; float w0, w1; };

// source taps of output index `o` (size 2*in) exactly as ATen's upsample_bilinear2d forms them
__device__ __forceinline__ Lerp lerp_taps(int o, int in, bool align) {
    float src;
    if (align) {
        const float scale = in > 1 ? (float)(in - 1) / (float)(2 * in - 1) : 0.f;
        src = scale * (float)o;
    } else {
        src = 0.5f * ((float)o + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
    }
    Lerp l;
    l.i0 = (int)src;
    l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

// ------------------------------------------------------------------------------ bilinear resize by any factor
// align_corners=False with an explicit scale factor: ATen maps output index o to the source coordinate
// max(ratio * (o + 0.5) - 0.5, 0) with ratio = 1 / scale_factor (upsample_bilinear2d), taps i0 = floor, i1 = min(i0 + 1, in - 1).
__device__ __forceinline__ Lerp lerp_taps_ratio(int o, int in, float ratio) {
    float src = ratio * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Lerp l;
    l.i0 = (int)src;
    if (l.i0 > in - 1) l.i0 = in - 1;
    l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void resize_fwd_kernel(const T* __restrict__ x, int64_t ldx, int H, int W, int C, int Ho, int Wo,
                                                              float rh, float rw, const float* __restrict__ stats, int relu,
                                                              T* __restrict__ y, int64_t ldy) {
    const int cv = C / 8;
    const PatchItem it = patch_item(blockIdx.x, threadIdx.x, Ho, Wo, cv);   // one OUTPUT pixel (8 channels) per thread
    if (!it.ok) return;
    const int oy = it.h, ox = it.w, v = it.v;
    const int64_t n = it.n;
    const Lerp ly = lerp_taps_ratio(oy, H, rh), lx = lerp_taps_ratio(ox, W, rw);
    const T* b = x + n * H * W * ldx + v * 8;
    float a00[8], a01[8], a10[8], a11[8], o[8];
    Elem<T>::load8(b + ((int64_t)ly.i0 * W + lx.i0) * ldx, a00);
    Elem<T>::load8(b + ((int64_t)ly.i0 * W + lx.i1) * ldx, a01);
    Elem<T>::load8(b + ((int64_t)ly.i1 * W + lx.i0) * ldx, a10);
    Elem<T>::load8(b + ((int64_t)ly.i1 * W + lx.i1) * ldx, a11);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float val = ly.w0 * (lx.w0 * a00[j] + lx.w1 * a01[j]) + ly.w1 * (lx.w0 * a10[j] + lx.w1 * a11[j]);
        if (stats) {
            const int c = v * 8 + j;
            val = (val - stats[(n * C + c) * 2]) * stats[(n * C + c) * 2 + 1];
        }
        o[j] = relu ? fmaxf(val, 0.f) : val;
    }
    Elem<T>::store8(y + ((n * Ho + oy) * Wo + ox) * ldy + v * 8, o);
}

// Adjoint in gather form (deterministic): input index h is a tap of the outputs whose source coordinate lies in
// (h - 1, h + 1), i.e. o in ((h - 0.5) / ratio - 0.5, (h + 1.5) / ratio - 0.5); the clamped first and last outputs are
// covered by widening that window by one on both sides and testing every candidate's taps.
__device__ __forceinline__ void resize_window(int h, int out, float ratio, int& lo, int& hi) {
    lo = (int)floorf(((float)h - 0.5f) / ratio - 0.5f) - 1;
    hi = (int)ceilf(((float)h + 1.5f) / ratio - 0.5f) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void resize_bwd_kernel(const T* __restrict__ dy, int64_t ld_dy, int H, int W, int C, int Ho, int Wo,
                                                              float rh, float rw, T* __restrict__ dx, int64_t ld_dx, int accumulate) {
    const int cv = C / 8;
    const PatchItem it = patch_item(blockIdx.x, threadIdx.x, H, W, cv);     // one INPUT pixel (8 channels) per thread
    if (!it.ok) return;
    const int h = it.h, w = it.w, v = it.v;
    const int64_t n = it.n;
    int ylo, yhi, xlo, xhi;
    resize_window(h, Ho, rh, ylo, yhi);
    resize_window(w, Wo, rw, xlo, xhi);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
        const Lerp l = lerp_taps_ratio(oy, H, rh);
        const float wy = (l.i0 == h ? l.w0 : 0.f) + (l.i1 == h ? l.w1 : 0.f);
        if (wy == 0.f) continue;
        float row[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = 0.f;
        for (int ox = xlo; ox <= xhi; ++ox) {
            const Lerp m = lerp_taps_ratio(ox, W, rw);
            const float wx = (m.i0 == w ? m.w0 : 0.f) + (m.i1 == w ? m.w1 : 0.f);
            if (wx == 0.f) continue;
            float d[8];
            Elem<T>::load8(dy + ((n * Ho + oy) * Wo + ox) * ld_dy + v * 8, d);
#pragma unroll
            for (int j = 0; j < 8; ++j) row[j] += wx * d[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wy * row[j];
    }
    T* dst = dx + ((n * H + h) * W + w) * ld_dx + v * 8;
    if (accumulate) {
        float o[8];
        Elem<T>::load8(dst, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += o[j];
    }
    Elem<T>::store8(dst, acc);
}

// Block-wise x2 kernels.  One thread owns one INPUT pixel (8 channels).
//
// forward (either corner convention): the four outputs (2h + dy, 2w + dx) of input pixel (h, w) interpolate inside its 3x3
// neighbourhood -- align_corners=False: rows (h-1, h) and (h, h+1) with weights (0.25, 0.75) and (0.75, 0.25);
// align_corners=True: source coordinate o*(H-1)/(2H-1), which lies in (h-1, h] for o = 2h and in (h, h+1) for o = 2h+1.
// The block stages the (TY+2) x 10 input pixels of its TY x 8 patch in LDS (rows and columns clamped at the image border,
// as the interpolation clamps them), so every input vector is fetched once per block; the taps of every output are
// looked up there by the indices ATen's formula gives, and combined in ATen's order.  Four stores per thread keep
// four times the bytes of a one-output-per-thread kernel in flight.
constexpr int kUpInSlots = 480;      // max over channel groups cg of (TY + 2) * 10 * cg with TY = 256 / (8*cg): cg = 8
template <typename T>
__global__ __launch_bounds__(kThreads) void upsample_fwd_block_kernel(const T* __restrict__ x, int64_t ldx, int H, int W, int C,
                                                                      const float* __restrict__ stats, int relu, int align,
                                                                      T* __restrict__ y, int64_t ldy) {
    struct alignas(16) Pack8 { T e[8]; };
    __shared__ Pack8 tile[kUpInSlots];
    const int cv = C / 8, Ho = 2 * H, Wo = 2 * W;
    const int cg = patch_cg(cv), ty = kThreads / (8 * cg);
    const PatchItem it = patch_item(xcd_contiguous_block(blockIdx.x, gridDim.x), threadIdx.x, H, W, cv);
    const int lh = threadIdx.x / (cg * 8), lw = (threadIdx.x / cg) % 8, lv = threadIdx.x % cg;
    const int h0 = it.h - lh, w0 = it.w - lw, v0 = it.v - lv;
    const int64_t n = it.n;
    constexpr int RW = 10;
    const T* b = x + n * H * W * ldx;
    const int slots = (ty + 2) * RW * cg;
    for (int s = threadIdx.x; s < slots; s += kThreads) {
        const int vv = s % cg, px = s / cg;
        const int ry = px / RW, rx = px - ry * RW;
        int iy = h0 - 1 + ry, ix = w0 - 1 + rx;
        iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
        ix = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
        tile[s] = *reinterpret_cast<const Pack8*>(b + ((int64_t)iy * W + ix) * ldx + (v0 + vv) * 8);
    }
    __syncthreads();
    if (!it.ok) return;
    const int h = it.h, w = it.w, v = it.v;
    float mean[8], rstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        mean[j] = stats ? stats[(n * C + c) * 2] : 0.f;
        rstd[j] = stats ? stats[(n * C + c) * 2 + 1] : 1.f;
    }
    Lerp lys[2], lxs[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) { lys[d] = lerp_taps(2 * h + d, H, align != 0); lxs[d] = lerp_taps(2 * w + d, W, align != 0); }
    auto tap = [&](int iy, int ix, float (&a)[8]) {
        const Pack8 t = tile[((iy - h0 + 1) * RW + ix - w0 + 1) * cg + lv];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = Elem<T>::to_f(t.e[j]);
    };
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const Lerp ly = lys[dy], lx = lxs[dx];
            float a00[8], a01[8], a10[8], a11[8], o[8];
            tap(ly.i0, lx.i0, a00);
            tap(ly.i0, lx.i1, a01);
            tap(ly.i1, lx.i0, a10);
            tap(ly.i1, lx.i1, a11);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float val = ly.w0 * (lx.w0 * a00[j] + lx.w1 * a01[j]) + ly.w1 * (lx.w0 * a10[j] + lx.w1 * a11[j]);
                if (stats) val = (val - mean[j]) * rstd[j];
                o[j] = relu ? fmaxf(val, 0.f) : val;
            }
            Elem<T>::store8(y + ((n * Ho + 2 * h + dy) * Wo + 2 * w + dx) * ldy + v * 8, o);
        }
}

// forward, align_corners=False: the same staging with the taps and weights fixed at compile time (output rows 2h and 2h+1
// read input rows (h-1, h) and (h, h+1) with weights (0.25, 0.75) and (0.75, 0.25); output 0 puts all its weight on input 0,
// and past the last row the clamped neighbour is the row itself).
template <typename T>
__global__ __launch_bounds__(kThreads) void upsample_fwd_half_kernel(const T* __restrict__ x, int64_t ldx, int H, int W, int C,
                                                                     const float* __restrict__ stats, int relu,
                                                                     T* __restrict__ y, int64_t ldy) {
    struct alignas(16) Pack8 { T e[8]; };
    __shared__ Pack8 tile[kUpInSlots];
    const int cv = C / 8, Ho = 2 * H, Wo = 2 * W;
    const int cg = patch_cg(cv), ty = kThreads / (8 * cg);
    const PatchItem it = patch_item(xcd_contiguous_block(blockIdx.x, gridDim.x), threadIdx.x, H, W, cv);
    const int lh = threadIdx.x / (cg * 8), lw = (threadIdx.x / cg) % 8, lv = threadIdx.x % cg;
    const int h0 = it.h - lh, w0 = it.w - lw, v0 = it.v - lv;
    const int64_t n = it.n;
    constexpr int RW = 10;
    const T* b = x + n * H * W * ldx;
    const int slots = (ty + 2) * RW * cg;
    for (int s = threadIdx.x; s < slots; s += kThreads) {
        const int vv = s % cg, px = s / cg;
        const int ry = px / RW, rx = px - ry * RW;
        int iy = h0 - 1 + ry, ix = w0 - 1 + rx;
        iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
        ix = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
        tile[s] = *reinterpret_cast<const Pack8*>(b + ((int64_t)iy * W + ix) * ldx + (v0 + vv) * 8);
    }
    __syncthreads();
    if (!it.ok) return;
    const int h = it.h, w = it.w, v = it.v;
    float a[3][3][8];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const Pack8 t = tile[((lh + r) * RW + lw + c) * cg + lv];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[r][c][j] = Elem<T>::to_f(t.e[j]);
        }
    float mean[8], rstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        mean[j] = stats ? stats[(n * C + c) * 2] : 0.f;
        rstd[j] = stats ? stats[(n * C + c) * 2 + 1] : 1.f;
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            // taps (dy, dy+1) x (dx, dx+1) of the neighbourhood; weight of the second tap
            const float wy1 = dy == 0 ? (h > 0 ? 0.75f : 0.f) : 0.25f, wy0 = 1.f - wy1;
            const float wx1 = dx == 0 ? (w > 0 ? 0.75f : 0.f) : 0.25f, wx0 = 1.f - wx1;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float val = wy0 * (wx0 * a[dy][dx][j] + wx1 * a[dy][dx + 1][j]) + wy1 * (wx0 * a[dy + 1][dx][j] + wx1 * a[dy + 1][dx + 1][j]);
                if (stats) val = (val - mean[j]) * rstd[j];
                o[j] = relu ? fmaxf(val, 0.f) : val;
            }
            Elem<T>::store8(y + ((n * Ho + 2 * h + dy) * Wo + 2 * w + dx) * ldy + v * 8, o);
        }
}

// backward, align_corners=False (the decoder blocks; fixed taps: output rows 2h and 2h+1 read input rows (h-1, h) and (h, h+1) with
// weights (0.25, 0.75) and (0.75, 0.25), clamped at the border): the 4x4 output neighbourhood [2h-1, 2h+2] x [2w-1, 2w+2] with
// separable weights.  Neighbouring input pixels
// share three quarters of that neighbourhood, so the block first stages the (2*TY+2) x 18 output pixels its TY x 8 patch needs
// in LDS -- every gradient vector is fetched once per block (already masked by the ReLU when asked) -- and each thread
// then gathers its sixteen taps from there.  Positions outside the image hold zeros; their weights are zero as well.
constexpr int kUpTileSlots = 1440;   // max over channel groups cg of (2*TY + 2) * 18 * cg with TY = 256 / (8*cg): cg = 8
template <typename T>
__global__ __launch_bounds__(kThreads) void upsample_bwd_half_kernel(const T* __restrict__ dy, int64_t ld_dy,
                                                                     const T* __restrict__ y, int64_t ldy, int H, int W, int C,
                                                                     int relu, T* __restrict__ dx, int64_t ld_dx) {
    struct alignas(16) Pack8 { T e[8]; };
    __shared__ Pack8 tile[kUpTileSlots];
    const int cv = C / 8, Ho = 2 * H, Wo = 2 * W;
    const int cg = patch_cg(cv), ty = kThreads / (8 * cg);
    const PatchItem it = patch_item(xcd_contiguous_block(blockIdx.x, gridDim.x), threadIdx.x, H, W, cv);
    const int lh = threadIdx.x / (cg * 8), lw = (threadIdx.x / cg) % 8, lv = threadIdx.x % cg;
    const int h0 = it.h - lh, w0 = it.w - lw, v0 = it.v - lv;      // the patch's first pixel and channel vector
    const int64_t n = it.n;
    constexpr int RW = 18;
    const int slots = (2 * ty + 2) * RW * cg;
    for (int s = threadIdx.x; s < slots; s += kThreads) {
        const int vv = s % cg, px = s / cg;
        const int ry = px / RW, rx = px - ry * RW;
        const int oy = 2 * h0 - 1 + ry, ox = 2 * w0 - 1 + rx;
        Pack8 val;
#pragma unroll
        for (int j = 0; j < 8; ++j) val.e[j] = Elem<T>::from_f(0.f);
        if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) {
            const int64_t op = (n * Ho + oy) * Wo + ox;
            val = *reinterpret_cast<const Pack8*>(dy + op * ld_dy + (v0 + vv) * 8);
            if (relu) {
                const Pack8 f = *reinterpret_cast<const Pack8*>(y + op * ldy + (v0 + vv) * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) val.e[j] = Elem<T>::to_f(f.e[j]) > 0.f ? val.e[j] : Elem<T>::from_f(0.f);
            }
        }
        tile[s] = val;
    }
    __syncthreads();
    if (!it.ok) return;
    const int h = it.h, w = it.w, v = it.v;
    const float wy[4] = {h > 0 ? 0.25f : 0.f, h > 0 ? 0.75f : 1.f, h < H - 1 ? 0.75f : 1.f, h < H - 1 ? 0.25f : 0.f};
    const float wx[4] = {w > 0 ? 0.25f : 0.f, w > 0 ? 0.75f : 1.f, w < W - 1 ? 0.75f : 1.f, w < W - 1 ? 0.25f : 0.f};
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        // separable: weighted sum along the row first, then one multiply-add per row
        float row[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = 0.f;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const Pack8 d = tile[((2 * lh + ky) * RW + 2 * lw + kx) * cg + lv];
#pragma unroll
            for (int j = 0; j < 8; ++j) row[j] += wx[kx] * Elem<T>::to_f(d.e[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wy[ky] * row[j];
    }
    Elem<T>::store8(dx + ((n * H + h) * W + w) * ld_dx + v * 8, acc);
}

// backward, either corner convention (used for align_corners=True): input pixel h is a tap of the outputs whose source
// coordinate lies in (h-1, h+1); for x2 up-sampling with source coordinate o*(H-1)/(2H-1) those are o in [2h-2, 2h+3]
// (the 4 of align_corners=False lie inside).  The block stages the (2*TY+4) x 20 output pixels around its TY x 8 patch in
// LDS once (masked by the ReLU when asked, zero outside the image); every thread then gathers its 6 x 6 window from LDS with
// the weights ATen's formula gives (rows of zero weight are skipped).
constexpr int kUpBlockSlots = 1920;  // max over channel groups cg of (2*TY + 4) * 20 * cg with TY = 256 / (8*cg): cg = 8
template <typename T>
__global__ __launch_bounds__(kThreads) void upsample_bwd_block_kernel(const T* __restrict__ dy, int64_t ld_dy,
                                                                      const T* __restrict__ y, int64_t ldy, int H, int W, int C,
                                                                      int relu, int align, T* __restrict__ dx, int64_t ld_dx) {
    struct alignas(16) Pack8 { T e[8]; };
    __shared__ Pack8 tile[kUpBlockSlots];
    const int cv = C / 8, Ho = 2 * H, Wo = 2 * W;
    const int cg = patch_cg(cv), ty = kThreads / (8 * cg);
    const PatchItem it = patch_item(xcd_contiguous_block(blockIdx.x, gridDim.x), threadIdx.x, H, W, cv);
    const int lh = threadIdx.x / (cg * 8), lw = (threadIdx.x / cg) % 8, lv = threadIdx.x % cg;
    const int h0 = it.h - lh, w0 = it.w - lw, v0 = it.v - lv;
    const int64_t n = it.n;
    constexpr int RW = 20;
    const int slots = (2 * ty + 4) * RW * cg;
    for (int s = threadIdx.x; s < slots; s += kThreads) {
        const int vv = s % cg, px = s / cg;
        const int ry = px / RW, rx = px - ry * RW;
        const int oy = 2 * h0 - 2 + ry, ox = 2 * w0 - 2 + rx;
        Pack8 val;
#pragma unroll
        for (int j = 0; j < 8; ++j) val.e[j] = Elem<T>::from_f(0.f);
        if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) {
            const int64_t op = (n * Ho + oy) * Wo + ox;
            val = *reinterpret_cast<const Pack8*>(dy + op * ld_dy + (v0 + vv) * 8);
            if (relu) {
                const Pack8 f = *reinterpret_cast<const Pack8*>(y + op * ldy + (v0 + vv) * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) val.e[j] = Elem<T>::to_f(f.e[j]) > 0.f ? val.e[j] : Elem<T>::from_f(0.f);
            }
        }
        tile[s] = val;
    }
    __syncthreads();
    if (!it.ok) return;
    const int h = it.h, w = it.w, v = it.v;
    float wy[6], wx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int oy = 2 * h - 2 + k, ox = 2 * w - 2 + k;
        wy[k] = 0.f; wx[k] = 0.f;
        if (oy >= 0 && oy < Ho) {
            const Lerp l = lerp_taps(oy, H, align != 0);
            wy[k] = (l.i0 == h ? l.w0 : 0.f) + (l.i1 == h ? l.w1 : 0.f);
        }
        if (ox >= 0 && ox < Wo) {
            const Lerp l = lerp_taps(ox, W, align != 0);
            wx[k] = (l.i0 == w ? l.w0 : 0.f) + (l.i1 == w ? l.w1 : 0.f);
        }
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
        if (wy[ky] == 0.f) continue;
        float row[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = 0.f;
#pragma unroll
        for (int kx = 0; kx < 6; ++kx) {
            const Pack8 t = tile[((2 * lh + ky) * RW + 2 * lw + kx) * cg + lv];
#pragma unroll
            for (int j = 0; j < 8; ++j) row[j] += wx[kx] * Elem<T>::to_f(t.e[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wy[ky] * row[j];
    }
    Elem<T>::store8(dx + ((n * H + h) * W + w) * ld_dx + v * 8, acc);
}

// ------------------------------------------------------------------------------ slices
template <typename T>
__global__ __launch_bounds__(kThreads) void copy_slice_kernel(const T* __restrict__ src, int64_t ld_src, T* __restrict__ dst,
                                                              int64_t ld_dst, int64_t P, int C, int accumulate) {
    const int cv = C / 8;
    const int64_t items = P * cv;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kThreads) {
        const int v = (int)(i % cv);
        const int64_t p = i / cv;
        float a[8];
        if (src) {
            Elem<T>::load8(src + p * ld_src + v * 8, a);
        } else {                                         // no source: fill the slice with zeros
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = 0.f;
        }
        if (accumulate) {
            float b[8];
            Elem<T>::load8(dst + p * ld_dst + v * 8, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += b[j];
        }
        Elem<T>::store8(dst + p * ld_dst + v * 8, a);
    }
}

// 2x2 blocks of pixels <-> channels: wide[n, oh, ow, (b*2 + e)*C + c] = x[n, 2*oh + b, 2*ow + e, c].
// With the weight of a Conv2d(kernel 2, stride 2) read as [Cout][(b, e, c)] the strided convolution is a 1x1 convolution
// over `wide`; `to_wide` = 0 runs the permutation backwards (its adjoint: the data gradient), optionally adding.
template <typename T>
__global__ __launch_bounds__(kThreads) void block2_kernel(T* __restrict__ x, int64_t ldx, T* __restrict__ wide, int64_t ldw, int64_t N,
                                                          int Ho, int Wo, int C, int to_wide, int accumulate) {
    const int cv = C / 8;
    const int64_t items = N * Ho * Wo * 4 * cv;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kThreads) {
        const int v = (int)(i % cv);
        int64_t p = i / cv;
        const int tap = (int)(p & 3); p >>= 2;                      // p: output pixel (n, oh, ow)
        const int ow = (int)(p % Wo);
        const int64_t r = p / Wo;                                   // n*Ho + oh
        T* xs = x + ((2 * r + (tap >> 1)) * (2 * Wo) + 2 * ow + (tap & 1)) * ldx + v * 8;
        T* ws = wide + p * ldw + tap * C + v * 8;
        float a[8];
        if (to_wide) {
            Elem<T>::load8(xs, a);
            Elem<T>::store8(ws, a);
        } else {
            Elem<T>::load8(ws, a);
            if (accumulate) {
                float b[8];
                Elem<T>::load8(xs, b);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += b[j];
            }
            Elem<T>::store8(xs, a);
        }
    }
}

// dx = dy where the saved activation y is positive, else 0 (backward of a ReLU that follows a layer without norm)
template <typename T>
__global__ __launch_bounds__(kThreads) void relu_bwd_kernel(const T* __restrict__ dy, int64_t ld_dy, const T* __restrict__ y,
                                                            int64_t ldy, int64_t P, int C, T* __restrict__ dx, int64_t ld_dx) {
    const int cv = C / 8;
    const int64_t items = P * cv;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kThreads) {
        const int v = (int)(i % cv);
        const int64_t p = i / cv;
        float d[8], f[8];
        Elem<T>::load8(dy + p * ld_dy + v * 8, d);
        Elem<T>::load8(y + p * ldy + v * 8, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = f[j] > 0.f ? d[j] : 0.f;
        Elem<T>::store8(dx + p * ld_dx + v * 8, d);
    }
}

// ------------------------------------------------------------------------------ 1x1 head
constexpr int kHeadMaxOut = 8;

// One thread per pixel computes all Cout dot products and writes them class-plane by class-plane (consecutive lanes =
// consecutive pixels: coalesced fp32 stores).  The pixel rows a block works on are first copied to LDS with coalesced
// 16-byte loads (TP pixels per round; rows padded by 16 bytes so that one-row-per-lane 16-byte reads are conflict-free):
// a thread walking its own pixel row in global memory would touch a new 128-byte line per lane and load.
// TP == 0: rows too long for LDS, threads read global memory directly.
template <typename T>
__global__ __launch_bounds__(kThreads) void head_fwd_kernel(const T* __restrict__ x, int64_t ld, int Cin, int64_t N,
                                                            int64_t HW, const float* __restrict__ w,
                                                            const float* __restrict__ bias, int Cout,
                                                            float* __restrict__ logits, int TP) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];   // [Cout][Cin] + [Cout], then the pixel rows
    struct alignas(16) Pack8 { T e[8]; };
    for (int i = threadIdx.x; i < Cout * Cin; i += kThreads) wsm[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += kThreads) wsm[Cout * Cin + i] = bias ? bias[i] : 0.f;
    const int64_t P = N * HW;
    auto finish = [&](int64_t p, const T* row) {
        float acc[kHeadMaxOut];
#pragma unroll
        for (int c = 0; c < kHeadMaxOut; ++c) acc[c] = c < Cout ? wsm[Cout * Cin + c] : 0.f;
        for (int k = 0; k < Cin; k += 8) {
            float a[8];
            Elem<T>::load8(row + k, a);
#pragma unroll
            for (int c = 0; c < kHeadMaxOut; ++c)
                if (c < Cout) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[c] += a[j] * wsm[c * Cin + k + j];
                }
        }
        const int64_t n = p / HW, hw = p % HW;
#pragma unroll
        for (int c = 0; c < kHeadMaxOut; ++c)
            if (c < Cout) logits[(n * Cout + c) * HW + hw] = acc[c];
    };
    if (TP == 0) {
        __syncthreads();
        for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < P; p += (int64_t)gridDim.x * kThreads) finish(p, x + p * ld);
        return;
    }
    const int cv = Cin / 8, rowe = Cin + 16 / (int)sizeof(T);      // padded row length in elements
    T* rows = reinterpret_cast<T*>(wsm + ((Cout * Cin + Cout + 3) & ~3));
    for (int64_t p0 = (int64_t)blockIdx.x * TP; p0 < P; p0 += (int64_t)gridDim.x * TP) {
        __syncthreads();                                            // weights written / previous rows consumed
        for (int s = threadIdx.x; s < TP * cv; s += kThreads) {
            const int px = s / cv, v = s - px * cv;
            if (p0 + px < P) *reinterpret_cast<Pack8*>(rows + px * rowe + v * 8) = *reinterpret_cast<const Pack8*>(x + (p0 + px) * ld + v * 8);
        }
        __syncthreads();
        if (threadIdx.x < TP && p0 + threadIdx.x < P) finish(p0 + threadIdx.x, rows + threadIdx.x * rowe);
    }
}

// backward: item = (pixel, 8-channel vector).  dx = sum_c dl[c]*w[c][ci] (masked by x>0 when the head's
// input is a ReLU output); per-thread dw[c][8] accumulators reduced over lanes sharing the channel vector.
// partial layout: ws[b*(Cout*Cin + Cout) + ...] = dw[c][ci] then db[c]
template <typename T, int COUT>
__global__ __launch_bounds__(kThreads) void head_bwd_kernel(const T* __restrict__ x, int64_t ld, int Cin, int64_t N,
                                                            int64_t HW, const float* __restrict__ w,
                                                            const float* __restrict__ dlogits, T* __restrict__ dx,
                                                            int64_t ld_dx, int relu_mask, float* __restrict__ ws) {
    __shared__ float red[kThreads / 64][8][COUT * 8 + 1];   // [wave][channel vector][c*8+j], padded against bank conflicts
    __shared__ float dbs[kThreads / 64][COUT];
    const int cv = Cin / 8;                      // 1, 2, 4 or 8 (Cin 8 .. 64): divides 64
    const int v = threadIdx.x % cv;
    const int64_t P = N * HW, items = P * cv;
    float wr[COUT][8], acc[COUT][8], db[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        db[c] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { wr[c][j] = w[c * Cin + v * 8 + j]; acc[c][j] = 0.f; }
    }
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kThreads) {
        const int64_t p = i / cv, n = p / HW, hw = p % HW;
        float a[8], o[8], dl[COUT];
        Elem<T>::load8(x + p * ld + v * 8, a);
#pragma unroll
        for (int c = 0; c < COUT; ++c) dl[c] = dlogits[(n * COUT + c) * HW + hw];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < COUT; ++c) { s += dl[c] * wr[c][j]; acc[c][j] += dl[c] * a[j]; }
            o[j] = (relu_mask && !(a[j] > 0.f)) ? 0.f : s;
        }
        if (v == 0) {
#pragma unroll
            for (int c = 0; c < COUT; ++c) db[c] += dl[c];
        }
        Elem<T>::store8(dx + p * ld_dx + v * 8, o);
    }
    // lanes l, l+cv, l+2cv ... share a channel vector: butterfly over the lane bits above log2(cv)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = acc[c][j];
            for (int o = 32; o >= cv; o >>= 1) t += __shfl_xor(t, o, 64);
            acc[c][j] = t;
        }
        float t = db[c];
        for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
        db[c] = t;
    }
    if (lane < cv) {
#pragma unroll
        for (int c = 0; c < COUT; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wv][lane][c * 8 + j] = acc[c][j];
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) dbs[wv][c] = db[c];
    }
    __syncthreads();
    float* out = ws + (int64_t)blockIdx.x * (COUT * Cin + COUT);
    for (int t = threadIdx.x; t < COUT * Cin; t += kThreads) {
        const int c = t / Cin, ci = t % Cin;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kThreads / 64; ++k) s += red[k][ci / 8][c * 8 + (ci & 7)];
        out[t] = s;
    }
    if (threadIdx.x < COUT) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kThreads / 64; ++k) s += dbs[k][threadIdx.x];
        out[COUT * Cin + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(kThreads) void head_bwd_finalize_kernel(const float* __restrict__ ws, int nblocks, int per,
                                                                     int split, float* __restrict__ dw,
                                                                     float* __restrict__ db) {
    const int t = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (t >= per) return;
    const double s = wave_sum_partials(ws, t, per, nblocks);
    if ((threadIdx.x & 63) != 0) return;
    if (t < split) dw[t] = (float)s;
    else if (db) db[t - split] = (float)s;
}

// ------------------------------------------------------------------------------ Adam
__global__ __launch_bounds__(kThreads) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                        float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                        float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        float gi = g[i] * gscale;
        if (wd != 0.f) gi += wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// Blocks along x for the row-strided streaming kernels: every thread should see >= 16 pixel rows so that its per-channel
// constants (statistics, affine parameters: 32 scalar loads) amortise, but the launch keeps >= ~1024 blocks when
// there is at least one row per thread for them.
static int64_t stream_blocks(int64_t M, int rows, int64_t groups) {
    const int64_t per_row = (M + rows - 1) / rows;
    int64_t gx = (per_row + 15) / 16;
    const int64_t floor_blocks = (1024 + groups - 1) / groups;
    if (gx < floor_blocks) gx = per_row < floor_blocks ? per_row : floor_blocks;
    const int64_t cap = (8192 + groups - 1) / groups;
    if (gx > cap) gx = cap;
    return gx < 1 ? 1 : gx;
}

static int stats_bpg(int64_t groups, int64_t M, int64_t C) {
    // ~16 pixel rows of work per thread; at most ~2048 blocks in total
    const int64_t rows = kThreads / (C / 8);
    int64_t bpg = M / (rows * 16);
    const int64_t cap = (2048 + groups - 1) / groups;
    if (bpg > cap) bpg = cap;
    // small tensors: a thread that walks 16 rows one after the other is a chain of 16 memory latencies on a GPU that is
    // mostly idle; spread the rows over ~1024 blocks instead, as long as every thread keeps two rows
    const int64_t want = (1024 + groups - 1) / groups, thin = M / (rows * 2);
    if (bpg < want) bpg = thin < want ? (thin > bpg ? thin : bpg) : want;
    if (bpg < 1) bpg = 1;
    return (int)bpg;
}

// whether a tensor of `elems` elements of `dtype` is swept with the big-tensor access pattern (SEGK_STREAM_BIG=0/1 forces it)
static bool big_tensor(int64_t elems, int dtype) {
    static const int forced = getenv("SEGK_STREAM_BIG") ? atoi(getenv("SEGK_STREAM_BIG")) : -1;
    if (forced >= 0) return forced != 0;
    return elems * (int64_t)esize(dtype) >= kBigTensorBytes;
}
#define SEGK_DISPATCH_BIG(flag, BIG, ...)                      \
    do {                                                       \
        if (flag) { constexpr bool BIG = true; __VA_ARGS__; }  \
        else { constexpr bool BIG = false; __VA_ARGS__; }      \
    } while (0)

static int check_act(const char* who, const void* p, int64_t ld, int64_t C, int dtype) {
    if (!p) return fail(SEGK_EINVAL, "%s: null activation pointer", who);
    if (dtype != SEGK_F32 && dtype != SEGK_BF16) return fail(SEGK_EINVAL, "%s: unknown dtype %d", who, dtype);
    if (C <= 0 || C % 8 != 0) return fail(SEGK_ESHAPE, "%s: channel count %lld is not a positive multiple of 8", who, (long long)C);
    if (ld < C) return fail(SEGK_ESHAPE, "%s: ld %lld < C %lld", who, (long long)ld, (long long)C);
    if (!aligned16(p) || (ld * esize(dtype)) % 16 != 0) return fail(SEGK_EALIGN, "%s: pointer/ld not 16-byte aligned", who);
    return SEGK_OK;
}
#define SEGK_CHECK_ACT(...)                          \
    do {                                             \
        int rc_ = check_act(__VA_ARGS__);            \
        if (rc_ != SEGK_OK) return rc_;              \
    } while (0)

static int check_norm_c(const char* who, int64_t C) {
    if (C % 8 != 0 || C <= 0 || C / 8 > kThreads) return fail(SEGK_ESHAPE, "%s: C must be a multiple of 8, at most 2048 (C=%lld)", who, (long long)C);
    return SEGK_OK;
}

}  // namespace segk

using namespace segk;

extern "C" int segk_nchw_to_nhwc(const float* src, int64_t N, int64_t Cin, int64_t H, int64_t W, void* dst, int dst_dtype,
                                 int64_t Cout, int64_t ld, int dev, void* stream) {
    SEGK_REQUIRE(src && N > 0 && Cin > 0 && H > 0 && W > 0 && Cout >= Cin, SEGK_ESHAPE, "nchw_to_nhwc: bad shape");
    SEGK_CHECK_ACT("nchw_to_nhwc", dst, ld, Cout, dst_dtype);
    SEGK_USE_DEVICE(dev);
    const int64_t items = N * H * W * (Cout / 8);
    SEGK_DISPATCH_DTYPE(dst_dtype, T, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(items)), dim3(kThreads), 0,
                                                         (hipStream_t)stream, src, N, (int)Cin, H * W, (T*)dst, (int)Cout, ld));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_im2col3x3(const float* src, int64_t N, int64_t Cin, int64_t H, int64_t W, void* dst, int dst_dtype,
                              int64_t Cout, int64_t ld, int dev, void* stream) {
    SEGK_REQUIRE(src && N > 0 && Cin > 0 && H > 0 && W > 0 && Cout >= 9 * Cin, SEGK_ESHAPE, "im2col3x3: bad shape");
    SEGK_CHECK_ACT("im2col3x3", dst, ld, Cout, dst_dtype);
    SEGK_USE_DEVICE(dev);
    if (Cout == 32 && Cin <= 3) {
        const int g = grid_for(N * H * W);
        hipStream_t st = (hipStream_t)stream;
        if (Cin == 1) { SEGK_DISPATCH_DTYPE(dst_dtype, T, hipLaunchKernelGGL((im2col3x3_pixel_kernel<T, 1>), dim3(g), dim3(kThreads), 0, st, src, N, (int)H, (int)W, (T*)dst, ld)); }
        else if (Cin == 2) { SEGK_DISPATCH_DTYPE(dst_dtype, T, hipLaunchKernelGGL((im2col3x3_pixel_kernel<T, 2>), dim3(g), dim3(kThreads), 0, st, src, N, (int)H, (int)W, (T*)dst, ld)); }
        else { SEGK_DISPATCH_DTYPE(dst_dtype, T, hipLaunchKernelGGL((im2col3x3_pixel_kernel<T, 3>), dim3(g), dim3(kThreads), 0, st, src, N, (int)H, (int)W, (T*)dst, ld)); }
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    const int64_t items = N * H * W * (Cout / 8);
    SEGK_DISPATCH_DTYPE(dst_dtype, T, hipLaunchKernelGGL((im2col3x3_kernel<T>), dim3(grid_for(items)), dim3(kThreads), 0,
                                                         (hipStream_t)stream, src, N, (int)Cin, (int)H, (int)W, (T*)dst, (int)Cout, ld));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" size_t segk_norm_workspace(int64_t groups, int64_t M, int64_t C) {
    if (groups <= 0 || M <= 0 || C <= 0 || C % 8) return 0;
    // block partials [groups][bpg][2][C] followed by the reduced backward sums [groups][C][2]
    return align_up((size_t)groups * stats_bpg(groups, M, C) * 2 * C * sizeof(float), 256) +
           align_up((size_t)groups * C * 2 * sizeof(float), 256);
}

extern "C" int segk_norm_stats(int dtype, const void* x, int64_t ld, int64_t groups, int64_t M, int64_t C, float eps,
                               float* stats, float* running_mean, float* running_var, float momentum, void* workspace,
                               size_t workspace_bytes, int dev, void* stream) {
    SEGK_CHECK_ACT("norm_stats", x, ld, C, dtype);
    if (int rc = check_norm_c("norm_stats", C)) return rc;
    SEGK_REQUIRE(stats && workspace && groups > 0 && groups < 65536 && M > 0, SEGK_EINVAL, "norm_stats: bad arguments");
    SEGK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), SEGK_EINVAL, "norm_stats: running buffers come in pairs");
    SEGK_REQUIRE(!running_mean || groups == 1, SEGK_EINVAL, "norm_stats: running buffers need groups == 1");
    SEGK_REQUIRE(workspace_bytes >= segk_norm_workspace(groups, M, C), SEGK_EWORKSPACE, "norm_stats: workspace too small");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int bpg = stats_bpg(groups, M, C);
    // the per-(group, channel) shift lives in the second workspace section (the backward's reduced sums use it later)
    float* kshift = (float*)((char*)workspace + align_up((size_t)groups * bpg * 2 * C * sizeof(float), 256));
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((stats_partial_kernel<T, 0>), dim3(bpg, (unsigned)groups), dim3(kThreads), 0, st,
                                                     (const T*)x, ld, M, (int)C, bpg, (float*)workspace, kshift));
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)C, (unsigned)groups), dim3(kThreads), 0, st,
                       (const float*)workspace, bpg, M, (int)C, eps, stats, running_mean, running_var, momentum, (const float*)kshift,
                       (int64_t)(2 * C));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" size_t segk_norm_stats_finalize_workspace(int64_t groups, int64_t rows_per_group, int64_t C) {
    return segk_norm_workspace(groups, rows_per_group, 2 * C);
}

extern "C" int segk_norm_stats_finalize(const float* partials, int64_t groups, int64_t rows_per_group, int64_t M, int64_t C, float eps,
                                        float* stats, float* running_mean, float* running_var, float momentum, void* workspace,
                                        size_t workspace_bytes, int dev, void* stream) {
    SEGK_REQUIRE(partials && stats && groups > 0 && groups < 65536 && rows_per_group > 0 && rows_per_group < (1ll << 31) && M > 0 && C > 0,
                 SEGK_EINVAL, "norm_stats_finalize: bad arguments");
    SEGK_REQUIRE(C % 8 == 0 && 2 * C / 8 <= kThreads && aligned16(partials), SEGK_ESHAPE, "norm_stats_finalize: C must be a multiple of 8, at most 1024");
    SEGK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), SEGK_EINVAL, "norm_stats_finalize: running buffers come in pairs");
    SEGK_REQUIRE(!running_mean || groups == 1, SEGK_EINVAL, "norm_stats_finalize: running buffers need groups == 1");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    if (rows_per_group <= 128) {
        hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)C, (unsigned)groups), dim3(kThreads), 0, st, partials, (int)rows_per_group, M,
                           (int)C, eps, stats, running_mean, running_var, momentum, (const float*)nullptr, (int64_t)(2 * C));
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    // Many rows: first a coalesced column sum of the [rows][2C] partial matrix by many workgroups, then the per-channel
    // finalize over the few rows that are left.  The column-sum kernel leaves every second output row unused, hence
    // the row stride of 4*C.
    SEGK_REQUIRE(workspace && workspace_bytes >= segk_norm_stats_finalize_workspace(groups, rows_per_group, C), SEGK_EWORKSPACE,
                 "norm_stats_finalize: workspace too small");
    const int bpg = stats_bpg(groups, rows_per_group, 2 * C);
    hipLaunchKernelGGL((stats_partial_kernel<float, 1>), dim3(bpg, (unsigned)groups), dim3(kThreads), 0, st, partials, 2 * C, rows_per_group,
                       (int)(2 * C), bpg, (float*)workspace, (float*)nullptr);
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)C, (unsigned)groups), dim3(kThreads), 0, st, (const float*)workspace, bpg, M, (int)C,
                       eps, stats, running_mean, running_var, momentum, (const float*)nullptr, (int64_t)(4 * C));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_norm_eval_stats(const float* running_mean, const float* running_var, int64_t C, float eps, float* stats,
                                    int dev, void* stream) {
    SEGK_REQUIRE(running_mean && running_var && stats && C > 0, SEGK_EINVAL, "norm_eval_stats: bad arguments");
    SEGK_USE_DEVICE(dev);
    hipLaunchKernelGGL(eval_stats_kernel, dim3((unsigned)((C + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                       running_mean, running_var, (int)C, eps, stats);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_norm_apply(int dtype, const void* x, int64_t ldx, int64_t groups, int64_t M, int64_t C, const float* stats,
                               const float* gamma, const float* beta, int flags, void* y, int64_t ldy, int dev, void* stream) {
    SEGK_CHECK_ACT("norm_apply(x)", x, ldx, C, dtype);
    SEGK_CHECK_ACT("norm_apply(y)", y, ldy, C, dtype);
    if (int rc = check_norm_c("norm_apply", C)) return rc;
    SEGK_REQUIRE(stats && groups > 0 && groups < 65536 && M > 0, SEGK_EINVAL, "norm_apply: bad arguments");
    SEGK_USE_DEVICE(dev);
    const int rows = kThreads / (int)(C / 8);
    const int64_t gx = stream_blocks(M, rows, groups);
    SEGK_DISPATCH_BIG(big_tensor(groups * M * C, dtype), BIG, SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((norm_apply_kernel<T, BIG>), dim3((unsigned)gx, (unsigned)groups), dim3(kThreads), 0,
                                                     (hipStream_t)stream, (const T*)x, ldx, M, (int)C, stats, gamma, beta, flags & 1,
                                                     (T*)y, ldy)));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_norm_bwd(int dtype, const void* dy, int64_t ld_dy, const void* x, int64_t ldx, int64_t groups, int64_t M,
                             int64_t C, const float* stats, const float* gamma, const float* beta, int flags, void* dx,
                             int64_t ld_dx, float* dgamma, float* dbeta, float* dbias_out, void* workspace,
                             size_t workspace_bytes, int dev, void* stream) {
    SEGK_CHECK_ACT("norm_bwd(dy)", dy, ld_dy, C, dtype);
    SEGK_CHECK_ACT("norm_bwd(x)", x, ldx, C, dtype);
    SEGK_CHECK_ACT("norm_bwd(dx)", dx, ld_dx, C, dtype);
    if (int rc = check_norm_c("norm_bwd", C)) return rc;
    SEGK_REQUIRE(stats && workspace && groups > 0 && groups < 65536 && M > 0, SEGK_EINVAL, "norm_bwd: bad arguments");
    SEGK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), SEGK_EINVAL, "norm_bwd: dgamma/dbeta come in pairs");
    SEGK_REQUIRE(!dgamma || groups == 1, SEGK_EINVAL, "norm_bwd: affine gradients need groups == 1");
    SEGK_REQUIRE(workspace_bytes >= segk_norm_workspace(groups, M, C), SEGK_EWORKSPACE, "norm_bwd: workspace too small");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int bpg = stats_bpg(groups, M, C);
    const int relu = flags & 1, accumulate = (flags >> 1) & 1, frozen = (flags >> 2) & 1;
    float* part = (float*)workspace;
    float* red = (float*)((char*)workspace + align_up((size_t)groups * bpg * 2 * C * sizeof(float), 256));
    SEGK_DISPATCH_BIG(big_tensor(groups * M * C, dtype), BIG, SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((norm_bwd_partial_kernel<T, BIG>), dim3(bpg, (unsigned)groups), dim3(kThreads), 0, st,
                                                     (const T*)dy, ld_dy, (const T*)x, ldx, M, (int)C, stats, gamma, beta, relu, bpg, part)));
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((unsigned)C, (unsigned)groups), dim3(kThreads), 0, st,
                       (const float*)part, bpg, (int)C, red, dgamma, dbeta, accumulate, (int64_t)(2 * C));
    SEGK_LAUNCH_CHECK();
    const int rows = kThreads / (int)(C / 8);
    const int64_t gx = stream_blocks(M, rows, groups);
    SEGK_DISPATCH_BIG(big_tensor(groups * M * C, dtype), BIG, SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((norm_bwd_dx_kernel<T, BIG>), dim3((unsigned)gx, (unsigned)groups), dim3(kThreads), 0, st,
                                                     (const T*)dy, ld_dy, (const T*)x, ldx, M, (int)C, stats, gamma, beta, relu, frozen,
                                                     (const float*)red, (T*)dx, ld_dx, (flags >> 3) & 1)));
    SEGK_LAUNCH_CHECK();
    if (dbias_out) {
        // With batch statistics the gradient reaching the convolution in front of the norm sums to zero
        // over the pixels of a group, so its bias gradient is identically zero.
        SEGK_REQUIRE(!frozen, SEGK_EINVAL, "norm_bwd: bias gradient with frozen statistics: use segk_colsum on dx");
        if (!accumulate) SEGK_HIP(hipMemsetAsync(dbias_out, 0, (size_t)C * sizeof(float), st));
    }
    return SEGK_OK;
}

// Second half of segk_norm_bwd for a BatchNorm (one group) whose two sums were already reduced per wave by the
// convolution that produced dy (segk_conv_dgrad_norm): column-sum the partial rows, finalize, write dx.
extern "C" size_t segk_norm_bwd_finish_workspace(int64_t rows, int64_t C) {
    if (rows <= 0 || C <= 0) return 0;
    return align_up(segk_norm_stats_finalize_workspace(1, rows, C), 256) + align_up((size_t)2 * C * sizeof(float), 256);
}

extern "C" int segk_norm_bwd_finish(int dtype, const float* partials, int64_t rows, const void* dy, int64_t ld_dy, const void* x,
                                    int64_t ldx, int64_t M, int64_t C, const float* stats, const float* gamma, const float* beta,
                                    int flags, void* dx, int64_t ld_dx, float* dgamma, float* dbeta, void* workspace,
                                    size_t workspace_bytes, int dev, void* stream) {
    SEGK_CHECK_ACT("norm_bwd_finish(dy)", dy, ld_dy, C, dtype);
    SEGK_CHECK_ACT("norm_bwd_finish(x)", x, ldx, C, dtype);
    SEGK_CHECK_ACT("norm_bwd_finish(dx)", dx, ld_dx, C, dtype);
    if (int rc = check_norm_c("norm_bwd_finish", C)) return rc;
    SEGK_REQUIRE(partials && aligned16(partials) && rows > 0 && rows < (1ll << 31) && stats && workspace && M > 0, SEGK_EINVAL,
                 "norm_bwd_finish: bad arguments");
    SEGK_REQUIRE(2 * C / 8 <= kThreads, SEGK_ESHAPE, "norm_bwd_finish: at most 1024 channels");
    SEGK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), SEGK_EINVAL, "norm_bwd_finish: dgamma/dbeta come in pairs");
    SEGK_REQUIRE(!(flags & 4), SEGK_EINVAL, "norm_bwd_finish: frozen statistics need no sums: use segk_norm_bwd");
    SEGK_REQUIRE(workspace_bytes >= segk_norm_bwd_finish_workspace(rows, C), SEGK_EWORKSPACE, "norm_bwd_finish: workspace too small");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int relu = flags & 1, accumulate = (flags >> 1) & 1;
    float* red = (float*)((char*)workspace + align_up(segk_norm_stats_finalize_workspace(1, rows, C), 256));
    if (rows <= 128) {
        hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((unsigned)C, 1), dim3(kThreads), 0, st, partials, (int)rows, (int)C, red, dgamma,
                           dbeta, accumulate, (int64_t)(2 * C));
    } else {
        // coalesced column sum of the [rows][2C] matrix by many workgroups first (every second output row stays unused,
        // hence the row stride of 4*C), then the per-channel finalize over the few rows that are left
        const int bpg = stats_bpg(1, rows, 2 * C);
        hipLaunchKernelGGL((stats_partial_kernel<float, 1>), dim3(bpg, 1), dim3(kThreads), 0, st, partials, 2 * C, rows, (int)(2 * C), bpg,
                           (float*)workspace, (float*)nullptr);
        SEGK_LAUNCH_CHECK();
        hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((unsigned)C, 1), dim3(kThreads), 0, st, (const float*)workspace, bpg, (int)C, red,
                           dgamma, dbeta, accumulate, (int64_t)(4 * C));
    }
    SEGK_LAUNCH_CHECK();
    const int rws = kThreads / (int)(C / 8);
    const int64_t gx = stream_blocks(M, rws, 1);
    SEGK_DISPATCH_BIG(big_tensor(M * C, dtype), BIG, SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((norm_bwd_dx_kernel<T, BIG>), dim3((unsigned)gx, 1), dim3(kThreads), 0, st,
                                                     (const T*)dy, ld_dy, (const T*)x, ldx, M, (int)C, stats, gamma, beta, relu, 0,
                                                     (const float*)red, (T*)dx, ld_dx, (flags >> 3) & 1)));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// The reduction half of segk_norm_bwd_finish on its own: partial rows [rows][2][C] of (sum g, sum g*xhat) -> red[C][2],
// dgamma, dbeta.  For a producer of the partials that also computes dx itself (segk_conv_first_norm, modes 3 and 4).
// workspace: segk_norm_stats_finalize_workspace(1, rows, C) bytes.
extern "C" int segk_norm_bwd_sums_finalize(const float* partials, int64_t rows, int64_t C, float* red, float* dgamma, float* dbeta,
                                           int flags, void* workspace, size_t workspace_bytes, int dev, void* stream) {
    if (int rc = check_norm_c("norm_bwd_sums_finalize", C)) return rc;
    SEGK_REQUIRE(partials && aligned16(partials) && rows > 0 && rows < (1ll << 31) && red, SEGK_EINVAL, "norm_bwd_sums_finalize: bad arguments");
    SEGK_REQUIRE(2 * C / 8 <= kThreads, SEGK_ESHAPE, "norm_bwd_sums_finalize: at most 1024 channels");
    SEGK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), SEGK_EINVAL, "norm_bwd_sums_finalize: dgamma/dbeta come in pairs");
    SEGK_REQUIRE(rows <= 128 || (workspace && workspace_bytes >= segk_norm_stats_finalize_workspace(1, rows, C)), SEGK_EWORKSPACE,
                 "norm_bwd_sums_finalize: workspace too small");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int accumulate = (flags >> 1) & 1;
    if (rows <= 128) {
        hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((unsigned)C, 1), dim3(kThreads), 0, st, partials, (int)rows, (int)C, red, dgamma,
                           dbeta, accumulate, (int64_t)(2 * C));
    } else {
        const int bpg = stats_bpg(1, rows, 2 * C);
        hipLaunchKernelGGL((stats_partial_kernel<float, 1>), dim3(bpg, 1), dim3(kThreads), 0, st, partials, 2 * C, rows, (int)(2 * C), bpg,
                           (float*)workspace, (float*)nullptr);
        SEGK_LAUNCH_CHECK();
        hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((unsigned)C, 1), dim3(kThreads), 0, st, (const float*)workspace, bpg, (int)C, red,
                           dgamma, dbeta, accumulate, (int64_t)(4 * C));
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_norm_pool_apply(int dtype, const void* x, int64_t ldx, int64_t N, int64_t H, int64_t W, int64_t C,
                                    const float* stats, const float* gamma, const float* beta, void* y, int64_t ldy, void* pool,
                                    int64_t ld_pool, int dev, void* stream) {
    SEGK_CHECK_ACT("norm_pool_apply(x)", x, ldx, C, dtype);
    SEGK_CHECK_ACT("norm_pool_apply(y)", y, ldy, C, dtype);
    SEGK_CHECK_ACT("norm_pool_apply(pool)", pool, ld_pool, C, dtype);
    if (int rc = check_norm_c("norm_pool_apply", C)) return rc;
    SEGK_REQUIRE(stats && N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, SEGK_ESHAPE,
                 "norm_pool_apply: H and W must be even (H=%lld W=%lld)", (long long)H, (long long)W);
    SEGK_USE_DEVICE(dev);
    const int64_t Q = N * (H / 2) * (W / 2);
    const int rows = kThreads / (int)(C / 8);
    const int64_t gx = stream_blocks(Q, rows, 1);
    SEGK_DISPATCH_BIG(big_tensor(4 * Q * C, dtype), BIG, SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((norm_pool_apply_kernel<T, BIG>), dim3((unsigned)gx), dim3(kThreads), 0,
                                                     (hipStream_t)stream, (const T*)x, ldx, Q, (int)H, (int)W, (int)C, stats, gamma, beta,
                                                     (T*)y, ldy, (T*)pool, ld_pool)));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_norm_pool_bwd(int dtype, const void* dy, int64_t ld_dy, const void* dpool, int64_t ld_dpool, const void* x,
                                  int64_t ldx, int64_t N, int64_t H, int64_t W, int64_t C, const float* stats,
                                  const float* gamma, const float* beta, int flags, void* dx, int64_t ld_dx, float* dgamma,
                                  float* dbeta, void* workspace, size_t workspace_bytes, int dev, void* stream) {
    if (dy) SEGK_CHECK_ACT("norm_pool_bwd(dy)", dy, ld_dy, C, dtype);
    SEGK_CHECK_ACT("norm_pool_bwd(dpool)", dpool, ld_dpool, C, dtype);
    SEGK_CHECK_ACT("norm_pool_bwd(x)", x, ldx, C, dtype);
    SEGK_CHECK_ACT("norm_pool_bwd(dx)", dx, ld_dx, C, dtype);
    if (int rc = check_norm_c("norm_pool_bwd", C)) return rc;
    SEGK_REQUIRE(stats && workspace && N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, SEGK_ESHAPE,
                 "norm_pool_bwd: H and W must be even (H=%lld W=%lld)", (long long)H, (long long)W);
    SEGK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), SEGK_EINVAL, "norm_pool_bwd: dgamma/dbeta come in pairs");
    const int64_t M = N * H * W, Q = M / 4;
    SEGK_REQUIRE(workspace_bytes >= segk_norm_workspace(1, M, C), SEGK_EWORKSPACE, "norm_pool_bwd: workspace too small");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int bpg = stats_bpg(1, Q, C);
    const int accumulate = (flags >> 1) & 1, frozen = (flags >> 2) & 1;
    float* part = (float*)workspace;
    float* red = (float*)((char*)workspace + align_up((size_t)bpg * 2 * C * sizeof(float), 256));
    SEGK_DISPATCH_BIG(big_tensor(M * C, dtype), BIG, SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((norm_pool_bwd_partial_kernel<T, BIG>), dim3(bpg), dim3(kThreads), 0, st, (const T*)dy, ld_dy,
                                                     (const T*)dpool, ld_dpool, (const T*)x, ldx, Q, (int)H, (int)W, (int)C, stats, gamma,
                                                     beta, bpg, part)));
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((unsigned)C, 1), dim3(kThreads), 0, st, (const float*)part, bpg, (int)C,
                       red, dgamma, dbeta, accumulate, (int64_t)(2 * C));
    SEGK_LAUNCH_CHECK();
    const int rows = kThreads / (int)(C / 8);
    const int64_t gx = stream_blocks(Q, rows, 1);
    SEGK_DISPATCH_BIG(big_tensor(M * C, dtype), BIG, SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((norm_pool_bwd_dx_kernel<T, BIG>), dim3((unsigned)gx), dim3(kThreads), 0, st, (const T*)dy,
                                                     ld_dy, (const T*)dpool, ld_dpool, (const T*)x, ldx, Q, (int)H, (int)W, (int)C, M, stats,
                                                     gamma, beta, frozen, (const float*)red, (T*)dx, ld_dx)));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_maxpool2x2_fwd(int dtype, const void* x, int64_t ldx, int64_t N, int64_t H, int64_t W, int64_t C, void* y,
                                   int64_t ldy, int dev, void* stream) {
    SEGK_CHECK_ACT("maxpool_fwd(x)", x, ldx, C, dtype);
    SEGK_CHECK_ACT("maxpool_fwd(y)", y, ldy, C, dtype);
    SEGK_REQUIRE(N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, SEGK_ESHAPE, "maxpool_fwd: H and W must be even (H=%lld W=%lld)", (long long)H, (long long)W);
    SEGK_USE_DEVICE(dev);
    const int64_t items = N * (H / 2) * (W / 2) * (C / 8);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(grid_for(items)), dim3(kThreads), 0, (hipStream_t)stream,
                                                     (const T*)x, ldx, N, (int)H, (int)W, (int)C, (T*)y, ldy));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_maxpool2x2_bwd(int dtype, const void* x, int64_t ldx, const void* dy, int64_t ld_dy, int64_t N, int64_t H,
                                   int64_t W, int64_t C, void* dx, int64_t ld_dx, int flags, int dev, void* stream) {
    SEGK_CHECK_ACT("maxpool_bwd(x)", x, ldx, C, dtype);
    SEGK_CHECK_ACT("maxpool_bwd(dy)", dy, ld_dy, C, dtype);
    SEGK_CHECK_ACT("maxpool_bwd(dx)", dx, ld_dx, C, dtype);
    SEGK_REQUIRE(N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, SEGK_ESHAPE, "maxpool_bwd: H and W must be even");
    SEGK_USE_DEVICE(dev);
    const int64_t items = N * (H / 2) * (W / 2) * (C / 8);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid_for(items)), dim3(kThreads), 0, (hipStream_t)stream,
                                                     (const T*)x, ldx, (const T*)dy, ld_dy, N, (int)H, (int)W, (int)C, (T*)dx, ld_dx,
                                                     (flags >> 1) & 1));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_upsample2x_fwd(int dtype, const void* x, int64_t ldx, int64_t N, int64_t H, int64_t W, int64_t C,
                                   const float* stats, int flags, void* y, int64_t ldy, int dev, void* stream) {
    SEGK_CHECK_ACT("upsample_fwd(x)", x, ldx, C, dtype);
    SEGK_CHECK_ACT("upsample_fwd(y)", y, ldy, C, dtype);
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0, SEGK_ESHAPE, "upsample_fwd: bad shape");
    SEGK_USE_DEVICE(dev);
    if (!((flags >> 2) & 1)) {
        SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((upsample_fwd_half_kernel<T>), dim3((unsigned)patch_blocks(N, H, W, C)), dim3(kThreads),
                                                         0, (hipStream_t)stream, (const T*)x, ldx, (int)H, (int)W, (int)C, stats,
                                                         flags & 1, (T*)y, ldy));
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((upsample_fwd_block_kernel<T>), dim3((unsigned)patch_blocks(N, H, W, C)), dim3(kThreads),
                                                     0, (hipStream_t)stream, (const T*)x, ldx, (int)H, (int)W, (int)C, stats,
                                                     flags & 1, 1, (T*)y, ldy));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_upsample2x_bwd(int dtype, const void* dy, int64_t ld_dy, const void* y, int64_t ldy, int64_t N, int64_t H,
                                   int64_t W, int64_t C, int flags, void* dx, int64_t ld_dx, int dev, void* stream) {
    SEGK_CHECK_ACT("upsample_bwd(dy)", dy, ld_dy, C, dtype);
    SEGK_CHECK_ACT("upsample_bwd(dx)", dx, ld_dx, C, dtype);
    if (flags & 1) SEGK_CHECK_ACT("upsample_bwd(y)", y, ldy, C, dtype);
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0, SEGK_ESHAPE, "upsample_bwd: bad shape");
    SEGK_USE_DEVICE(dev);
    if (!((flags >> 2) & 1)) {
        SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((upsample_bwd_half_kernel<T>), dim3((unsigned)patch_blocks(N, H, W, C)), dim3(kThreads), 0,
                                                         (hipStream_t)stream, (const T*)dy, ld_dy, (const T*)y, ldy, (int)H, (int)W,
                                                         (int)C, flags & 1, (T*)dx, ld_dx));
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((upsample_bwd_block_kernel<T>), dim3((unsigned)patch_blocks(N, H, W, C)), dim3(kThreads), 0,
                                                     (hipStream_t)stream, (const T*)dy, ld_dy, (const T*)y, ldy, (int)H, (int)W, (int)C,
                                                     flags & 1, (flags >> 2) & 1, (T*)dx, ld_dx));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_resize_bilinear_fwd(int dtype, const void* x, int64_t ldx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t Ho,
                                        int64_t Wo, float ratio_h, float ratio_w, const float* stats, int flags, void* y, int64_t ldy,
                                        int dev, void* stream) {
    SEGK_CHECK_ACT("resize_fwd(x)", x, ldx, C, dtype);
    SEGK_CHECK_ACT("resize_fwd(y)", y, ldy, C, dtype);
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && ratio_h > 0.f && ratio_w > 0.f, SEGK_ESHAPE, "resize_fwd: bad shape");
    SEGK_USE_DEVICE(dev);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((resize_fwd_kernel<T>), dim3((unsigned)patch_blocks(N, Ho, Wo, C)), dim3(kThreads), 0,
                                                     (hipStream_t)stream, (const T*)x, ldx, (int)H, (int)W, (int)C, (int)Ho, (int)Wo, ratio_h,
                                                     ratio_w, stats, flags & 1, (T*)y, ldy));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_resize_bilinear_bwd(int dtype, const void* dy, int64_t ld_dy, int64_t N, int64_t H, int64_t W, int64_t C, int64_t Ho,
                                        int64_t Wo, float ratio_h, float ratio_w, void* dx, int64_t ld_dx, int flags, int dev,
                                        void* stream) {
    SEGK_CHECK_ACT("resize_bwd(dy)", dy, ld_dy, C, dtype);
    SEGK_CHECK_ACT("resize_bwd(dx)", dx, ld_dx, C, dtype);
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && ratio_h > 0.f && ratio_w > 0.f, SEGK_ESHAPE, "resize_bwd: bad shape");
    SEGK_USE_DEVICE(dev);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((resize_bwd_kernel<T>), dim3((unsigned)patch_blocks(N, H, W, C)), dim3(kThreads), 0,
                                                     (hipStream_t)stream, (const T*)dy, ld_dy, (int)H, (int)W, (int)C, (int)Ho, (int)Wo, ratio_h,
                                                     ratio_w, (T*)dx, ld_dx, (flags >> 1) & 1));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" size_t segk_colsum_workspace(int64_t P, int64_t C) {
    if (P <= 0 || C <= 0 || C % 8) return 0;
    return align_up((size_t)stats_bpg(1, P, C) * 2 * C * sizeof(float), 256);
}

extern "C" int segk_colsum(int dtype, const void* x, int64_t ld, int64_t P, int64_t C, float* out, int flags, void* workspace,
                           size_t workspace_bytes, int dev, void* stream) {
    SEGK_CHECK_ACT("colsum", x, ld, C, dtype);
    if (int rc = check_norm_c("colsum", C)) return rc;
    SEGK_REQUIRE(out && workspace && P > 0, SEGK_EINVAL, "colsum: bad arguments");
    SEGK_REQUIRE(workspace_bytes >= segk_colsum_workspace(P, C), SEGK_EWORKSPACE, "colsum: workspace too small");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int bpg = stats_bpg(1, P, C);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((stats_partial_kernel<T, 1>), dim3(bpg, 1), dim3(kThreads), 0, st, (const T*)x, ld, P,
                                                     (int)C, bpg, (float*)workspace, (float*)nullptr));
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)C), dim3(kThreads), 0, st,
                       (const float*)workspace, bpg, (int)C, out, (flags >> 1) & 1);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_copy_slice(int dtype, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t P, int64_t C,
                               int flags, int dev, void* stream) {
    if (src) SEGK_CHECK_ACT("copy_slice(src)", src, ld_src, C, dtype);
    SEGK_CHECK_ACT("copy_slice(dst)", dst, ld_dst, C, dtype);
    SEGK_REQUIRE(P > 0, SEGK_ESHAPE, "copy_slice: bad shape");
    SEGK_REQUIRE(src || !(flags & 2), SEGK_EINVAL, "copy_slice: accumulating zeros is a no-op: pass a source");
    SEGK_USE_DEVICE(dev);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((copy_slice_kernel<T>), dim3(grid_for(P * (C / 8))), dim3(kThreads), 0,
                                                     (hipStream_t)stream, (const T*)src, ld_src, (T*)dst, ld_dst, P, (int)C, (flags >> 1) & 1));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_block2(int dtype, void* x, int64_t ldx, void* wide, int64_t ld_wide, int64_t N, int64_t H, int64_t W, int64_t C,
                           int flags, int dev, void* stream) {
    SEGK_CHECK_ACT("block2(x)", x, ldx, C, dtype);
    SEGK_CHECK_ACT("block2(wide)", wide, ld_wide, 4 * C, dtype);
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && H < (1 << 20) && W < (1 << 20), SEGK_ESHAPE,
                 "block2: H and W must be even (got %lldx%lld)", (long long)H, (long long)W);
    SEGK_USE_DEVICE(dev);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((block2_kernel<T>), dim3(grid_for(N * H * W * (C / 8))), dim3(kThreads), 0,
                                                     (hipStream_t)stream, (T*)x, ldx, (T*)wide, ld_wide, N, (int)(H / 2), (int)(W / 2), (int)C,
                                                     (flags & 1) ? 0 : 1, (flags >> 1) & 1));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_head1x1_fwd(int dtype, const void* x, int64_t ld, int64_t Cin, int64_t N, int64_t HW, const float* w,
                                const float* bias, int64_t Cout, float* logits, int dev, void* stream) {
    SEGK_CHECK_ACT("head1x1_fwd", x, ld, Cin, dtype);
    SEGK_REQUIRE(w && logits && N > 0 && HW > 0, SEGK_EINVAL, "head1x1_fwd: bad arguments");
    SEGK_REQUIRE(Cout >= 1 && Cout <= kHeadMaxOut && Cin <= 1024, SEGK_ESHAPE, "head1x1_fwd: Cout must be 1..8 (got %lld)", (long long)Cout);
    SEGK_USE_DEVICE(dev);
    // pixel rows staged in LDS, TP per round: as many as fit next to the weights in 48 KB (none: direct reads)
    const size_t wbytes = (size_t)((Cout * Cin + Cout + 3) & ~3) * sizeof(float);
    const size_t rowb = (size_t)Cin * esize(dtype) + 16;
    int tp = kThreads;
    while (tp >= 64 && wbytes + tp * rowb > 48 * 1024) tp /= 2;
    if (tp < 64) tp = 0;
    const size_t smem = wbytes + (size_t)tp * rowb;
    const int64_t units = tp ? (N * HW + tp - 1) / tp : (N * HW + kThreads - 1) / kThreads;
    const unsigned grid = (unsigned)(units < 4096 ? (units < 1 ? 1 : units) : 4096);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((head_fwd_kernel<T>), dim3(grid), dim3(kThreads), smem,
                                                     (hipStream_t)stream, (const T*)x, ld, (int)Cin, N, HW, w, bias, (int)Cout, logits, tp));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

static int head_bwd_blocks(int64_t Cin, int64_t N, int64_t HW) { return grid_for(N * HW * (Cin / 8), 1024); }

extern "C" size_t segk_head1x1_bwd_workspace(int64_t Cin, int64_t Cout, int64_t N, int64_t HW) {
    if (Cin <= 0 || Cout <= 0 || N <= 0 || HW <= 0) return 0;
    return align_up((size_t)head_bwd_blocks(Cin, N, HW) * (Cout * Cin + Cout) * sizeof(float), 256);
}

template <typename T>
static int launch_head_bwd(int Cout, dim3 grid, hipStream_t st, const void* x, int64_t ld, int Cin, int64_t N, int64_t HW,
                           const float* w, const float* dlogits, void* dx, int64_t ld_dx, int relu_mask, float* ws) {
#define SEGK_HB(K)                                                                                                            \
    case K:                                                                                                                   \
        hipLaunchKernelGGL((head_bwd_kernel<T, K>), grid, dim3(kThreads), 0, st, (const T*)x, ld, Cin, N, HW, w, dlogits, (T*)dx, \
                           ld_dx, relu_mask, ws);                                                                             \
        break;
    switch (Cout) {
        SEGK_HB(1) SEGK_HB(2) SEGK_HB(3) SEGK_HB(4) SEGK_HB(5) SEGK_HB(6) SEGK_HB(7) SEGK_HB(8)
        default: return fail(SEGK_ESHAPE, "head1x1_bwd: Cout must be 1..8");
    }
#undef SEGK_HB
    return SEGK_OK;
}

extern "C" int segk_head1x1_bwd(int dtype, const void* x, int64_t ld, int64_t Cin, int64_t N, int64_t HW, const float* w,
                                int64_t Cout, const float* dlogits, void* dx, int64_t ld_dx, int relu_mask, float* dw, float* db,
                                void* workspace, size_t workspace_bytes, int dev, void* stream) {
    SEGK_CHECK_ACT("head1x1_bwd(x)", x, ld, Cin, dtype);
    SEGK_CHECK_ACT("head1x1_bwd(dx)", dx, ld_dx, Cin, dtype);
    SEGK_REQUIRE(w && dlogits && dw && workspace && N > 0 && HW > 0, SEGK_EINVAL, "head1x1_bwd: bad arguments");
    SEGK_REQUIRE(Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64, SEGK_ESHAPE, "head1x1_bwd: Cin must be 8, 16, 32 or 64 (got %lld)", (long long)Cin);
    SEGK_REQUIRE(workspace_bytes >= segk_head1x1_bwd_workspace(Cin, Cout, N, HW), SEGK_EWORKSPACE, "head1x1_bwd: workspace too small");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int nb = head_bwd_blocks(Cin, N, HW);
    int rc = SEGK_OK;
    SEGK_DISPATCH_DTYPE(dtype, T, rc = launch_head_bwd<T>((int)Cout, dim3(nb), st, x, ld, (int)Cin, N, HW, w, dlogits, dx, ld_dx,
                                                          relu_mask, (float*)workspace));
    if (rc != SEGK_OK) return rc;
    SEGK_LAUNCH_CHECK();
    const int per = (int)(Cout * Cin + Cout);
    hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3((per + 3) / 4), dim3(kThreads), 0, st, (const float*)workspace, nb,
                       per, (int)(Cout * Cin), dw, db);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int64_t step, float grad_scale, int dev, void* stream) {
    SEGK_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0, SEGK_EINVAL, "adam_step: bad arguments");
    SEGK_USE_DEVICE(dev);
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 4096)), dim3(kThreads), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_relu_bwd(int dtype, const void* dy, int64_t ld_dy, const void* y, int64_t ldy, int64_t P, int64_t C, void* dx,
                             int64_t ld_dx, int dev, void* stream) {
    SEGK_CHECK_ACT("relu_bwd(dy)", dy, ld_dy, C, dtype);
    SEGK_CHECK_ACT("relu_bwd(y)", y, ldy, C, dtype);
    SEGK_CHECK_ACT("relu_bwd(dx)", dx, ld_dx, C, dtype);
    SEGK_REQUIRE(P > 0, SEGK_ESHAPE, "relu_bwd: bad shape");
    SEGK_USE_DEVICE(dev);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((relu_bwd_kernel<T>), dim3(grid_for(P * (C / 8))), dim3(kThreads), 0,
                                                     (hipStream_t)stream, (const T*)dy, ld_dy, (const T*)y, ldy, P, (int)C, (T*)dx, ld_dx));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}
