// Streaming kernels for volumes (the 3-D classifier behind models/resnt.py:26-45 and models/densenet3d.py).
// A volume batch [N, C, D, H, W] is stored slice by slice as [N*D, H, W, ld] (channels innermost), so every
// channel-wise kernel of stream.hip serves it unchanged; the kernels here are the ones that look across slices:
// patch gathering for the strided stem convolutions, MaxPool3d(3, 2, 1) and AvgPool3d(2, 2).
// All of them are HBM/L2-bound gathers: one thread moves 8 consecutive channels (16 bytes of bf16) per access.
#include "common.h"

namespace segk {

constexpr int kVThreads = 256;

static inline int vgrid(int64_t items, int cap = 16384) {
    int64_t g = (items + kVThreads - 1) / kVThreads;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

struct Geo3 {
    int N, C;            // volumes, channels gathered per voxel
    int D, H, W;         // input extent
    int Do, Ho, Wo;      // output extent
    int kd, kh, kw, sd, sh, sw, pd, ph, pw;
};

// ------------------------------------------------------------------------------------------------
// patches[n, od, oh, ow, ((a*kh + b)*kw + e)*C + c] = x[n, od*sd - pd + a, oh*sh - ph + b, ow*sw - pw + e, c]
// (zero outside the volume and in the padding columns behind K = C*kd*kh*kw): tap-major, channel-minor columns, i.e.
// a Conv3d weight permuted to [Cout, kd, kh, kw, C] and flattened, so the strided convolution becomes a 1x1
// convolution and the channels of one tap are contiguous both in x and in the patch row.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kVThreads) void im2col3d_kernel(const T* __restrict__ x, int64_t ldx, Geo3 g, T* __restrict__ out,
                                                             int64_t ldo, int Kp) {
    // column -> (depth tap, row tap, column tap, channel) once per workgroup (the integer divisions dominate otherwise):
    // a << 24 | b << 16 | e << 8 | c, or -1 for the zero columns behind K
    extern __shared__ int col_info[];
    const int K = g.C * g.kd * g.kh * g.kw, groups = Kp / 8;
    for (int k = threadIdx.x; k < Kp; k += kVThreads) {
        int r = k;
        const int c = r % g.C; r /= g.C;
        const int e = r % g.kw; r /= g.kw;
        const int b = r % g.kh;
        const int a = r / g.kh;
        col_info[k] = k < K ? ((a << 24) | (b << 16) | (e << 8) | c) : -1;
    }
    __syncthreads();
    const int64_t items = (int64_t)g.N * g.Do * g.Ho * g.Wo * groups;
    for (int64_t i = (int64_t)blockIdx.x * kVThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kVThreads) {
        const int gr = (int)(i % groups);
        int64_t p = i / groups;
        const int64_t vox = p;
        const int ow = (int)(p % g.Wo); p /= g.Wo;
        const int oh = (int)(p % g.Ho); p /= g.Ho;
        const int od = (int)(p % g.Do);
        const int64_t n = p / g.Do;
        const int d0 = od * g.sd - g.pd, y0 = oh * g.sh - g.ph, x0 = ow * g.sw - g.pw;
        const T* xn = x + n * g.D * g.H * (int64_t)g.W * ldx;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int info = col_info[gr * 8 + j];
            const int d = d0 + (info >> 24), y = y0 + ((info >> 16) & 255), xx = x0 + ((info >> 8) & 255);
            const bool ok = info >= 0 && (unsigned)d < (unsigned)g.D && (unsigned)y < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;
            v[j] = ok ? Elem<T>::to_f(xn[(((int64_t)d * g.H + y) * g.W + xx) * ldx + (info & 255)]) : 0.f;
        }
        Elem<T>::store8(out + vox * ldo + gr * 8, v);
    }
}

// Adjoint of the gather: dx[n, d, y, x, c] = sum over the patches that contain the voxel.  One thread per input voxel
// produces its (<= 8) channels.  Along each axis the taps that reach an integral output coordinate are k0, k0 + s,
// k0 + 2s, ...: the two innermost axes are walked in steps of the stride with a compile-time bound (MAXT taps per
// axis), all loads of one depth tap predicated instead of branched, so that a thread has MAXT*MAXT*C independent
// loads in flight instead of one at a time (the kernel is latency-bound: every element of the patch gradient is
// read exactly once, by a different thread than its neighbours in memory).
template <typename T, int MAXT>
__global__ __launch_bounds__(kVThreads) void col2im3d_kernel(const T* __restrict__ dcol, int64_t ldo, Geo3 g, T* __restrict__ dx,
                                                             int64_t ldx, int accumulate) {
    const int64_t items = (int64_t)g.N * g.D * g.H * g.W;
    for (int64_t i = (int64_t)blockIdx.x * kVThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kVThreads) {
        int64_t p = i;
        const int xx = (int)(p % g.W); p /= g.W;
        const int y = (int)(p % g.H); p /= g.H;
        const int d = (int)(p % g.D);
        const int64_t n = p / g.D;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
        // first tap along h / w with (coordinate + pad - tap) divisible by the stride
        const int b0 = (y + g.ph) % g.sh, e0 = (xx + g.pw) % g.sw;
        for (int a = 0; a < g.kd; ++a) {
            const int td = d + g.pd - a;
            if (td < 0 || td % g.sd) continue;
            const int od = td / g.sd;
            if (od >= g.Do) continue;
            const T* slab = dcol + ((n * g.Do + od) * g.Ho) * (int64_t)g.Wo * ldo + (int64_t)a * g.kh * g.kw * g.C;
#pragma unroll
            for (int jb = 0; jb < MAXT; ++jb) {
                const int b = b0 + jb * g.sh;
                const int oh = (y + g.ph - b) / g.sh;            // exact by construction of b0
                const bool okb = b < g.kh && y + g.ph - b >= 0 && oh < g.Ho;
#pragma unroll
                for (int je = 0; je < MAXT; ++je) {
                    const int e = e0 + je * g.sw;
                    const int ow = (xx + g.pw - e) / g.sw;
                    const bool ok = okb && e < g.kw && xx + g.pw - e >= 0 && ow < g.Wo;
                    const T* src = slab + ((int64_t)(ok ? oh : 0) * g.Wo + (ok ? ow : 0)) * ldo + ((ok ? b : 0) * g.kw + (ok ? e : 0)) * g.C;
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < g.C) acc[c] += ok ? Elem<T>::to_f(src[c]) : 0.f;
                }
            }
        }
        T* dst = dx + i * ldx;
        if (accumulate) {
            float o[8];
            Elem<T>::load8(dst, o);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += o[c];
        }
        Elem<T>::store8(dst, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// MaxPool3d(kernel 3, stride 2, padding 1).  The forward also stores, per output element, which of the 27 window
// positions won (first maximum in depth, row, column scan order, compared at storage precision), so that the
// backward is a gather: an input voxel belongs to one or two windows per axis and takes the gradient of those whose
// stored winner it is.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kVThreads) void maxpool3d_fwd_kernel(const T* __restrict__ x, int64_t ldx, int64_t N, int D, int H, int W,
                                                                  int C, int Do, int Ho, int Wo, T* __restrict__ y, int64_t ldy,
                                                                  unsigned char* __restrict__ arg) {
    const int cv = C / 8;
    const int64_t items = N * Do * Ho * Wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * kVThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kVThreads) {
        const int v = (int)(i % cv);
        int64_t p = i / cv;
        const int64_t vox = p;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho); p /= Ho;
        const int od = (int)(p % Do);
        const int64_t n = p / Do;
        float best[8];
        int win[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; win[j] = -1; }
        for (int a = 0; a < 3; ++a) {
            const int d = 2 * od - 1 + a;
            if ((unsigned)d >= (unsigned)D) continue;
            for (int b = 0; b < 3; ++b) {
                const int yy = 2 * oh - 1 + b;
                if ((unsigned)yy >= (unsigned)H) continue;
                for (int e = 0; e < 3; ++e) {
                    const int xx = 2 * ow - 1 + e;
                    if ((unsigned)xx >= (unsigned)W) continue;
                    float in[8];
                    Elem<T>::load8(x + (((n * D + d) * H + yy) * W + xx) * ldx + v * 8, in);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (win[j] < 0 || in[j] > best[j] || in[j] != in[j]) { best[j] = in[j]; win[j] = (a * 3 + b) * 3 + e; }
                }
            }
        }
        Elem<T>::store8(y + vox * ldy + v * 8, best);
        unsigned long long packed = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) packed |= (unsigned long long)(unsigned)win[j] << (8 * j);
        *reinterpret_cast<unsigned long long*>(arg + vox * C + v * 8) = packed;
    }
}

template <typename T>
__global__ __launch_bounds__(kVThreads) void maxpool3d_bwd_kernel(const T* __restrict__ dy, int64_t ld_dy,
                                                                  const unsigned char* __restrict__ arg, int64_t N, int D, int H, int W,
                                                                  int C, int Do, int Ho, int Wo, T* __restrict__ dx, int64_t ld_dx,
                                                                  int accumulate) {
    const int cv = C / 8;
    const int64_t items = N * D * H * W * cv;
    for (int64_t i = (int64_t)blockIdx.x * kVThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kVThreads) {
        const int v = (int)(i % cv);
        int64_t p = i / cv;
        const int64_t vox = p;
        const int xx = (int)(p % W); p /= W;
        const int yy = (int)(p % H); p /= H;
        const int d = (int)(p % D);
        const int64_t n = p / D;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        // windows along an axis that contain coordinate c: o = c/2 (offset c - 2o + 1), and for odd c also o = (c+1)/2
        for (int od = d >> 1; od <= ((d + 1) >> 1); ++od) {
            if (od >= Do) continue;
            const int a = d - 2 * od + 1;
            for (int oh = yy >> 1; oh <= ((yy + 1) >> 1); ++oh) {
                if (oh >= Ho) continue;
                const int b = yy - 2 * oh + 1;
                for (int ow = xx >> 1; ow <= ((xx + 1) >> 1); ++ow) {
                    if (ow >= Wo) continue;
                    const int e = xx - 2 * ow + 1;
                    const unsigned me = (unsigned)((a * 3 + b) * 3 + e);
                    const int64_t o = ((n * Do + od) * Ho + oh) * Wo + ow;
                    const unsigned long long w = *reinterpret_cast<const unsigned long long*>(arg + o * C + v * 8);
                    float gr[8];
                    Elem<T>::load8(dy + o * ld_dy + v * 8, gr);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (((unsigned)(w >> (8 * j)) & 255u) == me) acc[j] += gr[j];
                }
            }
        }
        T* dst = dx + vox * ld_dx + v * 8;
        if (accumulate) {
            float o[8];
            Elem<T>::load8(dst, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += o[j];
        }
        Elem<T>::store8(dst, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// AvgPool3d(kernel 2, stride 2) (trailing odd planes are dropped, as torch's floor mode does)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kVThreads) void avgpool3d_fwd_kernel(const T* __restrict__ x, int64_t ldx, int64_t N, int D, int H, int W,
                                                                  int C, T* __restrict__ y, int64_t ldy) {
    const int cv = C / 8, Do = D / 2, Ho = H / 2, Wo = W / 2;
    const int64_t items = N * Do * Ho * Wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * kVThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kVThreads) {
        const int v = (int)(i % cv);
        int64_t p = i / cv;
        const int64_t vox = p;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho); p /= Ho;
        const int od = (int)(p % Do);
        const int64_t n = p / Do;
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float in[8];
            Elem<T>::load8(x + (((n * D + 2 * od + (k >> 2)) * H + 2 * oh + ((k >> 1) & 1)) * W + 2 * ow + (k & 1)) * ldx + v * 8, in);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += in[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] *= 0.125f;
        Elem<T>::store8(y + vox * ldy + v * 8, s);
    }
}

template <typename T>
__global__ __launch_bounds__(kVThreads) void avgpool3d_bwd_kernel(const T* __restrict__ dy, int64_t ld_dy, int64_t N, int D, int H, int W,
                                                                  int C, T* __restrict__ dx, int64_t ld_dx, int accumulate) {
    const int cv = C / 8, Do = D / 2, Ho = H / 2, Wo = W / 2;
    const int64_t items = N * D * H * W * cv;
    for (int64_t i = (int64_t)blockIdx.x * kVThreads + threadIdx.x; i < items; i += (int64_t)gridDim.x * kVThreads) {
        const int v = (int)(i % cv);
        int64_t p = i / cv;
        const int64_t vox = p;
        const int xx = (int)(p % W); p /= W;
        const int yy = (int)(p % H); p /= H;
        const int d = (int)(p % D);
        const int64_t n = p / D;
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = 0.f;
        if ((d >> 1) < Do && (yy >> 1) < Ho && (xx >> 1) < Wo) {
            Elem<T>::load8(dy + (((n * Do + (d >> 1)) * Ho + (yy >> 1)) * Wo + (xx >> 1)) * ld_dy + v * 8, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] *= 0.125f;
        }
        T* dst = dx + vox * ld_dx + v * 8;
        if (accumulate) {
            float o[8];
            Elem<T>::load8(dst, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] += o[j];
        }
        Elem<T>::store8(dst, g);
    }
}

// ------------------------------------------------------------------------------ classifier tail
// Global average over the V voxels of every sample ([N, V, ld] slice-major, 8 channels per thread) -> fp32 [N, C]; its
// adjoint spreads dout / V over the voxels.  The tensors behind the last dense block hold a few thousand values.
template <typename T>
__global__ __launch_bounds__(kVThreads) void rows_mean_fwd_kernel(const T* __restrict__ x, int64_t ld, int64_t V, int C,
                                                                  float* __restrict__ out) {
    const int cv = C / 8;
    const int64_t n = blockIdx.x;
    for (int v = threadIdx.x; v < cv; v += kVThreads) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int64_t p = 0; p < V; ++p) {            // fixed order: reproducible
            float a[8];
            Elem<T>::load8(x + (n * V + p) * ld + v * 8, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += a[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) out[n * C + v * 8 + j] = acc[j] / (float)V;
    }
}

template <typename T>
__global__ __launch_bounds__(kVThreads) void rows_mean_bwd_kernel(const float* __restrict__ dout, int64_t V, int C,
                                                                  T* __restrict__ dx, int64_t ld) {
    const int cv = C / 8;
    const int64_t n = blockIdx.y;                            // gridDim.y = N
    for (int64_t i = (int64_t)blockIdx.x * kVThreads + threadIdx.x; i < V * cv; i += (int64_t)gridDim.x * kVThreads) {
        const int v = (int)(i % cv);
        const int64_t p = i / cv;
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = dout[n * C + v * 8 + j] / (float)V;
        Elem<T>::store8(dx + (n * V + p) * ld + v * 8, g);
    }
}

// nn.Linear on a handful of fp32 rows: y[n][o] = b[o] + sum_k x[n][k] * w[o][k].  One workgroup per output feature; the
// threads stride over k with one accumulator per row (rows in chunks of 16), a fixed-order tree adds the threads.
constexpr int kLinRows = 16;
__device__ __forceinline__ float block_sum_f(float v, float* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < kVThreads / 64; ++i) s += sm[i];
    return s;
}

__global__ __launch_bounds__(kVThreads) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, int N, int K, int O,
                                                               float* __restrict__ y) {
    __shared__ float sm[kVThreads / 64];
    const int o = blockIdx.x;
    for (int n0 = 0; n0 < N; n0 += kLinRows) {
        float acc[kLinRows];
#pragma unroll
        for (int r = 0; r < kLinRows; ++r) acc[r] = 0.f;
        for (int k = threadIdx.x; k < K; k += kVThreads) {
            const float wv = w[(int64_t)o * K + k];
#pragma unroll
            for (int r = 0; r < kLinRows; ++r)
                if (n0 + r < N) acc[r] += x[(int64_t)(n0 + r) * K + k] * wv;
        }
#pragma unroll
        for (int r = 0; r < kLinRows; ++r) {
            const float s = block_sum_f(acc[r], sm);
            if (threadIdx.x == 0 && n0 + r < N) y[(int64_t)(n0 + r) * O + o] = s + (b ? b[o] : 0.f);
        }
    }
}

// dw[o][k] = sum_n dy[n][o] * x[n][k] (threads over k), db[o] = sum_n dy[n][o]; one workgroup per output feature
__global__ __launch_bounds__(kVThreads) void linear_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int N,
                                                                 int K, int O, float* __restrict__ dw, float* __restrict__ db,
                                                                 int accumulate) {
    const int o = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += kVThreads) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dy[(int64_t)n * O + o] * x[(int64_t)n * K + k];
        float* dst = dw + (int64_t)o * K + k;
        *dst = accumulate ? *dst + s : s;
    }
    if (db && threadIdx.x == 0) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dy[(int64_t)n * O + o];
        db[o] = accumulate ? db[o] + s : s;
    }
}

// dx[n][k] = sum_o dy[n][o] * w[o][k]; workgroup = (k tile, row), threads over k (coalesced rows of w)
__global__ __launch_bounds__(kVThreads) void linear_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ dy, int K,
                                                                 int O, float* __restrict__ dx) {
    const int n = blockIdx.y, k = blockIdx.x * kVThreads + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s += dy[(int64_t)n * O + o] * w[(int64_t)o * K + k];
    dx[(int64_t)n * K + k] = s;
}

static int check_vol(const char* who, const void* p, int64_t ld, int64_t C, int dtype) {
    if (!p) return fail(SEGK_EINVAL, "%s: null activation pointer", who);
    if (dtype != SEGK_F32 && dtype != SEGK_BF16) return fail(SEGK_EINVAL, "%s: unknown dtype %d", who, dtype);
    if (C <= 0 || C % 8 != 0 || ld < C) return fail(SEGK_ESHAPE, "%s: channel count %lld must be a positive multiple of 8 and <= ld", who, (long long)C);
    if (!aligned16(p) || (ld * esize(dtype)) % 16 != 0) return fail(SEGK_EALIGN, "%s: pointer/ld not 16-byte aligned", who);
    return SEGK_OK;
}
#define SEGK_CHECK_VOL(...)                      \
    do {                                         \
        int rc_ = check_vol(__VA_ARGS__);        \
        if (rc_ != SEGK_OK) return rc_;          \
    } while (0)

static int make_geo(const char* who, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, const int* kernel, const int* stride,
                    const int* pad, Geo3* g) {
    if (!kernel || !stride || !pad) return fail(SEGK_EINVAL, "%s: null geometry", who);
    for (int i = 0; i < 3; ++i)
        if (kernel[i] < 1 || stride[i] < 1 || pad[i] < 0 || pad[i] >= kernel[i])
            return fail(SEGK_ESHAPE, "%s: bad kernel/stride/padding on axis %d", who, i);
    if (N <= 0 || C <= 0 || C > 8 || D <= 0 || H <= 0 || W <= 0) return fail(SEGK_ESHAPE, "%s: bad shape (gathers at most 8 channels)", who);
    g->N = (int)N; g->C = (int)C; g->D = (int)D; g->H = (int)H; g->W = (int)W;
    g->kd = kernel[0]; g->kh = kernel[1]; g->kw = kernel[2];
    g->sd = stride[0]; g->sh = stride[1]; g->sw = stride[2];
    g->pd = pad[0]; g->ph = pad[1]; g->pw = pad[2];
    g->Do = (int)((D + 2 * pad[0] - kernel[0]) / stride[0] + 1);
    g->Ho = (int)((H + 2 * pad[1] - kernel[1]) / stride[1] + 1);
    g->Wo = (int)((W + 2 * pad[2] - kernel[2]) / stride[2] + 1);
    if (g->Do < 1 || g->Ho < 1 || g->Wo < 1) return fail(SEGK_ESHAPE, "%s: empty output", who);
    if (N * D * H * W >= (1ll << 31) || N * g->Do * g->Ho * g->Wo >= (1ll << 31)) return fail(SEGK_ESHAPE, "%s: too many voxels", who);
    return SEGK_OK;
}

}  // namespace segk

using namespace segk;

extern "C" int segk_im2col3d(int dtype, const void* x, int64_t ldx, int64_t C, int64_t N, int64_t D, int64_t H, int64_t W,
                             const int* kernel, const int* stride, const int* pad, void* patches, int64_t ld_patches, int64_t Kp,
                             int dev, void* stream) {
    Geo3 g;
    if (int rc = make_geo("im2col3d", N, C, D, H, W, kernel, stride, pad, &g)) return rc;
    SEGK_CHECK_VOL("im2col3d(x)", x, ldx, 8, dtype);
    SEGK_CHECK_VOL("im2col3d(patches)", patches, ld_patches, Kp, dtype);
    SEGK_REQUIRE(Kp >= C * g.kd * g.kh * g.kw && Kp <= 8192, SEGK_ESHAPE, "im2col3d: Kp %lld must cover C*kd*kh*kw and be at most 8192", (long long)Kp);
    SEGK_REQUIRE(g.kd < 128 && g.kh < 256 && g.kw < 256, SEGK_ESHAPE, "im2col3d: kernel extent too large");
    SEGK_USE_DEVICE(dev);
    const int64_t items = (int64_t)g.N * g.Do * g.Ho * g.Wo * (Kp / 8);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((im2col3d_kernel<T>), dim3(vgrid(items, 16384)), dim3(kVThreads), (size_t)Kp * sizeof(int), (hipStream_t)stream,
                                                     (const T*)x, ldx, g, (T*)patches, ld_patches, (int)Kp));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_col2im3d(int dtype, const void* dpatches, int64_t ld_patches, int64_t C, int64_t N, int64_t D, int64_t H, int64_t W,
                             const int* kernel, const int* stride, const int* pad, void* dx, int64_t ld_dx, int flags, int dev,
                             void* stream) {
    Geo3 g;
    if (int rc = make_geo("col2im3d", N, C, D, H, W, kernel, stride, pad, &g)) return rc;
    SEGK_CHECK_VOL("col2im3d(dx)", dx, ld_dx, 8, dtype);
    SEGK_REQUIRE(dpatches && ld_patches >= C * g.kd * g.kh * g.kw, SEGK_ESHAPE, "col2im3d: bad patch gradient");
    SEGK_USE_DEVICE(dev);
    const int64_t items = (int64_t)g.N * g.D * g.H * g.W;
    // taps per axis that can reach one voxel: ceil(kernel / stride) over the two inner axes
    const int th = (g.kh + g.sh - 1) / g.sh, tw = (g.kw + g.sw - 1) / g.sw, mt = th > tw ? th : tw;
    SEGK_REQUIRE(mt <= 8, SEGK_ESHAPE, "col2im3d: more than 8 taps per axis reach one voxel (kernel/stride too large)");
    if (mt <= 4)
        SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((col2im3d_kernel<T, 4>), dim3(vgrid(items, 65536)), dim3(kVThreads), 0, (hipStream_t)stream,
                                                         (const T*)dpatches, ld_patches, g, (T*)dx, ld_dx, (flags >> 1) & 1));
    else
        SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((col2im3d_kernel<T, 8>), dim3(vgrid(items, 65536)), dim3(kVThreads), 0, (hipStream_t)stream,
                                                         (const T*)dpatches, ld_patches, g, (T*)dx, ld_dx, (flags >> 1) & 1));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_maxpool3d_fwd(int dtype, const void* x, int64_t ldx, int64_t N, int64_t D, int64_t H, int64_t W, int64_t C, void* y,
                                  int64_t ldy, unsigned char* argmax, int dev, void* stream) {
    SEGK_CHECK_VOL("maxpool3d_fwd(x)", x, ldx, C, dtype);
    SEGK_CHECK_VOL("maxpool3d_fwd(y)", y, ldy, C, dtype);
    SEGK_REQUIRE(argmax && (reinterpret_cast<uintptr_t>(argmax) & 7u) == 0, SEGK_EALIGN, "maxpool3d_fwd: argmax must be 8-byte aligned");
    SEGK_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && N * D * H * W < (1ll << 31), SEGK_ESHAPE, "maxpool3d_fwd: bad shape");
    SEGK_USE_DEVICE(dev);
    const int Do = (int)((D - 1) / 2 + 1), Ho = (int)((H - 1) / 2 + 1), Wo = (int)((W - 1) / 2 + 1);
    const int64_t items = N * Do * Ho * Wo * (C / 8);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool3d_fwd_kernel<T>), dim3(vgrid(items)), dim3(kVThreads), 0, (hipStream_t)stream,
                                                     (const T*)x, ldx, N, (int)D, (int)H, (int)W, (int)C, Do, Ho, Wo, (T*)y, ldy, argmax));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_maxpool3d_bwd(int dtype, const void* dy, int64_t ld_dy, const unsigned char* argmax, int64_t N, int64_t D, int64_t H,
                                  int64_t W, int64_t C, void* dx, int64_t ld_dx, int flags, int dev, void* stream) {
    SEGK_CHECK_VOL("maxpool3d_bwd(dy)", dy, ld_dy, C, dtype);
    SEGK_CHECK_VOL("maxpool3d_bwd(dx)", dx, ld_dx, C, dtype);
    SEGK_REQUIRE(argmax && (reinterpret_cast<uintptr_t>(argmax) & 7u) == 0, SEGK_EALIGN, "maxpool3d_bwd: argmax must be 8-byte aligned");
    SEGK_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && N * D * H * W < (1ll << 31), SEGK_ESHAPE, "maxpool3d_bwd: bad shape");
    SEGK_USE_DEVICE(dev);
    const int Do = (int)((D - 1) / 2 + 1), Ho = (int)((H - 1) / 2 + 1), Wo = (int)((W - 1) / 2 + 1);
    const int64_t items = N * D * H * W * (C / 8);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool3d_bwd_kernel<T>), dim3(vgrid(items)), dim3(kVThreads), 0, (hipStream_t)stream,
                                                     (const T*)dy, ld_dy, argmax, N, (int)D, (int)H, (int)W, (int)C, Do, Ho, Wo, (T*)dx, ld_dx,
                                                     (flags >> 1) & 1));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_avgpool3d_fwd(int dtype, const void* x, int64_t ldx, int64_t N, int64_t D, int64_t H, int64_t W, int64_t C, void* y,
                                  int64_t ldy, int dev, void* stream) {
    SEGK_CHECK_VOL("avgpool3d_fwd(x)", x, ldx, C, dtype);
    SEGK_CHECK_VOL("avgpool3d_fwd(y)", y, ldy, C, dtype);
    SEGK_REQUIRE(N > 0 && D >= 2 && H >= 2 && W >= 2 && N * D * H * W < (1ll << 31), SEGK_ESHAPE, "avgpool3d_fwd: every extent must be >= 2");
    SEGK_USE_DEVICE(dev);
    const int64_t items = N * (D / 2) * (H / 2) * (W / 2) * (C / 8);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((avgpool3d_fwd_kernel<T>), dim3(vgrid(items)), dim3(kVThreads), 0, (hipStream_t)stream,
                                                     (const T*)x, ldx, N, (int)D, (int)H, (int)W, (int)C, (T*)y, ldy));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_avgpool3d_bwd(int dtype, const void* dy, int64_t ld_dy, int64_t N, int64_t D, int64_t H, int64_t W, int64_t C, void* dx,
                                  int64_t ld_dx, int flags, int dev, void* stream) {
    SEGK_CHECK_VOL("avgpool3d_bwd(dy)", dy, ld_dy, C, dtype);
    SEGK_CHECK_VOL("avgpool3d_bwd(dx)", dx, ld_dx, C, dtype);
    SEGK_REQUIRE(N > 0 && D >= 2 && H >= 2 && W >= 2 && N * D * H * W < (1ll << 31), SEGK_ESHAPE, "avgpool3d_bwd: every extent must be >= 2");
    SEGK_USE_DEVICE(dev);
    const int64_t items = N * D * H * W * (C / 8);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((avgpool3d_bwd_kernel<T>), dim3(vgrid(items)), dim3(kVThreads), 0, (hipStream_t)stream,
                                                     (const T*)dy, ld_dy, N, (int)D, (int)H, (int)W, (int)C, (T*)dx, ld_dx, (flags >> 1) & 1));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_rows_mean_fwd(int dtype, const void* x, int64_t ld, int64_t N, int64_t V, int64_t C, float* out, int dev,
                                  void* stream) {
    if (int rc = check_vol("rows_mean_fwd", x, ld, C, dtype)) return rc;
    SEGK_REQUIRE(out && N > 0 && N < 65536 && V > 0, SEGK_EINVAL, "rows_mean_fwd: bad arguments");
    SEGK_USE_DEVICE(dev);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((rows_mean_fwd_kernel<T>), dim3((unsigned)N), dim3(kVThreads), 0, (hipStream_t)stream,
                                                     (const T*)x, ld, V, (int)C, out));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_rows_mean_bwd(int dtype, const float* dout, int64_t N, int64_t V, int64_t C, void* dx, int64_t ld, int dev,
                                  void* stream) {
    if (int rc = check_vol("rows_mean_bwd", dx, ld, C, dtype)) return rc;
    SEGK_REQUIRE(dout && N > 0 && N < 65536 && V > 0, SEGK_EINVAL, "rows_mean_bwd: bad arguments");
    SEGK_USE_DEVICE(dev);
    const int gx = vgrid(V * (C / 8), 1024);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((rows_mean_bwd_kernel<T>), dim3((unsigned)gx, (unsigned)N), dim3(kVThreads), 0,
                                                     (hipStream_t)stream, dout, V, (int)C, (T*)dx, ld));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_linear_fwd(const float* x, const float* w, const float* bias, int64_t N, int64_t K, int64_t O, float* y, int dev,
                               void* stream) {
    SEGK_REQUIRE(x && w && y, SEGK_EINVAL, "linear_fwd: null pointer");
    SEGK_REQUIRE(N > 0 && K > 0 && O > 0 && N < (1 << 20) && K < (1 << 24) && O < (1 << 24), SEGK_ESHAPE, "linear_fwd: bad shape");
    SEGK_USE_DEVICE(dev);
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)O), dim3(kVThreads), 0, (hipStream_t)stream, x, w, bias, (int)N, (int)K, (int)O, y);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_linear_bwd(const float* x, const float* w, const float* dy, int64_t N, int64_t K, int64_t O, float* dx, float* dw,
                               float* dbias, int flags, int dev, void* stream) {
    SEGK_REQUIRE(x && w && dy && dw, SEGK_EINVAL, "linear_bwd: null pointer");
    SEGK_REQUIRE(N > 0 && N < 65536 && K > 0 && O > 0 && K < (1 << 24) && O < (1 << 24), SEGK_ESHAPE, "linear_bwd: bad shape");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3((unsigned)O), dim3(kVThreads), 0, st, x, dy, (int)N, (int)K, (int)O, dw, dbias, (flags >> 1) & 1);
    SEGK_LAUNCH_CHECK();
    if (dx) {
        hipLaunchKernelGGL(linear_dgrad_kernel, dim3((unsigned)((K + kVThreads - 1) / kVThreads), (unsigned)N), dim3(kVThreads), 0, st, w, dy,
                           (int)K, (int)O, dx);
        SEGK_LAUNCH_CHECK();
    }
    return SEGK_OK;
}
