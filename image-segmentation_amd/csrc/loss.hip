// Fused softmax + squared-Dice + focal loss (forward partial sums, finalize, backward) and the
// arg-max Dice metric.  HBM-bound streams: every logit is read once per pass, reductions go
// wavefront shuffle -> LDS -> per-block partial -> fixed-order finalize (deterministic).
#include "common.h"
#include <mutex>

namespace segk {

constexpr int kMaxC = 8;          // classes supported by the register-resident per-pixel vector
constexpr int kLossThreads = 256;
constexpr int kLossVec = 4;       // pixels per thread per iteration on the vector path

__device__ __forceinline__ float pow_gamma(float x, float gamma) {
    if (gamma == 2.f) return x * x;
    if (gamma == 1.f) return x;
    if (gamma == 0.f) return 1.f;
    return x > 0.f ? powf(x, gamma) : 0.f;
}
// d/dx x^gamma, with torch's convention that the derivative is 0 for gamma == 0
__device__ __forceinline__ float dpow_gamma(float x, float gamma) {
    if (gamma == 2.f) return 2.f * x;
    if (gamma == 1.f) return 1.f;
    if (gamma == 0.f) return 0.f;
    return gamma * powf(x, gamma - 1.f);
}

template <typename TL>
__device__ __forceinline__ float ld_logit(const TL* p) { return Elem<TL>::to_f(*p); }

// Per-pixel quantities shared by forward and backward.
struct Pix {
    float p[kMaxC];
    float logp_t;  // log-softmax of the target class
};

template <int C>
__device__ __forceinline__ void softmax_pixel(const float (&z)[kMaxC], int t, bool no_sft, Pix& o) {
    if (no_sft) {
#pragma unroll
        for (int c = 0; c < C; ++c) o.p[c] = z[c];
        o.logp_t = 0.f;
        return;
    }
    float m = z[0];
#pragma unroll
    for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
    float e[kMaxC], s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { e[c] = expf(z[c] - m); s += e[c]; }
    const float inv = 1.f / s;
    float zt = z[0];
#pragma unroll
    for (int c = 0; c < C; ++c) { o.p[c] = e[c] * inv; if (c == t) zt = z[c]; }
    o.logp_t = (zt - m) - logf(s);
}

// ----------------------------------------------------------------------------------------------
// forward: per-block partials   ws[(n*bpi + b)*(4*C+1) + ...]
// ----------------------------------------------------------------------------------------------
template <typename TL, int C, bool VEC>
__global__ __launch_bounds__(kLossThreads) void dice_focal_fwd_kernel(
    const TL* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ hot,
    int64_t HW, float gamma, const float* __restrict__ alpha, int no_sft, float* __restrict__ focal_map,
    float* __restrict__ ws, int bpi, int sa) {
    const int n = blockIdx.y, b = blockIdx.x;
    const TL* zl = logits + (int64_t)n * C * HW;
    const int64_t* lab = labels ? labels + (int64_t)n * HW : nullptr;
    const float* ht = hot ? hot + (int64_t)n * C * HW : nullptr;
    float* fm = focal_map ? focal_map + (int64_t)n * C * HW : nullptr;
    constexpr int V = VEC ? kLossVec : 1;
    float accA[kMaxC], accB[kMaxC], accG[kMaxC], accF[kMaxC];
#pragma unroll
    for (int c = 0; c < C; ++c) accA[c] = accB[c] = accG[c] = accF[c] = 0.f;
    int fg = 0;
    float al[kMaxC];
#pragma unroll
    for (int c = 0; c < C; ++c) al[c] = alpha ? alpha[c] : 1.f;

    const int64_t chunk = (int64_t)kLossThreads * V;
    for (int64_t base = (int64_t)b * chunk + (int64_t)threadIdx.x * V; base < HW; base += (int64_t)bpi * chunk) {
        float z[V][kMaxC];
        int t[V];
        float g[V][kMaxC];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if constexpr (VEC) {
                float tmp[4];
                Elem<TL>::load4(zl + (int64_t)c * HW + base, tmp);
#pragma unroll
                for (int v = 0; v < V; ++v) z[v][c] = tmp[v];
            } else {
                z[0][c] = ld_logit(zl + (int64_t)c * HW + base);
            }
        }
        if (lab) {
#pragma unroll
            for (int v = 0; v < V; ++v) t[v] = (int)lab[base + v];
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) t[v] = -1;
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) g[v][c] = ht[(int64_t)c * HW + base + v];
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            Pix px;
            softmax_pixel<C>(z[v], t[v], no_sft != 0, px);
            if (lab) {
                fg |= (t[v] != 0);
                float at = al[0];
#pragma unroll
                for (int c = 0; c < C; ++c) if (c == t[v]) at = al[c];
                const float ce = -at * px.logp_t;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float gc = (c == t[v]) ? 1.f : 0.f;
                    // self-adjusting Dice (loss/dice.py:275-325) weights both sums by 1 - p
                    accA[c] += sa ? px.p[c] * gc * (1.f - px.p[c]) : px.p[c] * gc;
                    accB[c] += sa ? px.p[c] * (1.f - px.p[c]) : px.p[c] * px.p[c];
                    accG[c] += gc;
                    // focal weight exactly as the reference forms it: 1 - where(onehot==1, p, 1-p)
                    const float fw = 1.f - ((c == t[v]) ? px.p[c] : (1.f - px.p[c]));
                    const float f = pow_gamma(fw, gamma) * 1.f * ce;
                    accF[c] += f;
                    if (fm) fm[(int64_t)c * HW + base + v] = f;
                }
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    fg |= (g[v][c] != 0.f);
                    accA[c] += sa ? px.p[c] * g[v][c] * (1.f - px.p[c]) : px.p[c] * g[v][c];
                    accB[c] += sa ? px.p[c] * (1.f - px.p[c]) : px.p[c] * px.p[c];
                    accG[c] += sa ? g[v][c] : g[v][c] * g[v][c];
                }
            }
        }
    }
    // block reduction: shuffle within the wave, then LDS across the 4 waves
    __shared__ float red[kLossThreads / 64][4 * kMaxC + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float a = wave_sum(accA[c]), bb = wave_sum(accB[c]), gg = wave_sum(accG[c]), ff = wave_sum(accF[c]);
        if (lane == 0) { red[wv][4 * c] = a; red[wv][4 * c + 1] = bb; red[wv][4 * c + 2] = gg; red[wv][4 * c + 3] = ff; }
    }
    const int anyfg = __any(fg) ? 1 : 0;
    if (lane == 0) red[wv][4 * kMaxC] = (float)anyfg;
    __syncthreads();
    float* out = ws + ((int64_t)n * bpi + b) * (4 * C + 1);
    if (threadIdx.x < 4 * C) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kLossThreads / 64; ++w) s += red[w][threadIdx.x];
        out[threadIdx.x] = s;
    } else if (threadIdx.x == 4 * C) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kLossThreads / 64; ++w) s += red[w][4 * kMaxC];
        out[4 * C] = s;
    }
}

// one block per image: fixed-order sum of the block partials in double
__global__ __launch_bounds__(64) void dice_focal_collect_kernel(const float* __restrict__ ws, int bpi, int C,
                                                                float* __restrict__ sums, int32_t* __restrict__ flags) {
    const int n = blockIdx.x, k = threadIdx.x;
    const int per = 4 * C + 1;
    if (k >= per) return;
    double s = 0.0;
    for (int b = 0; b < bpi; ++b) s += (double)ws[((int64_t)n * bpi + b) * per + k];
    if (k < 4 * C) sums[(int64_t)n * 4 * C + k] = (float)s;
    else flags[n] = s > 0.0 ? 1 : 0;
}

// scalars: [0] dice coefficient (loss = 1 - it), [1] focal loss, [2] 1/div (0 if nothing selected), [3] focal scale
__global__ __launch_bounds__(64) void dice_focal_finalize_kernel(const float* __restrict__ sums,
                                                                 const int32_t* __restrict__ flags, int N, int C,
                                                                 int64_t HW, int all_chann, int focal_sum,
                                                                 float* __restrict__ scalars) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int c0 = (all_chann || C == 1) ? 0 : 1;
    const int per_img = C - c0;
    int nsel = 0;
    for (int n = 0; n < N; ++n) nsel += flags[n] ? 1 : 0;
    // reference: div = numel(kept table) unless (some image has foreground and C > 1)
    float div = (float)(N * per_img);
    if (nsel > 0 && C > 1) div = (float)(nsel * per_img);
    float total = 0.f;
    double focal = 0.0;
    for (int n = 0; n < N; ++n) {
        for (int c = 0; c < C; ++c) {
            const float* s = sums + ((int64_t)n * C + c) * 4;
            focal += (double)s[3];
            if (c >= c0 && flags[n]) total += (2.f * s[0] + 1.f) / (s[1] + s[2] + 1.f);
        }
    }
    const float fscale = focal_sum ? 1.f : (float)(1.0 / ((double)N * C * (double)HW));
    scalars[0] = total / div;
    scalars[1] = (float)(focal * (double)fscale);
    scalars[2] = nsel > 0 ? 1.f / div : 0.f;
    scalars[3] = fscale;
}

// ----------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------
template <typename TL, int C, bool VEC>
__global__ __launch_bounds__(kLossThreads) void dice_focal_bwd_kernel(
    const TL* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ hot,
    int64_t HW, float gamma, const float* __restrict__ alpha, int no_sft, int all_chann,
    const float* __restrict__ sums, const int32_t* __restrict__ flags, const float* __restrict__ scalars,
    const float* __restrict__ gscale, TL* __restrict__ dlogits, int bpi, int sa) {
    const int n = blockIdx.y, b = blockIdx.x;
    const TL* zl = logits + (int64_t)n * C * HW;
    TL* dz = dlogits + (int64_t)n * C * HW;
    const int64_t* lab = labels ? labels + (int64_t)n * HW : nullptr;
    const float* ht = hot ? hot + (int64_t)n * C * HW : nullptr;
    constexpr int V = VEC ? kLossVec : 1;
    const float gs_d = gscale[0], gs_f = gscale[1] * scalars[3];
    const float inv_div = scalars[2];
    const int c0 = (all_chann || C == 1) ? 0 : 1;
    // u_c = k1_c * g_c + k2_c * p_c   (d(dice loss)/dp_c), zero for dropped channels / unselected images
    float k1[kMaxC], k2[kMaxC], al[kMaxC];
    const bool sel = flags[n] != 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float* s = sums + ((int64_t)n * C + c) * 4;
        const float den = s[1] + s[2] + 1.f;
        const bool on = sel && c >= c0;
        k1[c] = on ? -inv_div * 2.f / den : 0.f;
        k2[c] = on ? inv_div * 2.f * (2.f * s[0] + 1.f) / (den * den) : 0.f;
        al[c] = alpha ? alpha[c] : 1.f;
    }
    const int64_t chunk = (int64_t)kLossThreads * V;
    for (int64_t base = (int64_t)b * chunk + (int64_t)threadIdx.x * V; base < HW; base += (int64_t)bpi * chunk) {
        float z[V][kMaxC], g[V][kMaxC], o[kMaxC][V];
        int t[V];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if constexpr (VEC) {
                float tmp[4];
                Elem<TL>::load4(zl + (int64_t)c * HW + base, tmp);
#pragma unroll
                for (int v = 0; v < V; ++v) z[v][c] = tmp[v];
            } else {
                z[0][c] = ld_logit(zl + (int64_t)c * HW + base);
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            t[v] = lab ? (int)lab[base + v] : -1;
#pragma unroll
            for (int c = 0; c < C; ++c) g[v][c] = lab ? ((c == t[v]) ? 1.f : 0.f) : ht[(int64_t)c * HW + base + v];
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            Pix px;
            softmax_pixel<C>(z[v], t[v], no_sft != 0, px);
            float u[kMaxC], up = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                // d(num)/dp = g, d(den)/dp = 2p;  self-adjusting: g (1 - 2p) and (1 - 2p)
                u[c] = sa ? gs_d * (1.f - 2.f * px.p[c]) * (k1[c] * g[v][c] + 0.5f * k2[c]) : gs_d * (k1[c] * g[v][c] + k2[c] * px.p[c]);
                up += u[c] * px.p[c];
            }
            float s_[kMaxC], sp = 0.f, S = 0.f, ce = 0.f, at = 0.f;
            const bool do_focal = lab != nullptr && gs_f != 0.f;
            if (do_focal) {
                at = al[0];
#pragma unroll
                for (int c = 0; c < C; ++c) if (c == t[v]) at = al[c];
                ce = -at * px.logp_t;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const bool tc = (c == t[v]);
                    const float fw = 1.f - (tc ? px.p[c] : (1.f - px.p[c]));
                    S += pow_gamma(fw, gamma);
                    s_[c] = (tc ? -1.f : 1.f) * dpow_gamma(fw, gamma);
                    sp += s_[c] * px.p[c];
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float d = no_sft ? u[c] : px.p[c] * (u[c] - up);
                if (do_focal) d += gs_f * (S * at * (px.p[c] - g[v][c]) + ce * px.p[c] * (s_[c] - sp));
                o[c][v] = d;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if constexpr (VEC) {
                Elem<TL>::store4(dz + (int64_t)c * HW + base, o[c]);
            } else {
                dz[(int64_t)c * HW + base] = Elem<TL>::from_f(o[c][0]);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// arg-max metric: mask + {sum phat*g, sum phat, sum g} per (n,c).  argmax of the logits equals the
// argmax of their softmax (monotone), first maximum wins on ties like torch.argmax.
// ----------------------------------------------------------------------------------------------
template <typename TL, int C>
__global__ __launch_bounds__(kLossThreads) void argmax_dice_kernel(const TL* __restrict__ logits,
                                                                   const int64_t* __restrict__ labels, int64_t HW,
                                                                   uint8_t* __restrict__ mask,
                                                                   float* __restrict__ counts,
                                                                   int32_t* __restrict__ flags) {
    const int n = blockIdx.x;
    const TL* zl = logits + (int64_t)n * C * HW;
    const int64_t* lab = labels + (int64_t)n * HW;
    int inter[kMaxC], np_[kMaxC], ng[kMaxC], fg = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) inter[c] = np_[c] = ng[c] = 0;
    for (int64_t i = threadIdx.x; i < HW; i += kLossThreads) {
        float best = ld_logit(zl + i);
        int arg = 0;
#pragma unroll
        for (int c = 1; c < C; ++c) {
            const float v = ld_logit(zl + (int64_t)c * HW + i);
            if (v > best) { best = v; arg = c; }
        }
        const int t = (int)lab[i];
        fg |= (t != 0);
        if (mask) mask[(int64_t)n * HW + i] = (uint8_t)arg;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            inter[c] += (arg == c && t == c);
            np_[c] += (arg == c);
            ng[c] += (t == c);
        }
    }
    __shared__ int red[kLossThreads / 64][3 * kMaxC + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        int a = inter[c], b = np_[c], g = ng[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); g += __shfl_xor(g, o, 64); }
        if (lane == 0) { red[wv][3 * c] = a; red[wv][3 * c + 1] = b; red[wv][3 * c + 2] = g; }
    }
    const int anyfg = __any(fg) ? 1 : 0;
    if (lane == 0) red[wv][3 * kMaxC] = anyfg;
    __syncthreads();
    if (threadIdx.x < 3 * C) {
        int s = 0;
        for (int w = 0; w < kLossThreads / 64; ++w) s += red[w][threadIdx.x];
        counts[(int64_t)n * 3 * C + threadIdx.x] = (float)s;
    } else if (threadIdx.x == 3 * C) {
        int s = 0;
        for (int w = 0; w < kLossThreads / 64; ++w) s += red[w][3 * kMaxC];
        flags[n] = s > 0;
    }
}

static int loss_bpi(int64_t N, int64_t HW) {
    // enough blocks to fill 256 CUs a few times over, at least ~2K pixels per block
    int64_t bpi = (HW + 2047) / 2048;
    const int64_t cap = (4096 + N - 1) / N;
    if (bpi > cap) bpi = cap;
    if (bpi < 1) bpi = 1;
    return (int)bpi;
}

template <typename TL, int C>
static int launch_fwd(const void* logits, const int64_t* labels, const float* hot, int64_t N, int64_t HW, float gamma,
                      const float* alpha, int flags, float* focal_map, float* ws, int bpi, hipStream_t st) {
    const bool vec = (HW % 4 == 0) && aligned16(logits) && (!labels || aligned16(labels)) && (!hot || aligned16(hot)) &&
                     (!focal_map || aligned16(focal_map));
    dim3 grid(bpi, (unsigned)N);
    if (vec)
        hipLaunchKernelGGL((dice_focal_fwd_kernel<TL, C, true>), grid, dim3(kLossThreads), 0, st, (const TL*)logits, labels,
                           hot, HW, gamma, alpha, flags & 1, focal_map, ws, bpi, (flags >> 3) & 1);
    else
        hipLaunchKernelGGL((dice_focal_fwd_kernel<TL, C, false>), grid, dim3(kLossThreads), 0, st, (const TL*)logits, labels,
                           hot, HW, gamma, alpha, flags & 1, focal_map, ws, bpi, (flags >> 3) & 1);
    return 0;
}

template <typename TL, int C>
static int launch_bwd(const void* logits, const int64_t* labels, const float* hot, int64_t N, int64_t HW, float gamma,
                      const float* alpha, int flags, const float* sums, const int32_t* iflags, const float* scalars,
                      const float* gscale, void* dlogits, int bpi, hipStream_t st) {
    const bool vec = (HW % 4 == 0) && aligned16(logits) && aligned16(dlogits) && (!labels || aligned16(labels)) &&
                     (!hot || aligned16(hot));
    dim3 grid(bpi, (unsigned)N);
    if (vec)
        hipLaunchKernelGGL((dice_focal_bwd_kernel<TL, C, true>), grid, dim3(kLossThreads), 0, st, (const TL*)logits, labels,
                           hot, HW, gamma, alpha, flags & 1, (flags >> 1) & 1, sums, iflags, scalars, gscale, (TL*)dlogits, bpi, (flags >> 3) & 1);
    else
        hipLaunchKernelGGL((dice_focal_bwd_kernel<TL, C, false>), grid, dim3(kLossThreads), 0, st, (const TL*)logits, labels,
                           hot, HW, gamma, alpha, flags & 1, (flags >> 1) & 1, sums, iflags, scalars, gscale, (TL*)dlogits, bpi, (flags >> 3) & 1);
    return 0;
}

#define SEGK_DISPATCH_C(C_, ...)                                                          \
    switch (C_) {                                                                         \
        case 1: { constexpr int CC = 1; __VA_ARGS__; } break;                             \
        case 2: { constexpr int CC = 2; __VA_ARGS__; } break;                             \
        case 3: { constexpr int CC = 3; __VA_ARGS__; } break;                             \
        case 4: { constexpr int CC = 4; __VA_ARGS__; } break;                             \
        case 5: { constexpr int CC = 5; __VA_ARGS__; } break;                             \
        case 6: { constexpr int CC = 6; __VA_ARGS__; } break;                             \
        case 7: { constexpr int CC = 7; __VA_ARGS__; } break;                             \
        case 8: { constexpr int CC = 8; __VA_ARGS__; } break;                             \
        default: return segk::fail(SEGK_ESHAPE, "class count %lld not in 1..8", (long long)(C_)); \
    }

}  // namespace segk

using namespace segk;

extern "C" size_t segk_dice_focal_workspace(int64_t N, int64_t C, int64_t HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return align_up((size_t)N * loss_bpi(N, HW) * (4 * C + 1) * sizeof(float), 256);
}

extern "C" int segk_dice_focal_fwd(const void* logits, int logits_dtype, const int64_t* labels, const float* hot,
                                   int64_t N, int64_t C, int64_t HW, float gamma, const float* alpha, int flags,
                                   float* out_sums, int32_t* out_flags, float* focal_map, void* workspace,
                                   size_t workspace_bytes, int dev, void* stream) {
    SEGK_REQUIRE(logits && out_sums && out_flags && workspace, SEGK_EINVAL, "dice_focal_fwd: null pointer");
    SEGK_REQUIRE((labels != nullptr) != (hot != nullptr), SEGK_EINVAL, "dice_focal_fwd: exactly one of labels/hot");
    SEGK_REQUIRE(N > 0 && HW > 0 && N < 65536, SEGK_ESHAPE, "dice_focal_fwd: bad N=%lld HW=%lld", (long long)N, (long long)HW);
    SEGK_REQUIRE(workspace_bytes >= segk_dice_focal_workspace(N, C, HW), SEGK_EWORKSPACE, "dice_focal_fwd: workspace %zu too small", workspace_bytes);
    SEGK_REQUIRE(!focal_map || labels, SEGK_EINVAL, "dice_focal_fwd: focal map needs integer labels");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int bpi = loss_bpi(N, HW);
    SEGK_DISPATCH_C(C, SEGK_DISPATCH_DTYPE(logits_dtype, TL, (launch_fwd<TL, CC>(logits, labels, hot, N, HW, gamma, alpha, flags,
                                                                                  focal_map, (float*)workspace, bpi, st))));
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(dice_focal_collect_kernel, dim3((unsigned)N), dim3(64), 0, st, (const float*)workspace, bpi, (int)C,
                       out_sums, out_flags);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// ----------------------------------------------------------------------------------------------
// The same forward for callers that cannot allocate (HIP graph capture, tight serving loops): the block partials go to
// a fixed device array of this code object instead of a caller-provided workspace.  `scratch` is only a host-side
// stand-in for the symbol; the device address is looked up once per device with hipGetSymbolAddress and that pointer
// is what the kernels receive.  Every GPU holds its own copy of the array until the process ends.  That look-up (the
// first call on a device) is not a capturable operation: callers warm the entry point up once per device before capturing.
// ----------------------------------------------------------------------------------------------
namespace segk {
constexpr size_t kStaticScratchBytes = 1u << 20;     // 1 MiB; what a shape needs: segk_dice_focal_workspace(N, C, HW)
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
static void* g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mutex;

// device address of `scratch` on device `dev`, which must be the calling thread's current device
static int static_scratch(int dev, void** out) {
    SEGK_REQUIRE(dev >= 0 && dev < kMaxDevices, SEGK_EINVAL, "static scratch: device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        SEGK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        SEGK_REQUIRE(ptr != nullptr, SEGK_EHIP, "static scratch: hipGetSymbolAddress returned a null pointer");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return SEGK_OK;
}
}  // namespace segk

extern "C" int segk_dice_focal_fwd_static(const void* logits, int logits_dtype, const int64_t* labels, const float* hot,
                                          int64_t N, int64_t C, int64_t HW, float gamma, const float* alpha, int flags,
                                          float* out_sums, int32_t* out_flags, float* focal_map, int dev, void* stream) {
    SEGK_REQUIRE(logits && out_sums && out_flags, SEGK_EINVAL, "dice_focal_fwd_static: null pointer");
    SEGK_REQUIRE((labels != nullptr) != (hot != nullptr), SEGK_EINVAL, "dice_focal_fwd_static: exactly one of labels/hot");
    SEGK_REQUIRE(N > 0 && HW > 0 && N < 65536 && C >= 1 && C <= kMaxC, SEGK_ESHAPE, "dice_focal_fwd_static: bad N=%lld C=%lld HW=%lld",
                 (long long)N, (long long)C, (long long)HW);
    const size_t need = segk_dice_focal_workspace(N, C, HW);
    SEGK_REQUIRE(need <= kStaticScratchBytes, SEGK_EWORKSPACE,
                 "dice_focal_fwd_static: this shape needs %zu bytes of block partials, the static scratch holds %zu; use "
                 "segk_dice_focal_fwd with a workspace", need, kStaticScratchBytes);
    SEGK_REQUIRE(!focal_map || labels, SEGK_EINVAL, "dice_focal_fwd_static: focal map needs integer labels");
    SEGK_USE_DEVICE(dev);
    void* ws = nullptr;
    if (int rc = static_scratch(dev, &ws)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int bpi = loss_bpi(N, HW);
    // every block writes all of its 4*C+1 partial slots and the collect kernel reads exactly the slots of this launch,
    // so nothing in the scratch has to start at zero and a previous call's (larger) contents are never read
    SEGK_DISPATCH_C(C, SEGK_DISPATCH_DTYPE(logits_dtype, TL, (launch_fwd<TL, CC>(logits, labels, hot, N, HW, gamma, alpha, flags,
                                                                                  focal_map, (float*)ws, bpi, st))));
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(dice_focal_collect_kernel, dim3((unsigned)N), dim3(64), 0, st, (const float*)ws, bpi, (int)C, out_sums,
                       out_flags);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_dice_focal_finalize(const float* sums, const int32_t* img_flags, int64_t N, int64_t C, int64_t HW,
                                        int flags, float* out_scalars, int dev, void* stream) {
    SEGK_REQUIRE(sums && img_flags && out_scalars, SEGK_EINVAL, "dice_focal_finalize: null pointer");
    SEGK_REQUIRE(N > 0 && C > 0 && C <= kMaxC && HW > 0, SEGK_ESHAPE, "dice_focal_finalize: bad shape");
    SEGK_USE_DEVICE(dev);
    hipLaunchKernelGGL(dice_focal_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, img_flags, (int)N, (int)C,
                       HW, (flags >> 1) & 1, (flags >> 2) & 1, out_scalars);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_dice_focal_bwd(const void* logits, int logits_dtype, const int64_t* labels, const float* hot,
                                   int64_t N, int64_t C, int64_t HW, float gamma, const float* alpha, int flags,
                                   const float* sums, const int32_t* img_flags, const float* scalars,
                                   const float* gscale, void* dlogits, int dev, void* stream) {
    SEGK_REQUIRE(logits && sums && img_flags && scalars && gscale && dlogits, SEGK_EINVAL, "dice_focal_bwd: null pointer");
    SEGK_REQUIRE((labels != nullptr) != (hot != nullptr), SEGK_EINVAL, "dice_focal_bwd: exactly one of labels/hot");
    SEGK_REQUIRE(N > 0 && HW > 0 && N < 65536, SEGK_ESHAPE, "dice_focal_bwd: bad N/HW");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int bpi = loss_bpi(N, HW);
    SEGK_DISPATCH_C(C, SEGK_DISPATCH_DTYPE(logits_dtype, TL, (launch_bwd<TL, CC>(logits, labels, hot, N, HW, gamma, alpha, flags, sums,
                                                                                  img_flags, scalars, gscale, dlogits, bpi, st))));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_argmax_dice(const void* logits, int logits_dtype, const int64_t* labels, int64_t N, int64_t C,
                                int64_t HW, uint8_t* mask, float* counts, int32_t* out_flags, int dev, void* stream) {
    SEGK_REQUIRE(logits && labels && counts && out_flags, SEGK_EINVAL, "argmax_dice: null pointer");
    SEGK_REQUIRE(N > 0 && HW > 0, SEGK_ESHAPE, "argmax_dice: bad N/HW");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    SEGK_DISPATCH_C(C, SEGK_DISPATCH_DTYPE(logits_dtype, TL,
                                           hipLaunchKernelGGL((argmax_dice_kernel<TL, CC>), dim3((unsigned)N), dim3(kLossThreads), 0, st,
                                                              (const TL*)logits, labels, HW, mask, counts, out_flags)));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}
