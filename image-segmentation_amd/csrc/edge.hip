// Edge maps for the edge-aware Dice loss (loss/hausdorff.py:71-128): per channel, the two Sobel derivatives of an
// fp32 NCHW tensor, each instance-normalised over its plane (biased variance, eps 1e-5, no affine), rectified in
// both directions and summed.  Since InstanceNorm(-s) = -InstanceNorm(s), relu(IN(s)) + relu(IN(-s)) = |IN(s)|:
//     edge = |(sx - mean sx) * rstd sx| + |(sy - mean sy) * rstd sy|
// All kernels are plane-local streams (a plane is a few hundred KB and stays in L2 between the statistics pass and
// the pass that uses them); the derivatives are recomputed from the input instead of being stored.
#include "common.h"

namespace segk {

constexpr int kEdgeThreads = 256;

// zero-padded 3x3 cross-correlation with the Sobel pair of loss/hausdorff.py:85-88 (x1 and y1; x2 = -x1, y2 = -y1)
__device__ __forceinline__ void sobel_at(const float* __restrict__ x, int H, int W, int h, int w, float& sx, float& sy) {
    float v[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int hh = h + i - 1, ww = w + j - 1;
            v[i][j] = ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) ? x[(int64_t)hh * W + ww] : 0.f;
        }
    sx = (v[0][2] - v[0][0]) + 2.f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0]);
    sy = (v[0][0] + 2.f * v[0][1] + v[0][2]) - (v[2][0] + 2.f * v[2][1] + v[2][2]);
}

// block-wide sum of four values -> out[0..3] (one partial row per block)
__device__ __forceinline__ void block_sum4(float a, float b, float c, float d, float* __restrict__ out) {
    __shared__ float red[kEdgeThreads / 64][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d);
    if (lane == 0) { red[wv][0] = a; red[wv][1] = b; red[wv][2] = c; red[wv][3] = d; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kEdgeThreads / 64; ++k) s += red[k][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

// pass 1: partial sums of sx, sx^2, sy, sy^2 per (plane, split)
__global__ __launch_bounds__(kEdgeThreads) void edge_stats_partial_kernel(const float* __restrict__ x, int H, int W, int splits,
                                                                         float* __restrict__ ws) {
    const int plane = blockIdx.y, s = blockIdx.x;
    const float* xp = x + (int64_t)plane * H * W;
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
    for (int i = s * kEdgeThreads + threadIdx.x; i < H * W; i += splits * kEdgeThreads) {
        float sx, sy;
        sobel_at(xp, H, W, i / W, i % W, sx, sy);
        a += sx; b += sx * sx; c += sy; d += sy * sy;
    }
    block_sum4(a, b, c, d, ws + ((int64_t)plane * splits + s) * 4);
}

// stats[plane] = (mean sx, rstd sx, mean sy, rstd sy); fixed-order sum of the partials in double
__global__ __launch_bounds__(kEdgeThreads) void edge_stats_finalize_kernel(const float* __restrict__ ws, int planes, int splits, int HW,
                                                                          float eps, float* __restrict__ stats) {
    const int plane = blockIdx.x * kEdgeThreads + threadIdx.x;
    if (plane >= planes) return;
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    for (int s = 0; s < splits; ++s)
        for (int k = 0; k < 4; ++k) t[k] += (double)ws[((int64_t)plane * splits + s) * 4 + k];
    for (int k = 0; k < 2; ++k) {
        const double mean = t[2 * k] / HW;
        double var = t[2 * k + 1] / HW - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[plane * 4 + 2 * k] = (float)mean;
        stats[plane * 4 + 2 * k + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// pass 2: the edge map
__global__ __launch_bounds__(kEdgeThreads) void edge_apply_kernel(const float* __restrict__ x, int H, int W, int splits,
                                                                 const float* __restrict__ stats, float* __restrict__ edge) {
    const int plane = blockIdx.y, s = blockIdx.x;
    const float* xp = x + (int64_t)plane * H * W;
    float* ep = edge + (int64_t)plane * H * W;
    const float mx = stats[plane * 4], rx = stats[plane * 4 + 1], my = stats[plane * 4 + 2], ry = stats[plane * 4 + 3];
    for (int i = s * kEdgeThreads + threadIdx.x; i < H * W; i += splits * kEdgeThreads) {
        float sx, sy;
        sobel_at(xp, H, W, i / W, i % W, sx, sy);
        ep[i] = fabsf((sx - mx) * rx) + fabsf((sy - my) * ry);
    }
}

__device__ __forceinline__ float sign_of(float u) { return u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f); }

// backward pass 1: with u = normalised derivative and du = sign(u) * dedge, the InstanceNorm backward needs the plane
// sums of du and du*u (for both derivatives)
__global__ __launch_bounds__(kEdgeThreads) void edge_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dedge,
                                                                       int H, int W, int splits, const float* __restrict__ stats,
                                                                       float* __restrict__ ws) {
    const int plane = blockIdx.y, s = blockIdx.x;
    const float* xp = x + (int64_t)plane * H * W;
    const float* dp = dedge + (int64_t)plane * H * W;
    const float mx = stats[plane * 4], rx = stats[plane * 4 + 1], my = stats[plane * 4 + 2], ry = stats[plane * 4 + 3];
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
    for (int i = s * kEdgeThreads + threadIdx.x; i < H * W; i += splits * kEdgeThreads) {
        float sx, sy;
        sobel_at(xp, H, W, i / W, i % W, sx, sy);
        const float ux = (sx - mx) * rx, uy = (sy - my) * ry, g = dp[i];
        const float dux = sign_of(ux) * g, duy = sign_of(uy) * g;
        a += dux; b += dux * ux; c += duy; d += duy * uy;
    }
    block_sum4(a, b, c, d, ws + ((int64_t)plane * splits + s) * 4);
}

// bstats[plane] = (mean du_x, mean du_x*u_x, mean du_y, mean du_y*u_y)
__global__ __launch_bounds__(kEdgeThreads) void edge_bwd_finalize_kernel(const float* __restrict__ ws, int planes, int splits, int HW,
                                                                        float* __restrict__ bstats) {
    const int plane = blockIdx.x * kEdgeThreads + threadIdx.x;
    if (plane >= planes) return;
    for (int k = 0; k < 4; ++k) {
        double t = 0.0;
        for (int s = 0; s < splits; ++s) t += (double)ws[((int64_t)plane * splits + s) * 4 + k];
        bstats[plane * 4 + k] = (float)(t / HW);
    }
}

// backward pass 2: dx(r) = sum over the 3x3 positions q that read r of  Kx[r - q] * dsx(q) + Ky[r - q] * dsy(q), where
// dsx(q) = rstd_x * (du_x(q) - mean du_x - u_x(q) * mean(du_x u_x)) is the InstanceNorm backward at q.
__global__ __launch_bounds__(kEdgeThreads) void edge_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dedge, int H, int W,
                                                               int splits, const float* __restrict__ stats,
                                                               const float* __restrict__ bstats, float* __restrict__ dx) {
    const int plane = blockIdx.y, s = blockIdx.x;
    const float* xp = x + (int64_t)plane * H * W;
    const float* dp = dedge + (int64_t)plane * H * W;
    float* op = dx + (int64_t)plane * H * W;
    const float mx = stats[plane * 4], rx = stats[plane * 4 + 1], my = stats[plane * 4 + 2], ry = stats[plane * 4 + 3];
    const float m1x = bstats[plane * 4], m2x = bstats[plane * 4 + 1], m1y = bstats[plane * 4 + 2], m2y = bstats[plane * 4 + 3];
    // Kx[i][j], Ky[i][j] of the forward correlation s(q) = sum K[i][j] x(q + (i-1, j-1)); x(r) is read by q = r - (i-1, j-1)
    const float kx[3][3] = {{-1.f, 0.f, 1.f}, {-2.f, 0.f, 2.f}, {-1.f, 0.f, 1.f}};
    const float ky[3][3] = {{1.f, 2.f, 1.f}, {0.f, 0.f, 0.f}, {-1.f, -2.f, -1.f}};
    for (int r = s * kEdgeThreads + threadIdx.x; r < H * W; r += splits * kEdgeThreads) {
        const int h = r / W, w = r % W;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int qh = h - (i - 1), qw = w - (j - 1);
                if ((unsigned)qh >= (unsigned)H || (unsigned)qw >= (unsigned)W) continue;
                float sx, sy;
                sobel_at(xp, H, W, qh, qw, sx, sy);
                const float ux = (sx - mx) * rx, uy = (sy - my) * ry, g = dp[(int64_t)qh * W + qw];
                const float dsx = rx * (sign_of(ux) * g - m1x - ux * m2x);
                const float dsy = ry * (sign_of(uy) * g - m1y - uy * m2y);
                acc += kx[i][j] * dsx + ky[i][j] * dsy;
            }
        op[r] = acc;
    }
}

static int edge_splits(int64_t planes, int64_t HW) {
    // enough workgroups to cover the chip a few times over, at least one trip of 256 pixels per block
    int64_t s = (2048 + planes - 1) / planes;
    const int64_t most = (HW + kEdgeThreads - 1) / kEdgeThreads;
    if (s > most) s = most;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

}  // namespace segk

using namespace segk;

extern "C" size_t segk_edge_workspace(int64_t planes, int64_t H, int64_t W) {
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    return align_up((size_t)planes * edge_splits(planes, H * W) * 4 * sizeof(float), 256);
}

extern "C" size_t segk_edge_bwd_workspace(int64_t planes, int64_t H, int64_t W) {
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    return segk_edge_workspace(planes, H, W) + align_up((size_t)planes * 4 * sizeof(float), 256);
}

static int edge_check(const char* who, const void* x, int64_t planes, int64_t H, int64_t W, const void* workspace, size_t workspace_bytes) {
    SEGK_REQUIRE(x && workspace, SEGK_EINVAL, "%s: null pointer", who);
    SEGK_REQUIRE(planes > 0 && planes < 65536 && H > 0 && W > 0 && H * W < (int64_t)1 << 30, SEGK_ESHAPE, "%s: bad shape %lld planes of %lldx%lld",
                 who, (long long)planes, (long long)H, (long long)W);
    SEGK_REQUIRE(workspace_bytes >= segk_edge_workspace(planes, H, W), SEGK_EWORKSPACE, "%s: workspace %zu too small", who, workspace_bytes);
    return SEGK_OK;
}

extern "C" int segk_edge_fwd(const float* x, int64_t planes, int64_t H, int64_t W, float eps, float* edge, float* stats,
                             void* workspace, size_t workspace_bytes, int dev, void* stream) {
    if (int rc = edge_check("edge_fwd", x, planes, H, W, workspace, workspace_bytes)) return rc;
    SEGK_REQUIRE(edge && stats, SEGK_EINVAL, "edge_fwd: null pointer");
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int splits = edge_splits(planes, H * W);
    const dim3 grid(splits, (unsigned)planes);
    hipLaunchKernelGGL(edge_stats_partial_kernel, grid, dim3(kEdgeThreads), 0, st, x, (int)H, (int)W, splits, (float*)workspace);
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(edge_stats_finalize_kernel, dim3((unsigned)((planes + kEdgeThreads - 1) / kEdgeThreads)), dim3(kEdgeThreads), 0, st,
                       (const float*)workspace, (int)planes, splits, (int)(H * W), eps, stats);
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(edge_apply_kernel, grid, dim3(kEdgeThreads), 0, st, x, (int)H, (int)W, splits, (const float*)stats, edge);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_edge_bwd(const float* x, const float* dedge, const float* stats, int64_t planes, int64_t H, int64_t W, float* dx,
                             void* workspace, size_t workspace_bytes, int dev, void* stream) {
    if (int rc = edge_check("edge_bwd", x, planes, H, W, workspace, workspace_bytes)) return rc;
    SEGK_REQUIRE(dedge && stats && dx, SEGK_EINVAL, "edge_bwd: null pointer");
    SEGK_REQUIRE(workspace_bytes >= segk_edge_bwd_workspace(planes, H, W), SEGK_EWORKSPACE, "edge_bwd: workspace %zu too small", workspace_bytes);
    SEGK_USE_DEVICE(dev);
    hipStream_t st = (hipStream_t)stream;
    const int splits = edge_splits(planes, H * W);
    const dim3 grid(splits, (unsigned)planes);
    float* ws = (float*)workspace;
    float* bstats = (float*)((char*)workspace + segk_edge_workspace(planes, H, W));
    hipLaunchKernelGGL(edge_bwd_partial_kernel, grid, dim3(kEdgeThreads), 0, st, x, dedge, (int)H, (int)W, splits, stats, ws);
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(edge_bwd_finalize_kernel, dim3((unsigned)((planes + kEdgeThreads - 1) / kEdgeThreads)), dim3(kEdgeThreads), 0, st,
                       (const float*)ws, (int)planes, splits, (int)(H * W), bstats);
    SEGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(edge_bwd_kernel, grid, dim3(kEdgeThreads), 0, st, x, dedge, (int)H, (int)W, splits, stats, (const float*)bstats, dx);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}
