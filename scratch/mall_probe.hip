// Does a consumer kernel that starts where its producer ENDED find that data in the memory-side cache?
// Producer: streams a buffer front to back (read src, write dst).  Consumer: reads dst (and a second operand) either front to
// back or back to front and writes a third buffer.  Prints the consumer's time for both directions at several sizes.
//   hipcc -O3 --offload-arch=gfx950 scratch/mall_probe.hip -o scratch/mall_probe && scratch/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// every block owns a contiguous chunk of the buffer; within it, 16 B per lane per trip
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out,
                                                     long n, int reverse, int chunked) {
    const long nb = gridDim.x;
    if (chunked) {
        const long per = (n + nb - 1) / nb;
        const long lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
        for (long i = lo + threadIdx.x; i < hi; i += 256) {
            const long j = reverse ? (n - 1 - i) : i;
            uint4 v = a[j];
            if (b) { const uint4 w = b[j]; v.x ^= w.x; v.y += w.y; v.z ^= w.z; v.w += w.w; }
            out[j] = v;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += nb * 256) {
            const long j = reverse ? (n - 1 - i) : i;
            uint4 v = a[j];
            if (b) { const uint4 w = b[j]; v.x ^= w.x; v.y += w.y; v.z ^= w.z; v.w += w.w; }
            out[j] = v;
        }
    }
}

int main() {
    const long sizes_mb[] = {75, 151, 302, 604, 1208};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long maxn = 1208L * 1000 * 1000 / 16;
    uint4 *src, *mid, *aux, *dst;
    CK(hipMalloc(&src, maxn * 16)); CK(hipMalloc(&mid, maxn * 16)); CK(hipMalloc(&aux, maxn * 16)); CK(hipMalloc(&dst, maxn * 16));
    CK(hipMemset(src, 1, maxn * 16)); CK(hipMemset(aux, 2, maxn * 16));
    for (int chunked = 0; chunked < 2; ++chunked)
        for (long mb : sizes_mb) {
            const long n = mb * 1000 * 1000 / 16;
            for (int two = 0; two < 2; ++two)
                for (int rev = 0; rev < 2; ++rev) {
                    float best = 1e9f, sum = 0.f;
                    const int reps = 8;
                    for (int r = 0; r < reps; ++r) {
                        // producer always runs front to back (grid-stride: the whole buffer is swept in address order)
                        hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, src, (const uint4*)nullptr, mid, n, 0, 0);
                        CK(hipEventRecord(e0));
                        hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, mid, two ? aux : (const uint4*)nullptr, dst, n, rev, chunked);
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (r >= 2) { sum += ms; if (ms < best) best = ms; }
                    }
                    const double bytes = (double)n * 16 * (two ? 3 : 2);
                    printf("%s consumer, %4ld MB, %d input(s), %s: avg %7.1f us  best %7.1f us  %5.2f TB/s\n", chunked ? "chunked" : "strided",
                           mb, two + 1, rev ? "back-to-front" : "front-to-back", sum / (reps - 2) * 1e3, best * 1e3, bytes / (sum / (reps - 2) * 1e-3) / 1e12);
                }
        }
    return 0;
}
