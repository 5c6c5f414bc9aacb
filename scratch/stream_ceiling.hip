// What rate can a streaming kernel reach on this part, by traffic mix?  Read-only, write-only, copy (1 read : 1 write) and
// 2 reads : 1 write, with 1 or 4 independent 16-byte loads in flight per lane, plain or non-temporal accesses, several grid
// sizes (grid-stride persistent blocks) and buffer sizes from inside the memory-side cache (75 MB) to far beyond it (1.2 GB).
//   hipcc -O3 --offload-arch=gfx950 scratch/stream_ceiling.hip -o scratch/stream_ceiling && scratch/stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// MODE 0: read only (result folded into one conditional store), 1: write only, 2: copy, 3: two inputs -> one output
template <int MODE, int U, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256 * U;
    u32x4 fold = {0u, 0u, 0u, 0u};
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
        u32x4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + (long)u * 256;
            if (i < n) {
                if constexpr (MODE != 1) v[u] = ld<NT>(a + i);
                if constexpr (MODE == 3) w[u] = ld<NT>(b + i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + (long)u * 256;
            if (i < n) {
                if constexpr (MODE == 0) fold ^= v[u];
                if constexpr (MODE == 1) st<NT>(out + i, u32x4{(unsigned)i, 1u, 2u, 3u});
                if constexpr (MODE == 2) st<NT>(out + i, v[u]);
                if constexpr (MODE == 3) st<NT>(out + i, v[u] ^ w[u]);
            }
        }
    }
    if constexpr (MODE == 0) {
        if (fold.x == 0x12345678u && fold.y == 0x9abcdef0u) out[threadIdx.x] = fold;   // never true on this data: keeps the loads alive
    }
}

template <int MODE, int U, bool NT>
static void run(const char* name, const u32x4* a, const u32x4* b, u32x4* out, long n, int grid, long mb, hipEvent_t e0, hipEvent_t e1) {
    float sum = 0.f, best = 1e9f;
    const int reps = 7;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_kernel<MODE, U, NT>), dim3(grid), dim3(256), 0, 0, a, b, out, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    const double streams = MODE == 0 ? 1 : MODE == 1 ? 1 : MODE == 2 ? 2 : 3;
    const double bytes = (double)n * 16 * streams;
    printf("%-10s %4ld MB/buffer  U=%d nt=%d grid=%5d : avg %7.1f us  best %7.1f us  %5.2f TB/s (best %5.2f)\n", name, mb, U, (int)NT, grid,
           sum / (reps - 2) * 1e3, best * 1e3, bytes / (sum / (reps - 2) * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
    fflush(stdout);
}

template <int U, bool NT>
static void all_modes(const u32x4* a, const u32x4* b, u32x4* out, long n, int grid, long mb, hipEvent_t e0, hipEvent_t e1) {
    run<0, U, NT>("read", a, b, out, n, grid, mb, e0, e1);
    run<1, U, NT>("write", a, b, out, n, grid, mb, e0, e1);
    run<2, U, NT>("copy", a, b, out, n, grid, mb, e0, e1);
    run<3, U, NT>("2in-1out", a, b, out, n, grid, mb, e0, e1);
}

int main() {
    const long sizes_mb[] = {75, 302, 604, 1208};
    const int grids[] = {1024, 2048, 4096, 16384};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long maxn = 1208L * 1000 * 1000 / 16;
    u32x4 *a, *b, *out;
    CK(hipMalloc(&a, maxn * 16)); CK(hipMalloc(&b, maxn * 16)); CK(hipMalloc(&out, maxn * 16));
    CK(hipMemset(a, 1, maxn * 16)); CK(hipMemset(b, 2, maxn * 16)); CK(hipMemset(out, 0, maxn * 16));
    for (long mb : sizes_mb) {
        const long n = mb * 1000 * 1000 / 16;
        for (int grid : grids) {
            all_modes<1, false>(a, b, out, n, grid, mb, e0, e1);
            all_modes<4, false>(a, b, out, n, grid, mb, e0, e1);
            all_modes<1, true>(a, b, out, n, grid, mb, e0, e1);
            all_modes<4, true>(a, b, out, n, grid, mb, e0, e1);
        }
    }
    return 0;
}
