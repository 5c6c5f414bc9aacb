// What the matrix pipes of this chip sustain when nothing else is in the way: every wave keeps its operands in registers and
// issues v_mfma_f32_16x16x32_bf16 back to back (16 independent accumulators, no LDS, no memory, no barrier), on random bf16
// data, for long enough that the clock settles under load.  Prints the sustained FLOP rate and the shader clock measured
// inside the kernel (s_memtime ticks per 100 MHz s_memrealtime tick).  Build: hipcc -O3 --offload-arch=gfx950 scratch/mfma_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// The same 64 x 64 wave tile (same operand registers, same FLOP per iteration) multiplied with v_mfma_f32_32x32x16_bf16:
// 2 x 2 accumulator tiles of 32 x 32, two k-steps of 16 -- eight instructions where the 16x16x32 form issues sixteen.
__global__ __launch_bounds__(256) void mfma_loop_32(const uint4* __restrict__ data, float* __restrict__ out, int iters,
                                                    unsigned long long* __restrict__ clocks) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 ra[4], rb[4];
    for (int i = 0; i < 4; ++i) { ra[i] = data[(tid * 8 + i) & 0xffff]; rb[i] = data[(tid * 8 + 4 + i) & 0xffff]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[i * 2 + k]), __builtin_bit_cast(bf16x8, rb[j * 2 + k]),
                                                                             acc[i * 2 + j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[tid] = s;
    if (threadIdx.x == 0 && blockIdx.x < 256) { clocks[blockIdx.x * 2] = t1 - t0; clocks[blockIdx.x * 2 + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ data, float* __restrict__ out, int iters,
                                                 unsigned long long* __restrict__ clocks) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 ra[4], rb[4];
    for (int i = 0; i < 4; ++i) { ra[i] = data[(tid * 8 + i) & 0xffff]; rb[i] = data[(tid * 8 + 4 + i) & 0xffff]; }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ra[i]), __builtin_bit_cast(bf16x8, rb[j]),
                                                                         acc[i * 4 + j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[tid] = s;
    if (threadIdx.x == 0 && blockIdx.x < 256) { clocks[blockIdx.x * 2] = t1 - t0; clocks[blockIdx.x * 2 + 1] = r1 - r0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;       // 1, 2 or 4 waves per SIMD
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    const bool big = argc > 3 && atoi(argv[3]) == 32;              // 32: the 32x32x16 instruction
    const int blocks = 256 * waves_per_simd;                       // 256 CUs x 4 SIMDs; a block is 4 waves
    std::vector<unsigned short> h(65536 * 8);
    srand(1);
    for (auto& v : h) {                                            // random bf16 in [-2, 2): sign, exponent 126..128, random mantissa
        const unsigned e = 126 + rand() % 3;
        v = (unsigned short)(((rand() & 1) << 15) | (e << 7) | (rand() & 127));
    }
    uint4* d; float* out; unsigned long long* clk;
    CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&clk, 256 * 2 * 8));
    CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        if (big) hipLaunchKernelGGL(mfma_loop_32, dim3(blocks), dim3(256), 0, 0, d, out, iters, clk);
        else hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, d, out, iters, clk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long hc[512]; CK(hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost));
        double ghz = 0; for (int b = 0; b < 256; ++b) ghz += (double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1;
        const double flop = (double)blocks * 4 * iters * 16 * 16384.0;
        printf("%s waves/SIMD %d  launch %d: %.2f ms  %.0f TFLOP/s  shader clock %.2f GHz  -> %.0f %% of 1024 FLOP/clk/SIMD at that clock\n",
               big ? "32x32x16" : "16x16x32", waves_per_simd, rep, ms, flop / ms / 1e9, ghz / 256, 100.0 * flop / (ms * 1e-3) / (1024.0 * 1024 * ghz / 256 * 1e9));
    }
    return 0;
}
