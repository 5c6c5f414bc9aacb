// Shared device/host helpers of the segk library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/segk.h"

namespace segk {

typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- error reporting (host)
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define SEGK_HIP(call)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return segk::fail(SEGK_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define SEGK_REQUIRE(cond, code, ...)                     \
    do {                                                  \
        if (!(cond)) return segk::fail(code, __VA_ARGS__); \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t esize(int dtype) { return dtype == SEGK_BF16 ? 2 : 4; }
static inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Make `dev` the calling thread's HIP device for the duration of an entry point (autograd's backward runs on another
// host thread than forward, and a tensor may live on a device other than the thread's current one); the previous
// device is restored on return, so the library leaves the caller's (and PyTorch's) notion of "current device" alone.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define SEGK_USE_DEVICE(dev)             \
    segk::DeviceGuard device_guard_(dev); \
    SEGK_HIP(device_guard_.err)

// Error check after a launch without synchronising.
#define SEGK_LAUNCH_CHECK() SEGK_HIP(hipGetLastError())

// ---------------------------------------------------------------- device helpers
struct alignas(16) Vec8f { float v[8]; };

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kBytes = 4;
    // 8 consecutive elements <-> registers
    static __device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    // streaming variants: non-temporal accesses for tensors that are swept once per kernel
    static __device__ __forceinline__ void load8s(const float* p, float (&v)[8]) {
        const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    static __device__ __forceinline__ void store8s(float* p, const float (&v)[8]) {
        __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(p));
        __builtin_nontemporal_store(f32x4{v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4*>(p + 4));
    }
    static __device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
    static __device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    static __device__ __forceinline__ float to_f(float x) { return x; }
    static __device__ __forceinline__ float from_f(float x) { return x; }
};
template <> struct Elem<bf16> {
    static constexpr int kBytes = 2;
    static __device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
    static __device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
        bf16x8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
        *reinterpret_cast<bf16x8*>(p) = a;
    }
    static __device__ __forceinline__ void load8s(const bf16* p, float (&v)[8]) {
        const bf16x8 a = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
    static __device__ __forceinline__ void store8s(bf16* p, const float (&v)[8]) {
        bf16x8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
        __builtin_nontemporal_store(a, reinterpret_cast<bf16x8*>(p));
    }
    static __device__ __forceinline__ void load4(const bf16* p, float (&v)[4]) {
        const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
    }
    static __device__ __forceinline__ void store4(bf16* p, const float (&v)[4]) {
        bf16x4 a;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = (bf16)v[i];
        *reinterpret_cast<bf16x4*>(p) = a;
    }
    static __device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
    static __device__ __forceinline__ bf16 from_f(float x) { return (bf16)x; }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Dispatch a templated launcher on the activation dtype.
#define SEGK_DISPATCH_DTYPE(dtype, T, ...)                                   \
    do {                                                                     \
        if ((dtype) == SEGK_F32) { typedef float T; __VA_ARGS__; }           \
        else if ((dtype) == SEGK_BF16) { typedef segk::bf16 T; __VA_ARGS__; } \
        else return segk::fail(SEGK_EINVAL, "unknown dtype %d", (int)(dtype)); \
    } while (0)

}  // namespace segk
