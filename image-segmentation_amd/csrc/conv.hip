// Implicit-GEMM 3x3 / 1x1 convolutions on the gfx950 matrix cores (NHWC activations).
//
//   forward / data-gradient :  y[p][co] = sum_{tap,ci} x[p + off(tap)][ci] * w[co][tap][ci]
//   weight-gradient         :  dw[co][tap][ci] = sum_p dy[p][co] * x[p + off(tap)][ci]
//
// Geometry shared by both kernels: a workgroup (4 wavefronts) owns a spatial tile of TH x 16 pixels
// of one image.  Per 128-byte "K-row" of input channels (64 bf16 / 32 fp32) the tile's input halo
// ((TH+2) x 18 pixels) is staged ONCE in LDS and re-used by all nine taps through shifted reads,
// so the activation traffic into LDS is ~1.4x the tile instead of 9x.  LDS rows are 128 B; the
// 16-byte slots of a row are XOR-swizzled with the row index so that the 16 pixels (or 16 output
// channels) a ds_read_b128 group touches fall on distinct banks.  Packed weights are stored in
// global memory already in that swizzled image, so staging them is a straight 16-B copy.
//
// MFMA orientation (forward): A = weights (rows = output channels), B = pixels (columns), so a lane's
// four accumulator registers are four CONSECUTIVE output channels of one pixel -> one 8/16-byte store.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace segk {

constexpr int kConvThreads = 256;
constexpr int kRowBytes = 128;   // one K-row: 64 bf16 or 32 fp32 channels of one pixel / output channel
constexpr int kTW = 16;          // tile width in pixels = MFMA N

template <typename T> struct KRow { static constexpr int kElems = kRowBytes / (int)sizeof(T); };

typedef uint4 Slot;   // 16 bytes

// Packed weight images keep their rows (output channels of the forward image, input channels of the data-gradient
// image) in MFMA order: row r of the 16-row tile t of a 32-row group holds channel 8*(r/4) + 4*(t%2) + r%4 of that
// group.  A lane of the 16x16 MFMA output owns rows 4g..4g+3 of every tile, so across the two tiles of a group it
// owns the EIGHT consecutive channels 8g..8g+7 and the epilogues move 16 bytes (bf16) per lane and pixel.
__host__ __device__ inline int row_channel(int row) { return (row & ~31) + 8 * ((row & 15) >> 2) + 4 * ((row >> 4) & 1) + (row & 3); }

// A strided window onto an NHWC image: logical pixel (y, x) of the convolution grid is physical pixel
// (y*sy + oy, x*sx + ox) of an image with PH x PW pixels.  Identity for ordinary convolutions; the four
// phases of a stride-2 transposed convolution read/write every other row and column.
struct View {
    int PH, PW, sy, sx, oy, ox;
    __device__ __forceinline__ int pixel(int n, int y, int x) const { return (n * PH + y * sy + oy) * PW + x * sx + ox; }
};
static inline View identity_view(int H, int W) { return View{H, W, 1, 1, 0, 0}; }

struct ConvArgs {
    View vin, vout;                   // windows of the input segments / of the output (and mask)
    int tky[9], tkx[9];               // tap offsets + 1 (used when TAPS == 4; 9 and 1 are compile-time)
    const char* x0; const char* x1;   // input segments (bytes)
    int64_t ld0, ld1;                 // pixel strides in elements
    int C0, C1;                       // channels of each segment (a last partial K-row is zero-filled)
    int nch0, nch;                    // K-rows in segment 0 / in total
    int N, H, W, tiles_x, tiles_y;
    const char* w;                    // packed weights [nch][TAPS][cout_pad][128 B]
    int Cout, cout_pad;
    const float* bias;
    char* y; int64_t ldy;
    char* y1 = nullptr; int64_t ldy1 = 0;   // second destination: output channels >= split go to y1 (channel - split)
    int split = 1 << 30, accumulate1 = 0;
    int xcd_remap = 1;
    const char* mask; int64_t ld_mask; int mask_c;   // output channels below mask_c are zeroed where mask <= 0
    int relu, accumulate;
    // volumes (depth > 0): image n is slice n % depth of its volume, and the K dimension runs over nkd (odd, <= 7) depth
    // taps: K-row chunk = kdi * nch0 + c reads channels c of slice n + kdi - nkd/2 (zero beyond the volume's faces)
    int depth = 0, nkd = 3;
    int64_t slice_bytes = 0;          // bytes between consecutive slices of the input
    float* stat_part = nullptr;       // optional [N*tiles][pixel-wave][2][Cout]: per-wave sum / sum of squares of the stored output
    int stat_rpt = 0;                 // rows per tile of stat_part when the caller sized it for four (0: one per pixel-wave)
    // Data gradient whose output is the gradient of relu(norm(z)): the epilogue also produces the two sums the norm's
    // backward needs (sum g, sum g*zhat with g masked by the ReLU), rows laid out as stat_part's.
    const char* nb_z = nullptr; int64_t nb_ldz = 0;   // z: what the norm read, same pixels and channels as the output
    const float* nb_stats = nullptr;                  // [Cout][2] mean, 1/std
    const float* nb_gamma = nullptr; const float* nb_beta = nullptr;
    float* nb_part = nullptr;
};

__device__ __forceinline__ Slot zero_slot() { return make_uint4(0u, 0u, 0u, 0u); }

// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), left in every lane of the row: four rotate-and-add steps
// on the vector ALU (row_ror 8, 4, 2, 1) -- no LDS traffic, unlike a cross-lane permute.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    // two k-substeps of 32 channels per K-row; slot = ks*4 + (lane>>4)
    static __device__ __forceinline__ void run(const Slot& a, const Slot& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // a 16-byte slot holds 4 consecutive channels; lane group g of k-substep t reads slot t*4+g and
    // issues 4 MFMAs, one per element: both operands use the same (slot, element) -> channel map.
    static __device__ __forceinline__ void run(const Slot& a, const Slot& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), c, 0, 0, 0);
    }
};

#ifdef SEGK_CLOCK_PROBE
// Diagnostic build only (make probe): shader-clock and 100 MHz real-time stamps around the main loop of the first
// workgroups, written to a buffer nothing else reads; the ratio is the clock the chip holds inside the kernel.
__device__ unsigned long long g_clock_probe[1024][4];
#define SEGK_PROBE_ENTRY() const unsigned long long probe_e = __builtin_amdgcn_s_memtime();
#define SEGK_PROBE_BEGIN()                                                                              \
    unsigned long long probe_t0 = 0, probe_r0 = 0;                                                      \
    if (threadIdx.x == 0 && blockIdx.x < 1024) { probe_t0 = __builtin_amdgcn_s_memtime(); probe_r0 = __builtin_amdgcn_s_memrealtime(); \
        g_clock_probe[blockIdx.x][2] = probe_t0 - probe_e; }
#define SEGK_PROBE_EXIT()                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_clock_probe[blockIdx.x][3] = __builtin_amdgcn_s_memtime() - probe_e;
#define SEGK_PROBE_END()                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                        \
        g_clock_probe[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - probe_t0;                         \
        g_clock_probe[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - probe_r0;                     \
    }
#else
#define SEGK_PROBE_ENTRY()
#define SEGK_PROBE_BEGIN()
#define SEGK_PROBE_END()
#define SEGK_PROBE_EXIT()
#endif

// ------------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ------------------------------------------------------------------------------------------------
// 16 zero bytes that padding lanes of an LDS-DMA load read instead of branching around the load
__device__ const uint4 kZeroSlot = {0u, 0u, 0u, 0u};

// One LDS-DMA transfer: every active lane moves 16 bytes from its own global address to
// lds_wave_base + 16*lane (the destination is wave-uniform by construction of the hardware).
__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same transfer issued from inline assembly, invisible to hipcc's memory-dependence tracking: the compiler
// then neither waits for it in front of later LDS reads (which target OTHER buffers) nor counts it in vmcnt, so
// the issuing kernel must execute dma_wait_all() before the barrier that publishes the destination buffer.
// lds_addr is the wave-uniform LDS byte address (an SGPR); M0 is saved and restored around the instruction.
__device__ __forceinline__ void glds16_hidden(const char* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's youngest vector-memory operations are still in flight
template <int N> __device__ __forceinline__ void dma_wait_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// WGS: workgroups per CU the kernel is built for.  2 (default): two halo buffers, the next K-row's halo streams in behind
// the MFMAs.  3 or 4 (thin layers, one or two K-rows: memory-latency bound, few steps per tile): ONE halo buffer and no
// halo prefetch, 33-49 KB of LDS and <= 128/170 registers, so that three or four workgroups share a CU and their
// load phases overlap each other's MFMAs.
//
// PIPE (every K-row full, i.e. two k-substeps each; two halo buffers): the loop is skewed by half a step.  The barrier
// that publishes the weights of step s+1 sits between the two k-substeps' MFMA groups: behind it come the MFMAs of
// (s, substep 1), whose fragments are already in registers, so the LDS latency of the first reads from the freshly
// published weight buffer hides under 16 MFMAs instead of being paid after every barrier; the fragments of substep 1
// are read while substep 0 is multiplied.  All loop conditions except "first step of the tile" are compile-time
// (the last K-row is peeled), so a K-row is one straight-line block of 9 taps.
template <typename T, int TAPS, int TH, int BN, int WPX_WAVES, int WCO_WAVES, int WGS = 2, bool PIPE = false>
__global__ __launch_bounds__(64 * WPX_WAVES * WCO_WAVES, WGS * 256 / (64 * WPX_WAVES * WCO_WAVES)) void conv_fwd_kernel(const ConvArgs a) {
    SEGK_PROBE_ENTRY();
    constexpr bool ONE_HALO = WGS > 2;
    static_assert(!PIPE || (!ONE_HALO && sizeof(T) == 2), "the skewed loop is built for bf16 with two halo buffers");
    constexpr int NT = 64 * WPX_WAVES * WCO_WAVES;   // threads: 4 waves (two workgroups per CU) or 8 waves (one)
    constexpr int PAD = TAPS == 1 ? 0 : 1;
    constexpr int HH = TH + 2 * PAD, HW_ = kTW + 2 * PAD, HPIX = HH * HW_;
    constexpr int HSLOTS = HPIX * 8;
    constexpr int HL = (HSLOTS + TAPS * NT - 1) / (TAPS * NT);   // halo transfers / thread / step
    constexpr int WL = BN * 8 / NT;                                        // weight transfers / thread / step
    static_assert(BN * 8 % NT == 0, "weight tile must split evenly");
    constexpr int WCO = BN / WCO_WAVES, WPX = TH * kTW / WPX_WAVES;
    constexpr int NI = WCO / 16, NJ = WPX / 16;
    static_assert((WPX_WAVES * WCO_WAVES == 4 || WPX_WAVES * WCO_WAVES == 8) && NI >= 1 && NJ >= 1, "bad wave layout");
    constexpr int EPS = 16 / (int)sizeof(T);   // elements per 16-byte slot
    // the halo of a K-row is moved in HPARTS parts of HL transfers per thread, one part per tap; the buffer is padded
    // to whole parts so that every wave issues every transfer (the counted DMA wait relies on equal counts)
    constexpr int HPARTS = (HSLOTS + HL * NT - 1) / (HL * NT);
    static_assert(HPARTS <= TAPS, "halo parts must fit the taps of a K-row");
    constexpr int HBYTES = HPARTS * HL * NT * 16, WBYTES = BN * kRowBytes;

    // one LDS object: [halo 0 | halo 1 | weights 0 | weights 1]
    constexpr int NHALO = ONE_HALO ? 1 : 2;
    __shared__ __attribute__((aligned(16))) char smem[NHALO * HBYTES + 2 * WBYTES];
    char* const s_halo0 = smem;
    char* const s_w0 = smem + NHALO * HBYTES;
    const unsigned lds0 = lds_address(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wpx = wave % WPX_WAVES, wco = wave / WPX_WAVES;

    // block -> (output-channel block, image, tile): channel block slowest so that concurrently
    // resident workgroups stream the same weight slab through L2.
    const int nsp = a.N * a.tiles_y * a.tiles_x;
    // Workgroups are handed to the 8 XCDs round-robin (block b runs on XCD b % 8) and every XCD has its own L2.
    // Renumber so that each XCD works through ONE contiguous eighth of the (channel block, tile) list: neighbouring
    // tiles, whose halos overlap, then share an L2 instead of each fetching the overlap from beyond it.
    int vb = blockIdx.x;
    if (a.xcd_remap) {
        const int G = gridDim.x, base = G >> 3, rem = G & 7, x = vb & 7;
        vb = x * base + (x < rem ? x : rem) + (vb >> 3);
    }
    const int cb = vb / nsp;
    int sp = vb % nsp;
    const int tx = sp % a.tiles_x; sp /= a.tiles_x;
    const int ty = sp % a.tiles_y;
    const int n = sp / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * kTW;
    // depth taps of a volume convolution: bit kdi set = slice n + kdi - 1 lies inside the volume
    int slice_mask = 0;
    if (a.depth > 0) {
        const int d = n % a.depth - a.nkd / 2;
        for (int k = 0; k < a.nkd; ++k) slice_mask |= ((unsigned)(d + k) < (unsigned)a.depth ? 1 : 0) << k;
    }

    // ---- staging: global -> LDS by DMA (no registers, stays in flight behind the MFMAs) -------------
    // LDS slot q of the halo image holds pixel q>>3, and -- because the DMA destination is linear in the
    // lane index -- the swizzle is applied to the SOURCE: slot q receives channel slot (q&7) ^ (pixel&7).
    // Padding (outside the image, past the segment's channels, the pad slots behind the halo) reads the 16 zero
    // bytes of kZeroSlot.  The slot -> pixel arithmetic does not depend on the K-row, so every thread keeps, for
    // each of its HPARTS*HL transfers per K-row, (pixel index << 3 | channel slot) or -1 in a register.
    int hinfo[HPARTS * HL];
#pragma unroll
    for (int p = 0; p < HPARTS * HL; ++p) {
        const int q = p * NT + tid;
        const int pix = q >> 3, sl = (q & 7) ^ (pix & 7);
        const int hy = pix / HW_, hx = pix - hy * HW_;
        const int gy = y0 + hy - PAD, gx = x0 + hx - PAD;
        const bool in = q < HSLOTS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        hinfo[p] = in ? ((a.vin.pixel(n, gy, gx) << 3) | sl) : -1;
    }
    // transfers of halo part `part` (HL per thread) of K-row `chunk` into halo buffer `buf`
    auto stage_halo_part = [&](int chunk, int buf, int part) {
        bool s0 = chunk < a.nch0, live = true;
        const char* base = s0 ? a.x0 : a.x1;
        int c = s0 ? chunk : chunk - a.nch0;
        if (a.depth > 0) {                      // block-uniform: depth tap kdi of a volume convolution
            const int kdi = chunk / a.nch0;
            c = chunk - kdi * a.nch0;
            s0 = true;
            base = a.x0 + (int64_t)(kdi - a.nkd / 2) * a.slice_bytes;
            live = (slice_mask >> kdi) & 1;
        }
        const int64_t ld = s0 ? a.ld0 : a.ld1;
        const int ch0 = c * KRow<T>::kElems;
        const int cmax = s0 ? a.C0 : a.C1;
#pragma unroll
        for (int k = 0; k < HL; ++k) {
            const int info = hinfo[part * HL + k];
            const int ch = ch0 + (info & 7) * EPS;
            const bool ok = info >= 0 && ch < cmax && live;
            const char* src = ok ? base + ((int64_t)(info >> 3) * ld + ch) * (int64_t)sizeof(T) : reinterpret_cast<const char*>(&kZeroSlot);
            glds16_hidden(src, lds0 + buf * HBYTES + ((part * HL + k) * NT + wave * 64) * 16);
        }
    };
    const char* const wsrc0 = a.w + (int64_t)cb * BN * kRowBytes + (int64_t)tid * 16;
    auto stage_w = [&](int step, int buf) {
        const char* src = wsrc0 + (int64_t)step * a.cout_pad * kRowBytes;
#pragma unroll
        for (int k = 0; k < WL; ++k)
            glds16_hidden(src + k * NT * 16, lds0 + NHALO * HBYTES + buf * WBYTES + (k * NT + wave * 64) * 16);
    };

    const int nsteps = a.nch * TAPS;

    // ---- prologue: whole halo of K-row 0 and the weights of step 0 -----------------------------------
#pragma unroll
    for (int part = 0; part < HPARTS; ++part) stage_halo_part(0, 0, part);
    stage_w(0, 0);
    // Vectors the epilogue needs besides the accumulators, one per (channel pair tile, pixel row) of the wave: z for the
    // norm-backward sums, or the ReLU mask of a data gradient.  The skewed-loop kernel has the registers to request them
    // here, where the wait for the first halo covers their latency; the other variants request all of them at the start of
    // the epilogue (one latency instead of one per vector).
    struct alignas(16) Pack8 { T e[8]; };
    constexpr bool kEarlyZ = PIPE;
    Pack8 zr[NI / 2][NJ];
    auto load_side = [&](const char* base, int64_t ld, int cmax) {
#pragma unroll
        for (int k2 = 0; k2 < NI / 2; ++k2) {
            const int co = cb * BN + (wave / WPX_WAVES) * WCO + 32 * k2 + 8 * (lane >> 4);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int yy = y0 + (wave % WPX_WAVES) * NJ + j, xx = x0 + (lane & 15);
                if (co < cmax && yy < a.H && xx < a.W)
                    zr[k2][j] = *reinterpret_cast<const Pack8*>(reinterpret_cast<const T*>(base) + (int64_t)a.vout.pixel(n, yy, xx) * ld + co);
            }
        }
    };
    auto load_z = [&]() { load_side(a.nb_z, a.nb_ldz, a.Cout); };
    auto load_mask = [&]() { load_side(a.mask, a.ld_mask, a.mask_c); };
    if constexpr (kEarlyZ) {
        if (a.nb_part) load_z();
        else if (a.mask) load_mask();
    }
    dma_wait_all();
    __syncthreads();

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS read bases; everything tap-dependent below is a compile-time offset (taps are unrolled)
    const int co_l = wco * WCO + l15, hp_l = wpx * NJ * HW_ + l15;
    // Pixel-fragment reads: halo pixel hp = hp_l + c with c = (row + ky) * HW_ + kx known at compile time, swizzled slot
    // (ks*4 + g) ^ (hp & 7).  Only c & 7 enters the swizzle, so the eight possible byte offsets per k-substep are
    // computed once per lane and every read is base[c & 7] + an immediate -- no address arithmetic inside the tap loop.
    int boff[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 8; ++c) boff[ks][c] = hp_l * kRowBytes + (((ks * 4 + g) ^ ((hp_l + c) & 7)) << 4);

    SEGK_PROBE_BEGIN();
    if constexpr (PIPE) {
        Slot fa0[NI], fb0[NJ], fa1[NI], fb1[NJ];
        auto mma_all = [&](const Slot (&fa)[NI], const Slot (&fb)[NJ]) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        };
        // one K-row: NEXT = another K-row follows (its halo streams in behind taps 0 .. HPARTS-1)
        auto krow = [&](int chunk, auto next_tag) {
            constexpr bool NEXT = decltype(next_tag)::value;
            const char* hb = s_halo0 + (chunk & 1) * HBYTES;
            auto read_b = [&](int tap, int ks, Slot (&fb)[NJ]) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (TAPS != 4) {
                        const int ky = TAPS == 9 ? tap / 3 : 0, kx = TAPS == 9 ? tap % 3 : 0;
                        const int c = (j + ky) * HW_ + kx;
                        fb[j] = *reinterpret_cast<const Slot*>(hb + boff[ks][c & 7] + c * kRowBytes);
                    } else {
                        const int hp = hp_l + (j + a.tky[tap]) * HW_ + a.tkx[tap];
                        fb[j] = *reinterpret_cast<const Slot*>(hb + hp * kRowBytes + (((ks * 4 + g) ^ (hp & 7)) << 4));
                    }
                }
            };
            auto read_a = [&](const char* wb, int ks, Slot (&fa)[NI]) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int co = co_l + i * 16;
                    fa[i] = *reinterpret_cast<const Slot*>(wb + co * kRowBytes + (((ks * 4 + g) ^ (co & 7)) << 4));
                }
            };
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int step = chunk * TAPS + tap;
                const char* wb = s_w0 + (step & 1) * WBYTES;
                constexpr bool kMoreW = true;
                const bool more_w = NEXT || tap + 1 < TAPS;
                const bool halo_now = NEXT && tap < HPARTS;
                // behind the barrier: DMA for the next step (its buffer was last read one step ago), this step's
                // substep-0 fragments from the buffer that barrier published
                if (more_w) stage_w(step + 1, (step + 1) & 1);
                if (halo_now) stage_halo_part(chunk + 1, (chunk + 1) & 1, tap);
                read_a(wb, 0, fa0);
                read_b(tap, 0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                if (tap > 0 || chunk > 0) mma_all(fa1, fb1);          // (previous step, substep 1)
                __builtin_amdgcn_sched_barrier(0);
                read_a(wb, 1, fa1);
                read_b(tap, 1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                mma_all(fa0, fb0);                                     // (this step, substep 0)
                __builtin_amdgcn_sched_barrier(0);
                // every read of this step's weight buffer has returned (the next step's DMA overwrites it), the next
                // step's weights have landed; the halo parts issued in this step may stay in flight, except behind the
                // K-row's last tap, where the next K-row's halo is about to be read
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (tap == TAPS - 1 || !halo_now) dma_wait_all();
                else dma_wait_but<HL>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                (void)kMoreW;
            }
        };
        for (int chunk = 0; chunk + 1 < a.nch; ++chunk) krow(chunk, std::true_type{});
        krow(a.nch - 1, std::false_type{});
        mma_all(fa1, fb1);                                             // (last step, substep 1)
    } else
    for (int chunk = 0; chunk < a.nch; ++chunk) {
        // a K-row whose second half holds no real channels (C = 32 in bf16, the 27-value patch row of the first
        // convolution, the tail of a 96-channel concat) is multiplied with one k-substep instead of two
        const int rc = a.depth > 0 ? a.C0 - (chunk % a.nch0) * KRow<T>::kElems
                                   : (chunk < a.nch0 ? a.C0 - chunk * KRow<T>::kElems : a.C1 - (chunk - a.nch0) * KRow<T>::kElems);
        const int nks = rc > KRow<T>::kElems / 2 ? 2 : 1;
        const bool next_chunk = !ONE_HALO && chunk + 1 < a.nch;
        const char* hb = s_halo0 + (ONE_HALO ? 0 : (chunk & 1)) * HBYTES;
        if (ONE_HALO && chunk > 0) {
            // single halo buffer: everyone has passed the previous K-row's last barrier, so the buffer is free
#pragma unroll
            for (int part = 0; part < HPARTS; ++part) stage_halo_part(chunk, 0, part);
            dma_wait_all();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        auto tap_ky = [&](int tap) { return TAPS == 9 ? tap / 3 : (TAPS == 4 ? a.tky[tap] : 0); };
        auto tap_kx = [&](int tap) { return TAPS == 9 ? tap % 3 : (TAPS == 4 ? a.tkx[tap] : 0); };
        auto load_b = [&](int tap, int ks, Slot (&fb)[NJ]) {
            const int slot = ks * 4 + g;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (TAPS != 4) {
                    const int c = (j + tap_ky(tap)) * HW_ + tap_kx(tap);       // a constant once the tap loop is unrolled
                    fb[j] = *reinterpret_cast<const Slot*>(hb + boff[ks][c & 7] + c * kRowBytes);
                } else {
                    const int hp = hp_l + (j + tap_ky(tap)) * HW_ + tap_kx(tap);
                    fb[j] = *reinterpret_cast<const Slot*>(hb + hp * kRowBytes + ((slot ^ (hp & 7)) << 4));
                }
            }
        };
        auto load_a = [&](const char* wb, int ks, Slot (&fa)[NI]) {
            const int slot = ks * 4 + g;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int co = co_l + i * 16;
                fa[i] = *reinterpret_cast<const Slot*>(wb + co * kRowBytes + ((slot ^ (co & 7)) << 4));
            }
        };
        // Pixel fragments of a tap's first k-substep are read BEFORE the barrier that ends the previous tap (the halo
        // buffer does not change within a K-row), so only the weight fragments wait for that barrier.
        Slot fb0[NJ];
        load_b(0, 0, fb0);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int step = chunk * TAPS + tap;
            const char* wb = s_w0 + (step & 1) * WBYTES;
            Slot fa0[NI];
            load_a(wb, 0, fa0);
            __builtin_amdgcn_sched_barrier(0);
            // DMA for later steps goes out while those reads are in flight: the weights of the NEXT (K-row, tap) into the
            // other weight buffer, then one part of the NEXT K-row's halo into the other halo buffer.  Issue order
            // matters: the wait at the end of the step leaves the (younger) halo transfers in flight, so their HBM
            // latency is spread over the remaining taps of the K-row instead of being paid every step.
            if (step + 1 < nsteps) stage_w(step + 1, (step + 1) & 1);
            const bool halo_now = tap < HPARTS && next_chunk;
            if (halo_now) stage_halo_part(chunk + 1, (chunk + 1) & 1, tap);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) Mma<T>::run(fa0[i], fb0[j], acc[i][j]);
            if (nks == 2) {
                Slot fa1[NI], fb1[NJ];
                load_a(wb, 1, fa1);
                load_b(tap, 1, fb1);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) Mma<T>::run(fa1[i], fb1[j], acc[i][j]);
            }
            if (tap + 1 < TAPS) load_b(tap + 1, 0, fb0);
            // keep the MFMA cluster and the prefetch in front of the DMA wait.  Wait for everything but this step's halo
            // transfers (the last tap of a K-row waits for all: the next K-row reads that halo), then: next weight
            // buffer complete, current buffers free.  A bare s_barrier: the LDS reads in flight across it are of a buffer
            // that nobody overwrites before the NEXT barrier, and DMA completion is tracked by the counted waits.
            __builtin_amdgcn_sched_barrier(0);
            if (tap == TAPS - 1 || !halo_now) dma_wait_all();
            else dma_wait_but<HL>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    SEGK_PROBE_END();
    // ---- epilogue: bias, ReLU, mask, store 8 consecutive channels per lane (rows are in MFMA order) ----------
    static_assert(NI % 2 == 0, "channel tiles come in pairs");
    if (a.nb_part) {
        // The output is g = dL/d relu(norm(z)).  Store it, and reduce over the wave's pixels what the norm's backward
        // needs: s1 = sum g*[pre > 0], s2 = sum g*[pre > 0]*zhat, with pre and zhat computed from z by the very
        // expressions of the forward and of the dx pass, and g rounded to T as that pass will read it.
        // All of the wave's z vectors are requested before the stores of g, so that their latency is paid once.
        if constexpr (!kEarlyZ) load_z();
#pragma unroll
        for (int k2 = 0; k2 < NI / 2; ++k2) {
            const int co = cb * BN + wco * WCO + 32 * k2 + 8 * g;
            if (co >= a.Cout) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int yy = y0 + wpx * NJ + j, xx = x0 + l15;
                if (yy >= a.H || xx >= a.W) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[2 * k2][j][e]; v[4 + e] = acc[2 * k2 + 1][j][e]; }
                Elem<T>::store8(reinterpret_cast<T*>(a.y) + (int64_t)a.vout.pixel(n, yy, xx) * a.ldy + co, v);
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < NI / 2; ++k2) {
            const int co = cb * BN + wco * WCO + 32 * k2 + 8 * g;
            if (co >= a.Cout) continue;
            float mean[8], rstd[8], sc[8], sh[8], s1[8], s2[8];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 st = *reinterpret_cast<const float4*>(a.nb_stats + 2 * (co + 2 * q4));
                mean[2 * q4] = st.x; rstd[2 * q4] = st.y; mean[2 * q4 + 1] = st.z; rstd[2 * q4 + 1] = st.w;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float ga = a.nb_gamma ? a.nb_gamma[co + r] : 1.f, be = a.nb_beta ? a.nb_beta[co + r] : 0.f;
                sc[r] = rstd[r] * ga;
                sh[r] = be - mean[r] * sc[r];
                s1[r] = s2[r] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int yy = y0 + wpx * NJ + j, xx = x0 + l15;
                if (yy >= a.H || xx >= a.W) continue;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float zz = Elem<T>::to_f(zr[k2][j].e[r]);
                    const float pre = zz * sc[r] + sh[r];
                    const float zh = (zz - mean[r]) * rstd[r];
                    const float gv = r < 4 ? acc[2 * k2][j][r & 3] : acc[2 * k2 + 1][j][r & 3];
                    const float gg = pre > 0.f ? Elem<T>::to_f(Elem<T>::from_f(gv)) : 0.f;
                    s1[r] += gg;
                    s2[r] += gg * zh;
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
            if (l15 == 0) {
                const int tile = (n * a.tiles_y + ty) * a.tiles_x + tx;
                float* dst = a.nb_part + ((int64_t)(tile * WPX_WAVES + wpx) * 2) * a.Cout + co;
                *reinterpret_cast<float4*>(dst) = make_float4(s1[0], s1[1], s1[2], s1[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(s1[4], s1[5], s1[6], s1[7]);
                *reinterpret_cast<float4*>(dst + a.Cout) = make_float4(s2[0], s2[1], s2[2], s2[3]);
                *reinterpret_cast<float4*>(dst + a.Cout + 4) = make_float4(s2[4], s2[5], s2[6], s2[7]);
            }
        }
        SEGK_PROBE_EXIT();
        return;
    }
    if (!a.stat_part) {
        if constexpr (!kEarlyZ) {
            if (a.mask) load_mask();
        }
        // the bias of the wave's channels, requested once (a load between the stores of two rows would wait for memory again)
        float bv[NI / 2][8];
#pragma unroll
        for (int k2 = 0; k2 < NI / 2; ++k2) {
            const int co = cb * BN + wco * WCO + 32 * k2 + 8 * g;
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            if (a.bias && co < a.Cout) { b0 = *reinterpret_cast<const float4*>(a.bias + co); b1 = *reinterpret_cast<const float4*>(a.bias + co + 4); }
            bv[k2][0] = b0.x; bv[k2][1] = b0.y; bv[k2][2] = b0.z; bv[k2][3] = b0.w; bv[k2][4] = b1.x; bv[k2][5] = b1.y; bv[k2][6] = b1.z; bv[k2][7] = b1.w;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int yy = y0 + wpx * NJ + j, xx = x0 + l15;
            if (yy >= a.H || xx >= a.W) continue;
            const int64_t pix = a.vout.pixel(n, yy, xx);
#pragma unroll
            for (int k2 = 0; k2 < NI / 2; ++k2) {
                const int co = cb * BN + wco * WCO + 32 * k2 + 8 * g;
                if (co >= a.Cout) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[2 * k2][j][e]; v[4 + e] = acc[2 * k2 + 1][j][e]; }
                if (a.bias) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += bv[k2][r];
                }
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (a.mask && co < a.mask_c) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = Elem<T>::to_f(zr[k2][j].e[r]) > 0.f ? v[r] : 0.f;
                }
                const bool second = co >= a.split;
                T* dst = second ? reinterpret_cast<T*>(a.y1) + pix * a.ldy1 + (co - a.split) : reinterpret_cast<T*>(a.y) + pix * a.ldy + co;
                if (second ? a.accumulate1 : a.accumulate) {
                    float o[8];
                    Elem<T>::load8(dst, o);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += o[r];
                }
                Elem<T>::store8(dst, v);
            }
        }
        SEGK_PROBE_EXIT();
        return;
    }
    // Epilogue of a convolution in front of a normalisation (single destination, no mask, no accumulation): besides
    // the store, the per-channel sum and sum of squares of what was stored (rounded to T).  16 lanes hold the 16 pixels
    // of a tile row for the same 8 channels, so a 4-step butterfly over them plus the sum over the wave's rows gives one
    // partial per (tile, pixel-wave, channel), written once -- no atomics.
#pragma unroll
    for (int k2 = 0; k2 < NI / 2; ++k2) {
        const int co = cb * BN + wco * WCO + 32 * k2 + 8 * g;
        if (co >= a.Cout) continue;
        float ssum[8], ssq[8], bv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) ssum[r] = ssq[r] = bv[r] = 0.f;
        if (a.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co), b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int yy = y0 + wpx * NJ + j, xx = x0 + l15;
            if (yy >= a.H || xx >= a.W) continue;
            const int64_t pix = a.vout.pixel(n, yy, xx);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[2 * k2][j][e] + bv[e]; v[4 + e] = acc[2 * k2 + 1][j][e] + bv[4 + e]; }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            Elem<T>::store8(reinterpret_cast<T*>(a.y) + pix * a.ldy + co, v);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float q = Elem<T>::to_f(Elem<T>::from_f(v[r]));
                ssum[r] += q;
                ssq[r] += q * q;
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) { ssum[r] = row16_sum(ssum[r]); ssq[r] = row16_sum(ssq[r]); }
        if (l15 == 0) {
            const int tile = (n * a.tiles_y + ty) * a.tiles_x + tx;
            const int rpt = a.stat_rpt > WPX_WAVES ? a.stat_rpt : WPX_WAVES;
            float* dst = a.stat_part + ((int64_t)(tile * rpt + wpx) * 2) * a.Cout + co;
            *reinterpret_cast<float4*>(dst) = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(ssum[4], ssum[5], ssum[6], ssum[7]);
            *reinterpret_cast<float4*>(dst + a.Cout) = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
            *reinterpret_cast<float4*>(dst + a.Cout + 4) = make_float4(ssq[4], ssq[5], ssq[6], ssq[7]);
            // a buffer sized for more rows per tile than this variant has pixel-waves: the rows it does not own are zero
            for (int extra = WPX_WAVES + wpx; extra < rpt; extra += WPX_WAVES) {
                float* z = a.stat_part + ((int64_t)(tile * rpt + extra) * 2) * a.Cout + co;
                const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(z) = zero; *reinterpret_cast<float4*>(z + 4) = zero;
                *reinterpret_cast<float4*>(z + a.Cout) = zero; *reinterpret_cast<float4*>(z + a.Cout + 4) = zero;
            }
        }
    }
    SEGK_PROBE_EXIT();
}

// ------------------------------------------------------------------------------------------------
// forward / data-gradient kernel for THIN layers (bf16, 3x3, few input channels, one output-channel block):
// 64 -> 64 and 96 -> 32 at 384^2, 128 -> 32 at 192^2 and the data gradients 32 -> 96, 32 -> 128.  These layers move
// about as many bytes as they multiply (the machine balance), and the LDS-staged kernel above cannot keep enough bytes in
// flight for them: its in-flight data is what fits in LDS next to the weights (~45 KB per CU, ~3 TB/s at the latency
// under load).  Here the roles are swapped:
//   * the WHOLE packed weight image of the layer (all taps, <= 147 KB) is copied into LDS once per workgroup and stays;
//   * every wavefront is an independent pipeline with no barrier in its loop: it owns a strip 16 pixels wide and walks down
//     it two output rows at a time, loading its pixel fragments STRAIGHT from global memory into registers in the matrix
//     instruction's operand layout (lane (g, p) = 16 bytes: channels 8g..8g+7 of a 32-channel k-substep of pixel p), rows
//     requested PD steps ahead, so the register file (512 KB per CU), not LDS, holds the bytes in flight;
//   * the fragments of the taps kx = 0 and 2 are the centre fragment shifted by one lane within its 16-lane row (one DPP
//     move per register), with the strip's neighbouring pixel column, loaded separately, entering at lane 0 or 15.
// The accumulation order per output element is that of the kernel above (K-row, tap, k-substep), so the results are bit
// for bit the same.
// ------------------------------------------------------------------------------------------------
struct DirectArgs {
    const char* seg0; const char* seg1;   // input segments
    int64_t ldb0, ldb1;                   // their pixel strides in bytes
    int off1;                             // byte offset of K-row 1 inside a pixel of segment 0 (when there is no segment 1)
    int N, H, W, strips_x, seg_rows, segs_y, units;
    const char* w;                        // packed image [K-row][tap][BN][128 B]
    int Cout; const float* bias;
    char* y; int64_t ldy; char* y1; int64_t ldy1; int split, accumulate, accumulate1;
    const char* mask; int64_t ld_mask; int mask_c; int relu;
    float* stat_part; int tiles_x, tiles_y;
    int pair_first;                       // 32-channel output groups from this one on are stored in pairs (whole cache lines)
};

constexpr int kDirectWaves = 8;

// NK0 / NK1: k-substeps (32 channels each) of K-row 0 / K-row 1 (0: a single K-row); NI: 16-channel output tiles (BN = 16 NI);
// PD: how many steps ahead the rows of a step are requested
// STATS: the epilogue also produces the per-channel sums for the normalisation that follows (a.stat_part is set)
// MASK: a.mask is set (a data gradient that applies its producers' ReLU backward): the mask vectors of a step are requested
// before its matrix instructions, like the pixel rows, instead of in the epilogue
// ABUF: 2 = the weight fragments of the next block are read while this block is multiplied; 1 = read right before use (wide
// output blocks, whose sixteen matrix instructions per block cover most of the read by themselves and whose registers are short)
// NCB: output-channel blocks of BN channels; the weights of all of them stay in LDS and a unit of work is (block, strip segment)
template <int NK0, int NK1, int NI, int PD, bool STATS, bool MASK, int ABUF = 2, int NCB = 1>
__global__ __launch_bounds__(64 * kDirectWaves, 2) void conv_direct_kernel(const DirectArgs a) {
    constexpr int NKS = NK0 + NK1, NCH = NK1 ? 2 : 1, BN = NI * 16, RING = 4 + 2 * PD;
    constexpr int WBYTES = NCH * 9 * NCB * BN * kRowBytes;
    static_assert(NI % 2 == 0 && WBYTES <= 150 * 1024, "weight image must fit in LDS");
    __shared__ __attribute__((aligned(16))) char sw[WBYTES];
    __shared__ __attribute__((aligned(16))) float sbias[NCB * BN];   // the bias beside the weights: an epilogue that waited on
                                                                     // global memory for it would stall every step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    for (int q = tid; q < WBYTES / 16; q += 64 * kDirectWaves)
        *reinterpret_cast<Slot*>(sw + q * 16) = *reinterpret_cast<const Slot*>(a.w + (int64_t)q * 16);
    if (a.bias)
        for (int q = tid; q < NCB * BN; q += 64 * kDirectWaves) sbias[q] = q < a.Cout ? a.bias[q] : 0.f;
    __syncthreads();

    // per k-substep: segment base, pixel stride and byte offset of this lane's 16 bytes inside a pixel
    const char* kbase[NKS];
    int64_t kld[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        const bool row1 = s >= NK0;
        const bool from1 = row1 && a.seg1 != nullptr;
        kbase[s] = (from1 ? a.seg1 : a.seg0) + (row1 && !from1 ? a.off1 : 0) + (row1 ? s - NK0 : s) * 64 + g * 16;
        kld[s] = from1 ? a.ldb1 : a.ldb0;
    }
    // A (weight) fragment addresses: row co = i*16 + l15 of the (K-row, tap) image, slot (ks*4 + g) ^ (co & 7)
    int aoff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) aoff[ks] = l15 * kRowBytes + (((ks * 4 + g) ^ (l15 & 7)) << 4);

    auto shift = [](const Slot& edge, const Slot& ctr, auto dir) -> Slot {
        // lane p of every 16-lane row takes the centre fragment of lane p-1 (dir 0: tap kx = 0) or p+1 (dir 1: kx = 2); the lane
        // that has no such neighbour keeps `edge`, the pixel beside the strip
        constexpr int ctrl = decltype(dir)::value == 0 ? 0x111 : 0x101;   // row_shr:1 / row_shl:1
        Slot r;
        r.x = (unsigned)__builtin_amdgcn_update_dpp((int)edge.x, (int)ctr.x, ctrl, 0xf, 0xf, false);
        r.y = (unsigned)__builtin_amdgcn_update_dpp((int)edge.y, (int)ctr.y, ctrl, 0xf, 0xf, false);
        r.z = (unsigned)__builtin_amdgcn_update_dpp((int)edge.z, (int)ctr.z, ctrl, 0xf, 0xf, false);
        r.w = (unsigned)__builtin_amdgcn_update_dpp((int)edge.w, (int)ctr.w, ctrl, 0xf, 0xf, false);
        return r;
    };

    const int total_waves = gridDim.x * kDirectWaves;
    for (int u = blockIdx.x * kDirectWaves + wave; u < a.units; u += total_waves) {
        const int sx = u % a.strips_x;
        const int sy = (u / a.strips_x) % a.segs_y;
        const int ncb_n = u / (a.strips_x * a.segs_y);
        const int n = ncb_n / NCB, cb = ncb_n % NCB;
        const int x0 = sx * kTW, ys = sy * a.seg_rows;
        const int rows_here = min(a.seg_rows, a.H - ys);
        const int nsteps = rows_here >> 1;
        // the strip's neighbouring pixel column: lanes 0-7 of a row carry the left one, lanes 8-15 the right one
        const int ex = l15 < 8 ? x0 - 1 : x0 + kTW;
        const bool ex_ok = ex >= 0 && ex < a.W;

        Slot ctr[RING][NKS], edg[RING][NKS];
        auto load_row = [&](auto slot_tag, int r) {
            constexpr int sl = decltype(slot_tag)::value;
            const int yy = ys - 1 + r;
            if (r >= rows_here + 2) return;                       // not needed by any step of this unit
            if ((unsigned)yy < (unsigned)a.H) {
                const int64_t rowpix = ((int64_t)n * a.H + yy) * a.W;
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    ctr[sl][s] = *reinterpret_cast<const Slot*>(kbase[s] + (rowpix + x0 + l15) * kld[s]);
                    edg[sl][s] = ex_ok ? *reinterpret_cast<const Slot*>(kbase[s] + (rowpix + ex) * kld[s]) : zero_slot();
                }
            } else {
#pragma unroll
                for (int s = 0; s < NKS; ++s) { ctr[sl][s] = zero_slot(); edg[sl][s] = zero_slot(); }
            }
        };
        // rows of the first PD steps
        [&]<int... R>(std::integer_sequence<int, R...>) { (load_row(std::integral_constant<int, R % RING>{}, R), ...); }(
            std::make_integer_sequence<int, 2 * PD + 2>{});

        auto step = [&](auto q_tag, int st) {
            constexpr int q = decltype(q_tag)::value;             // position inside the unrolled group: window slots 2q .. 2q+3
            // request the rows of step st + PD
            load_row(std::integral_constant<int, (2 * q + 2 * PD + 2) % RING>{}, 2 * st + 2 * PD + 2);
            load_row(std::integral_constant<int, (2 * q + 2 * PD + 3) % RING>{}, 2 * st + 2 * PD + 3);
            Slot mk[MASK ? 2 : 1][MASK ? NI / 2 : 1];
            if constexpr (MASK) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int64_t pix = ((int64_t)n * a.H + ys + 2 * st + j) * a.W + x0 + l15;
#pragma unroll
                    for (int k2 = 0; k2 < NI / 2; ++k2) {
                        const int co = cb * BN + 32 * k2 + 8 * g;
                        mk[j][k2] = co < a.mask_c ? *reinterpret_cast<const Slot*>(reinterpret_cast<const bf16*>(a.mask) + pix * a.ld_mask + co)
                                                  : make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // ones: keep
                    }
                }
            }
            f32x4 acc[NI][2];
#pragma unroll
            for (int i = 0; i < NI; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // blocks (K-row, tap, k-substep) in the accumulation order of conv_fwd_kernel; the weight fragments of block b+1 are
            // read from LDS before the matrix instructions of block b are issued
            constexpr int NBLK = 9 * NKS;
            auto read_a = [&](auto b_tag, Slot (&fa)[NI]) {
                constexpr int b = decltype(b_tag)::value;
                constexpr int c = b < 9 * NK0 ? 0 : 1, bb = c == 0 ? b : b - 9 * NK0, nk = c == 0 ? NK0 : NK1;
                constexpr int tap = bb / nk, ks = bb % nk;
                const char* wt = sw + ((c * 9 + tap) * NCB + cb) * BN * kRowBytes;
#pragma unroll
                for (int i = 0; i < NI; ++i) fa[i] = *reinterpret_cast<const Slot*>(wt + i * 16 * kRowBytes + aoff[ks]);
            };
            Slot fa[ABUF][NI];
            if constexpr (ABUF == 2) read_a(std::integral_constant<int, 0>{}, fa[0]);
            [&]<int... B>(std::integer_sequence<int, B...>) {
                ([&] {
                    constexpr int b = B;
                    constexpr int c = b < 9 * NK0 ? 0 : 1, bb = c == 0 ? b : b - 9 * NK0, nk = c == 0 ? NK0 : NK1;
                    constexpr int tap = bb / nk, ks = bb % nk, s = c == 0 ? ks : NK0 + ks;
                    constexpr int ky = tap / 3, kx = tap % 3;
                    if constexpr (ABUF == 1) read_a(std::integral_constant<int, b>{}, fa[0]);
                    else if constexpr (b + 1 < NBLK) read_a(std::integral_constant<int, b + 1>{}, fa[(b + 1) & 1]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int sl = (2 * q + j + ky) % RING;
                        Slot fb;
                        if constexpr (kx == 1) fb = ctr[sl][s];
                        else if constexpr (kx == 0) fb = shift(edg[sl][s], ctr[sl][s], std::integral_constant<int, 0>{});
                        else fb = shift(edg[sl][s], ctr[sl][s], std::integral_constant<int, 1>{});
#pragma unroll
                        for (int i = 0; i < NI; ++i) Mma<bf16>::run(fa[ABUF == 2 ? (b & 1) : 0][i], fb, acc[i][j]);
                    }
                    // keep the scheduler from pulling the weight reads of later blocks (and their registers) up here
                    __builtin_amdgcn_sched_barrier(0);
                }(), ...);
            }(std::make_integer_sequence<int, NBLK>{});
            float ssum[STATS ? NI / 2 : 1][8], ssq[STATS ? NI / 2 : 1][8];   // this step's two rows
            if constexpr (STATS) {
#pragma unroll
                for (int k2 = 0; k2 < NI / 2; ++k2)
#pragma unroll
                    for (int r = 0; r < 8; ++r) ssum[k2][r] = ssq[k2][r] = 0.f;
            }
            // epilogue of the two rows: the conv_fwd_kernel epilogue for one pixel row per j.  The results are first packed to
            // bf16, then stored: a lane holds 16 bytes (8 channels) of a pixel, four lanes (g) one 64-byte half of a 64-channel
            // stretch, so one store instruction per 32-channel group writes HALF cache lines of 16 pixels.  Where two groups
            // are neighbours in the same tensor they are stored as a pair instead: the lanes of pixels 8-15 trade their group-A
            // vector for the group-B vector of pixels 0-7 (one rotate by 8 lanes per register), so that each of the two
            // instructions writes whole 128-byte lines of 8 pixels.
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int yy = ys + 2 * st + j;
                const int64_t pix0 = ((int64_t)n * a.H + yy) * a.W + x0;
                const int64_t pix = pix0 + l15;
                Slot outp[NI / 2];
#pragma unroll
                for (int k2 = 0; k2 < NI / 2; ++k2) {
                    const int co = cb * BN + 32 * k2 + 8 * g;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = acc[2 * k2][j][e]; v[4 + e] = acc[2 * k2 + 1][j][e]; }
                    if (a.bias) {
                        const float4 b0 = *reinterpret_cast<const float4*>(sbias + co), b1 = *reinterpret_cast<const float4*>(sbias + co + 4);
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    }
                    if (a.relu) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if constexpr (MASK) {
                        // a bf16 is positive exactly when its 16 bits, read as a signed integer, are
                        const unsigned mw[4] = {mk[j][k2].x, mk[j][k2].y, mk[j][k2].z, mk[j][k2].w};
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = (short)(mw[r >> 1] >> (16 * (r & 1))) > 0 ? v[r] : 0.f;
                    }
                    const bool second = co >= a.split;
                    if ((second ? a.accumulate1 : a.accumulate) && co < a.Cout) {
                        const bf16* dst = second ? reinterpret_cast<const bf16*>(a.y1) + pix * a.ldy1 + (co - a.split)
                                                 : reinterpret_cast<const bf16*>(a.y) + pix * a.ldy + co;
                        float o[8];
                        Elem<bf16>::load8(dst, o);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += o[r];
                    }
                    bf16x8 pk;
#pragma unroll
                    for (int r = 0; r < 8; ++r) pk[r] = (bf16)v[r];
                    outp[k2] = __builtin_bit_cast(Slot, pk);
                    if constexpr (STATS) {
                        if (co < a.Cout) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                const float qv = (float)pk[r];
                                ssum[k2][r] += qv;
                                ssq[k2][r] += qv * qv;
                            }
                        }
                    }
                }
                auto dst_of = [&](int64_t p, int co) -> char* {
                    return co >= a.split ? a.y1 + (p * a.ldy1 + (co - a.split)) * 2 : a.y + (p * a.ldy + co) * 2;
                };
#pragma unroll
                for (int k2 = 0; k2 < NI / 2; ++k2) {
                    const int cg0 = cb * BN + 32 * k2;                          // first channel of the group
                    if (cg0 >= a.Cout) continue;
                    const int rel = k2 - a.pair_first;
                    if (rel >= 0 && (rel & 1)) continue;                        // second half of a pair: already stored
                    if (rel >= 0 && k2 + 1 < NI / 2 && cg0 + 32 < a.Cout) {
                        const bool hi = l15 >= 8;
                        const Slot A = outp[k2], B = outp[k2 + 1 < NI / 2 ? k2 + 1 : k2];
                        auto ror8 = [](unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false); };
                        const Slot Ar = make_uint4(ror8(A.x), ror8(A.y), ror8(A.z), ror8(A.w));
                        const Slot Br = make_uint4(ror8(B.x), ror8(B.y), ror8(B.z), ror8(B.w));
                        const Slot SA = hi ? Br : A, SB = hi ? B : Ar;
                        const int co = cg0 + 8 * g + (hi ? 32 : 0);
                        if (co < a.Cout) {                                      // Cout need not fill its last 32-channel group
                            *reinterpret_cast<Slot*>(dst_of(pix0 + (l15 & 7), co)) = SA;
                            *reinterpret_cast<Slot*>(dst_of(pix0 + 8 + (l15 & 7), co)) = SB;
                        }
                    } else if (cg0 + 8 * g < a.Cout) {
                        *reinterpret_cast<Slot*>(dst_of(pix, cg0 + 8 * g)) = outp[k2];
                    }
                }
            }
            // one statistics row per step: four rows per 8x16 tile, one for each pair of pixel rows (the sums live only inside
            // the step, so that they hold no registers while rows are in flight)
            if constexpr (STATS) {
                const int yy0 = ys + 2 * st;
                const int64_t row = ((int64_t)(n * a.tiles_y + yy0 / 8) * a.tiles_x + sx) * 4 + (yy0 % 8) / 2;
#pragma unroll
                for (int k2 = 0; k2 < NI / 2; ++k2) {
                    const int co = cb * BN + 32 * k2 + 8 * g;
                    if (co >= a.Cout) continue;
#pragma unroll
                    for (int r = 0; r < 8; ++r) { ssum[k2][r] = row16_sum(ssum[k2][r]); ssq[k2][r] = row16_sum(ssq[k2][r]); }
                    if (l15 == 0) {
                        float* dst = a.stat_part + row * 2 * a.Cout + co;
                        *reinterpret_cast<float4*>(dst) = make_float4(ssum[k2][0], ssum[k2][1], ssum[k2][2], ssum[k2][3]);
                        *reinterpret_cast<float4*>(dst + 4) = make_float4(ssum[k2][4], ssum[k2][5], ssum[k2][6], ssum[k2][7]);
                        *reinterpret_cast<float4*>(dst + a.Cout) = make_float4(ssq[k2][0], ssq[k2][1], ssq[k2][2], ssq[k2][3]);
                        *reinterpret_cast<float4*>(dst + a.Cout + 4) = make_float4(ssq[k2][4], ssq[k2][5], ssq[k2][6], ssq[k2][7]);
                    }
                }
            }
        };
        for (int s0 = 0; s0 < nsteps; s0 += RING / 2) {
            [&]<int... Q>(std::integer_sequence<int, Q...>) {
                ((s0 + Q < nsteps ? step(std::integral_constant<int, Q>{}, s0 + Q) : (void)0), ...);
            }(std::make_integer_sequence<int, RING / 2>{});
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 OIHW -> swizzled K-row images
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void conv_pack_body(const float* __restrict__ w, int Cout, int Cin, int taps, int seg0, char* __restrict__ wf,
                                               char* __restrict__ wb, int64_t w_so, int64_t w_si, int64_t first, int64_t step) {
    constexpr int EPS = 16 / (int)sizeof(T);
    constexpr int KR = KRow<T>::kElems;
    // forward image: rows = Cout rounded up to 32 per (chunk,tap); K = input channels, each input
    // segment starting on a K-row boundary
    const int cout_pad = (Cout + 31) / 32 * 32;
    const int k0_pad = (seg0 + KR - 1) / KR * KR;                    // K-rows span of segment 0
    const int kf_pad = k0_pad + (Cin - seg0 + KR - 1) / KR * KR;
    const int64_t nf = wf ? (int64_t)(kf_pad / KR) * taps * cout_pad * 8 : 0;
    // backward image: rows = dx channels (logical input channels, rounded up to 32); K = output channels
    const int kb_pad = (Cout + KR - 1) / KR * KR;
    const int rows_b = (Cin + 31) / 32 * 32;
    const int64_t nb = wb ? (int64_t)(kb_pad / KR) * taps * rows_b * 8 : 0;
    for (int64_t i = first; i < nf + nb; i += step) {
        const bool fwd = i < nf;
        // tap fastest: the taps of a (co, ci) pair are contiguous in the fp32 weight, so neighbouring threads read
        // neighbouring floats and a wavefront's eight loads cover one contiguous stretch of the tensor
        int64_t r = fwd ? i : i - nf;
        const int tap = (int)(r % taps); r /= taps;
        const int sl = (int)(r & 7); r >>= 3;
        const int rows = fwd ? cout_pad : rows_b;
        const int row = (int)(r % rows);
        const int chunk = (int)(r / rows);
        float v[EPS];
#pragma unroll
        for (int j = 0; j < EPS; ++j) {
            const int k = chunk * KR + sl * EPS + j;
            int co, ci, t;
            if (fwd) {
                co = row_channel(row); t = tap;
                // K index -> logical input channel (-1 inside the zero padding that ends a segment)
                if (k < k0_pad) ci = k < seg0 ? k : -1;
                else ci = seg0 + (k - k0_pad);
            } else { co = k; ci = row_channel(row); t = taps - 1 - tap; }
            v[j] = (co < Cout && ci >= 0 && ci < Cin) ? w[co * w_so + ci * w_si + t] : 0.f;
        }
        char* dst = (fwd ? wf : wb) + (((int64_t)chunk * taps + tap) * rows + row) * kRowBytes + ((sl ^ (row & 7)) << 4);
        if constexpr (sizeof(T) == 2) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
            *reinterpret_cast<bf16x8*>(dst) = o;
        } else {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kConvThreads) void conv_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int taps,
                                                                 int seg0, char* __restrict__ wf, char* __restrict__ wb,
                                                                 int64_t w_so, int64_t w_si) {
    conv_pack_body<T>(w, Cout, Cin, taps, seg0, wf, wb, w_so, w_si, (int64_t)blockIdx.x * kConvThreads + threadIdx.x,
                      (int64_t)gridDim.x * kConvThreads);
}

// Many weight tensors in one launch: blockIdx.y selects the descriptor, the workgroups of a row share its slots.
template <typename T>
__global__ __launch_bounds__(kConvThreads) void conv_pack_batch_kernel(const segk_pack_desc* __restrict__ descs) {
    const segk_pack_desc d = descs[blockIdx.y];
    conv_pack_body<T>(d.w, d.Cout, d.Cin, d.taps, d.seg0, (char*)d.wf, (char*)d.wb, d.w_so, d.w_si,
                      (int64_t)blockIdx.x * kConvThreads + threadIdx.x, (int64_t)gridDim.x * kConvThreads);
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose2d(kernel 4, stride 2, padding 1) as four phase convolutions.  Output pixel (2m+a, 2n+b) only
// sees kernel rows ky with ky = a+1 (input row m) or ky = 3 (a=0, row m-1) / ky = 0 (a=1, row m+1), likewise for
// columns: every phase (a,b) is a 2x2-tap convolution over the INPUT grid whose result is written to every other
// output row/column.  Tap t = ty*2+tx of phase (a,b): input offset (dy,dx), kernel element (ky,kx).
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline int ct_off(int a, int t) { return t == 0 ? 0 : (a == 0 ? -1 : 1); }
__host__ __device__ inline int ct_k(int a, int t) { return t == 0 ? 1 + a : (a == 0 ? 3 : 0); }

// weight [Cin][Cout][4][4] fp32 -> per phase: forward image (rows = Cout, K = Cin) and data-gradient image
// (rows = Cin, K = Cout), both [K-row][4 taps][rows][128 B] swizzled like conv_pack_kernel's
template <typename T>
__global__ __launch_bounds__(kConvThreads) void convt_pack_kernel(const float* __restrict__ w, int Cin, int Cout,
                                                                  char* __restrict__ wf, char* __restrict__ wb) {
    constexpr int EPS = 16 / (int)sizeof(T);
    constexpr int KR = KRow<T>::kElems;
    const int rows_f = (Cout + 31) / 32 * 32, kf = (Cin + KR - 1) / KR * KR;
    const int rows_b = (Cin + 31) / 32 * 32, kb = (Cout + KR - 1) / KR * KR;
    const int64_t per_f = (int64_t)(kf / KR) * 4 * rows_f * 8, per_b = (int64_t)(kb / KR) * 4 * rows_b * 8;
    const int64_t nf = wf ? 4 * per_f : 0, nb = wb ? 4 * per_b : 0;
    for (int64_t i = (int64_t)blockIdx.x * kConvThreads + threadIdx.x; i < nf + nb; i += (int64_t)gridDim.x * kConvThreads) {
        const bool fwd = i < nf;
        int64_t r = fwd ? i : i - nf;
        const int64_t per = fwd ? per_f : per_b;
        const int phase = (int)(r / per); r -= phase * per;
        const int sl = (int)(r & 7); r >>= 3;
        const int rows = fwd ? rows_f : rows_b;
        const int row = (int)(r % rows); r /= rows;
        const int tap = (int)(r & 3);
        const int chunk = (int)(r >> 2);
        const int ky = ct_k(phase >> 1, tap >> 1), kx = ct_k(phase & 1, tap & 1);
        float v[EPS];
#pragma unroll
        for (int j = 0; j < EPS; ++j) {
            const int k = chunk * KR + sl * EPS + j;
            const int co = fwd ? row_channel(row) : k, ci = fwd ? k : row_channel(row);
            v[j] = (co < Cout && ci < Cin) ? w[(((int64_t)ci * Cout + co) * 4 + ky) * 4 + kx] : 0.f;
        }
        char* dst = (fwd ? wf : wb) + ((int64_t)phase * per + (((int64_t)chunk * 4 + tap) * rows + row) * 8) * 16 + ((sl ^ (row & 7)) << 4);
        if constexpr (sizeof(T) == 2) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
            *reinterpret_cast<bf16x8*>(dst) = o;
        } else {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel.  Workgroup = (output-channel block of one K-row width, input K-row, pixel
// split).  Both MFMA operands are needed "pixel-major" (K = pixels) while memory is channel-major;
// for bf16 the transposing LDS read ds_read_b64_tr_b16 delivers exactly the A/B fragment layout from
// the [pixel][channel] image, for fp32 a lane reads single elements.
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    View vdy;                 // window of dy (identity, or one phase of a transposed convolution's output gradient)
    int tky[9], tkx[9];       // tap offsets + 1 when TAPS == 4
    const char* x0; const char* x1;
    int64_t ld0, ld1;
    int C0, C1;
    int nch0, nch;
    int N, H, W, tiles_x, tiles_y;
    const char* dy; int64_t ldy;
    int Cout, ncb;            // dy channels, output-channel blocks (of KRow elems)
    int splits, tiles_per_split;
    float* slab;              // [splits][ncb*KR][TAPS][nch*KR]
    // volumes (depth > 0): nkd times the workgroups, one share per depth tap kdi; x is read from slice n + kdi - nkd/2 of
    // the volume that holds slice n (zero outside); slab gets a leading [nkd] dimension
    int depth = 0, nkd = 3;
    int xcd_remap = 1;
};

template <typename T> __device__ __forceinline__ int wg_swz(int pix);
template <> __device__ __forceinline__ int wg_swz<bf16>(int pix) { return ((pix >> 1) & 3) << 1; }
template <> __device__ __forceinline__ int wg_swz<float>(int pix) { return (pix & 1) << 2; }

// TDI / TDJ: 16-wide tiles of the output-channel / input-channel side that hold real channels (a K-row has KR/16);
// GI x GJ = 4: wave grid.  <4,4,2,2> is the square 64x64 block (bf16); <2,4,1,4> serves dy with <= 32 channels,
// <4,2,4,1> an input K-row with <= 32 channels, without multiplying the zero half.
template <typename T, int TAPS, int TDI, int TDJ, int GI, int GJ>
__global__ __launch_bounds__(kConvThreads, 2) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int TH = 8;
    constexpr int PAD = TAPS == 1 ? 0 : 1;
    constexpr int HH = TH + 2 * PAD, HW_ = kTW + 2 * PAD, HPIX = HH * HW_, TPIX = TH * kTW;
    constexpr int KR = KRow<T>::kElems;
    constexpr int EPS = 16 / (int)sizeof(T);
    static_assert(GI * GJ == 4 && TDI % GI == 0 && TDJ % GJ == 0 && TDI <= KR / 16 && TDJ <= KR / 16, "bad wgrad tiling");
    constexpr int WTI = TDI / GI, WTJ = TDJ / GJ;   // tiles per wave
    constexpr bool kBf16 = sizeof(T) == 2;

    constexpr int XBYTES = HPIX * kRowBytes, DBYTES = TPIX * kRowBytes;
    // one LDS object, two stages: [x halo 0 | dy 0 | x halo 1 | dy 1]
    __shared__ __attribute__((aligned(16))) char smem[2 * (XBYTES + DBYTES)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wi = wave / GJ, wj = wave % GJ;

    // Workgroups that share a pixel range (same split, different channel-block pairs) read the same dy and x pixels:
    // blocks go to the 8 XCDs round-robin, so renumber them XCD-contiguously over a list with the split slowest --
    // each XCD's L2 then holds one or two splits' pixels instead of every split's (the split count is not a multiple
    // of 8 for the 192- and 768-channel concat inputs).
    int b = blockIdx.x;
    const int npairs = a.nch * a.ncb;
    int kdi, split, chunk, cb;                 // kdi: 0 unless a volume convolution (then 0..2)
    if (a.xcd_remap) {
        const int G = gridDim.x, base = G >> 3, rem = G & 7, x = b & 7;
        b = x * base + (x < rem ? x : rem) + (b >> 3);
        kdi = b / (npairs * a.splits);
        b -= kdi * npairs * a.splits;
        split = b / npairs; b -= split * npairs;
        chunk = b % a.nch;
        cb = b / a.nch;
    } else {                                   // plain order, split fastest
        split = b % a.splits; b /= a.splits;
        chunk = b % a.nch; b /= a.nch;
        cb = b % a.ncb;
        kdi = b / a.ncb;
    }
    const int kd = a.depth > 0 ? kdi - a.nkd / 2 : 0;

    const bool s0 = chunk < a.nch0;
    const char* xbase = s0 ? a.x0 : a.x1;
    const int64_t ldx = s0 ? a.ld0 : a.ld1;
    const int xch = (s0 ? chunk : chunk - a.nch0) * KR;
    const int xcmax = s0 ? a.C0 : a.C1;

    f32x4 acc[TAPS][WTI][WTJ];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < WTI; ++i)
#pragma unroll
            for (int j = 0; j < WTJ; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = a.N * a.tiles_y * a.tiles_x;
    const int t_lo = split * a.tiles_per_split;
    const int t_hi = min(t_lo + a.tiles_per_split, ntiles);
    const unsigned lds0 = lds_address(smem);

    // Stage one spatial tile (input halo + output-gradient tile) into LDS stage `st` by DMA.  The LDS image is
    // linear in the lane index, so the bank swizzle is applied to the SOURCE: slot q = (pixel, s') receives channel
    // slot s' ^ swz(pixel).  Padding reads the 16 zero bytes of kZeroSlot.  The slot -> (row, column, channel slot)
    // arithmetic is tile-independent and kept in registers: hy<<16 | hx<<8 | channel offset, or -1 for "no slot".
    constexpr int XL = (HPIX * 8 + kConvThreads - 1) / kConvThreads, DL = TPIX * 8 / kConvThreads;
    int xinfo[XL], dinfo[DL];
#pragma unroll
    for (int k = 0; k < XL; ++k) {
        const int q = k * kConvThreads + tid;
        const int pix = q >> 3;
        const int hy = pix / HW_, hx = pix - hy * HW_;
        const int sl = (q & 7) ^ wg_swz<T>(hx);   // halo swizzle by column (== by pixel parity for fp32)
        xinfo[k] = (q < HPIX * 8 && xch + sl * EPS < xcmax) ? ((hy << 16) | (hx << 8) | (sl * EPS)) : -1;
    }
#pragma unroll
    for (int k = 0; k < DL; ++k) {
        const int q = k * kConvThreads + tid;
        const int pix = q >> 3, sl = (q & 7) ^ wg_swz<T>(pix);
        dinfo[k] = (cb * KR + sl * EPS < a.Cout) ? (((pix >> 4) << 16) | ((pix & 15) << 8) | (sl * EPS)) : -1;
    }
    const char* const xsrc0 = xbase + (int64_t)xch * (int64_t)sizeof(T);
    const char* const dsrc0 = a.dy + (int64_t)cb * KR * (int64_t)sizeof(T);
    auto stage = [&](int tile, int st) {
        int sp = tile;
        const int tx = sp % a.tiles_x; sp /= a.tiles_x;
        const int ty = sp % a.tiles_y;
        const int n = sp / a.tiles_y;
        const int y0 = ty * TH - PAD, x0 = tx * kTW - PAD;
        const bool slice_ok = a.depth == 0 || (unsigned)(n % a.depth + kd) < (unsigned)a.depth;
        const unsigned xdst = lds0 + st * (XBYTES + DBYTES), ddst = xdst + XBYTES;
#pragma unroll
        for (int k = 0; k < XL; ++k) {
            const int info = xinfo[k];
            const int gy = y0 + (info >> 16), gx = x0 + ((info >> 8) & 255);
            const bool ok = info >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && slice_ok;
            const char* src = ok ? xsrc0 + ((((int64_t)(n + kd) * a.H + gy) * a.W + gx) * ldx + (info & 255)) * (int64_t)sizeof(T)
                                 : reinterpret_cast<const char*>(&kZeroSlot);
            // the last transfer of the halo is partial: lanes past the image's end are masked off
            if (k * kConvThreads + tid < HPIX * 8) glds16_hidden(src, xdst + (k * kConvThreads + wave * 64) * 16);
        }
#pragma unroll
        for (int k = 0; k < DL; ++k) {
            const int info = dinfo[k];
            const int gy = y0 + PAD + (info >> 16), gx = x0 + PAD + ((info >> 8) & 255);
            const bool ok = info >= 0 && gy < a.H && gx < a.W;
            const char* src = ok ? dsrc0 + ((int64_t)a.vdy.pixel(n, gy, gx) * a.ldy + (info & 255)) * (int64_t)sizeof(T)
                                 : reinterpret_cast<const char*>(&kZeroSlot);
            glds16_hidden(src, ddst + (k * kConvThreads + wave * 64) * 16);
        }
    };

    // bf16 operand reads (transposing 8-byte LDS reads).  A K-step is 32 pixels = two tile rows; half h of a fragment
    // is row h: lane (li = lane&15, g) with q = li>>2, p = li&3 supplies the address of pixel (row h, column 4g + q)
    // and of 8-byte unit (tile*4 + p), and receives channel tile*16 + li for K indices 8g + 4h .. +3.  Both operands
    // use this pixel <-> K map.  One read therefore covers one pixel row, and the halo rows of a K-step are shared by
    // the taps that differ in ky: four row reads serve ky = 0, 1, 2.  The swizzle depends on the COLUMN only (eight
    // consecutive columns hit eight different 32-byte bank groups), so rows, taps and K-steps differ from the per-lane
    // base by immediate offsets.  LDS byte addresses for stage 0:
    constexpr int NV = TAPS == 9 ? 3 : (TAPS == 4 ? 4 : 1);
    unsigned adrA[WTI], adrB[NV][WTJ];
    if constexpr (kBf16) {
        const int q = l15 >> 2, p = l15 & 3, pcol = 4 * g + q;
#pragma unroll
        for (int i = 0; i < WTI; ++i) {
            const int u = (wi * WTI + i) * 4 + p;
            adrA[i] = lds0 + XBYTES + pcol * kRowBytes + (((u >> 1) ^ wg_swz<T>(pcol)) << 4) + (u & 1) * 8;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int j = 0; j < WTJ; ++j) {
                // column variant: kx for 3x3 taps; the whole tap (run-time offsets) for the 2x2 phase taps
                const int kx = TAPS == 9 ? v : (TAPS == 4 ? a.tkx[v] : 0), kyr = TAPS == 4 ? a.tky[v] : 0;
                const int hx = pcol + kx, hp = kyr * HW_ + hx, u = (wj * WTJ + j) * 4 + p;
                adrB[v][j] = lds0 + hp * kRowBytes + (((u >> 1) ^ wg_swz<T>(hx)) << 4) + (u & 1) * 8;
            }
    }

    if (t_lo < t_hi) stage(t_lo, 0);
    dma_wait_all();
    __syncthreads();

    if constexpr (kBf16 && TAPS == 9) {
        // 3x3, bf16: a software pipeline over "groups".  Group (K-step kk, kx, j) = the three ky fragments of input-channel
        // tile j at column shift kx (six transposing row reads) and their 3 x WTI MFMAs.  The reads of group g+DEPTH are
        // issued in front of the MFMAs of group g, so every fragment has DEPTH groups of MFMAs to arrive.  A halo row that two ky
        // taps share (as the high half of one fragment and the low half of the next) is read twice, once into each
        // fragment's registers: the LDS has the bandwidth, and the matrix pipe is spared two register moves per MFMA.
        // The pipeline runs across tiles: the barrier that ends a tile sits in front of its last DEPTH groups' MFMAs, whose
        // fragments are in registers by then, and the first DEPTH groups of the next tile are read right behind it.
        typedef __attribute__((address_space(3))) char* lds_cptr;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        constexpr int STAGE = XBYTES + DBYTES;
        constexpr int KSTEPS = TPIX / 32, GPK = 3 * WTJ, NG = KSTEPS * GPK;
        // prefetch distance in groups: 2 where the registers allow it, 1 for the square 64 x 64 block (144 accumulators)
        constexpr int DEPTH = WTI * WTJ >= 4 ? 1 : 2, NS = DEPTH + 1;
        static_assert(NG % NS == 0 && NG > DEPTH, "the fragment slots rotate with period NS");
        auto rd = [&](unsigned adr, int off) {
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)((lds_cptr)(size_t)adr + off));
        };
        auto join = [](const s16x4& lo, const s16x4& hi) {
            s16x8 f;
            f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
            return f;
        };
        s16x8 fb[NS][3];       // [slot = group % NS][ky]
        s16x8 fa[2][WTI];      // [K-step parity][output-channel tile]
        // reads of group gi of the tile staged at LDS offset stoff; the group that opens a K-step also brings that
        // K-step's output-gradient fragments
        auto read_group = [&](auto gtag, unsigned stoff) {
            constexpr int gi = decltype(gtag)::value;
            constexpr int kk = gi / GPK, kx = (gi % GPK) / WTJ, j = gi % WTJ, slot = gi % NS;
            const unsigned b = adrB[kx][j] + stoff;
            s16x4 lo[3], hi[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) lo[ky] = rd(b, (2 * kk + ky) * HW_ * kRowBytes);
            asm volatile("" ::: "memory");        // the second read of a shared row is a read of its own, not a copy
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) hi[ky] = rd(b, (2 * kk + ky + 1) * HW_ * kRowBytes);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) fb[slot][ky] = join(lo[ky], hi[ky]);
            if constexpr (gi % GPK == 0) {
#pragma unroll
                for (int i = 0; i < WTI; ++i) {
                    const unsigned aa = adrA[i] + stoff;
                    fa[kk & 1][i] = join(rd(aa, 2 * kk * kTW * kRowBytes), rd(aa, (2 * kk + 1) * kTW * kRowBytes));
                }
            }
            asm volatile("" ::: "memory");
        };
        auto mma_group = [&](auto gtag) {
            constexpr int gi = decltype(gtag)::value;
            constexpr int kk = gi / GPK, kx = (gi % GPK) / WTJ, j = gi % WTJ, slot = gi % NS;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int i = 0; i < WTI; ++i)
                    acc[ky * 3 + kx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fb[slot][ky]), __builtin_bit_cast(bf16x8, fa[kk & 1][i]), acc[ky * 3 + kx][i][j], 0, 0, 0);
        };
        auto for_groups = [&](auto&& body) {
            [&]<int... G>(std::integer_sequence<int, G...>) { (body(std::integral_constant<int, G>{}), ...); }(
                std::make_integer_sequence<int, NG>{});
        };
        if (t_lo < t_hi) {
            if (t_lo + 1 < t_hi) stage(t_lo + 1, 1);
            read_group(std::integral_constant<int, 0>{}, 0u);
            if constexpr (DEPTH == 2) read_group(std::integral_constant<int, 1>{}, 0u);
        }
        unsigned stoff = 0;
        for (int tile = t_lo; tile < t_hi; ++tile, stoff = STAGE - stoff) {
            const bool has_next = tile + 1 < t_hi;
            for_groups([&](auto gtag) {
                constexpr int gi = decltype(gtag)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (gi + DEPTH < NG) {
                    read_group(std::integral_constant<int, gi + DEPTH>{}, stoff);
                } else if constexpr (gi == NG - DEPTH) {
                    if (has_next) {
                        // every fragment of this tile is in registers; the next tile has landed in the other stage
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        dma_wait_all();
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        if (tile + 2 < t_hi) stage(tile + 2, stoff ? 1 : 0);      // into the stage just released
                        read_group(std::integral_constant<int, 0>{}, STAGE - stoff);
                        if constexpr (DEPTH == 2) read_group(std::integral_constant<int, 1>{}, STAGE - stoff);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                mma_group(gtag);
            });
        }
    } else {
    int st = 0;
    for (int tile = t_lo; tile < t_hi; ++tile, st ^= 1) {
        // next tile's DMA goes out first and lands in the other stage while this tile is multiplied
        if (tile + 1 < t_hi) stage(tile + 1, st ^ 1);
        const char* s_x = smem + st * (XBYTES + DBYTES);
        const char* s_dy = s_x + XBYTES;

        if constexpr (kBf16) {
            // K-step = 32 pixels = two tile rows; every address is a per-lane base (advanced with the stage) plus a
            // compile-time offset, so the loop issues no address arithmetic between the MFMAs.
            typedef __attribute__((address_space(3))) char* lds_cptr;
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
            auto row_read = [&](unsigned base, int row, int row_pixels) {
                return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)((lds_cptr)(size_t)base + row * row_pixels * kRowBytes));
            };
            auto join = [](const s16x4& lo, const s16x4& hi) {
                s16x8 f;
                f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
                return f;
            };
            // fragments that open a K-step are read while the previous K-step's last MFMAs issue (software pipeline
            // across K-steps); only the first K-step of a tile, whose stage has just been published, reads them up front
            s16x8 fa[WTI];
            s16x4 rf0[4];
#pragma unroll
            for (int i = 0; i < WTI; ++i) fa[i] = join(row_read(adrA[i], 0, kTW), row_read(adrA[i], 1, kTW));
            if constexpr (TAPS == 9) {
#pragma unroll
                for (int r = 0; r < 4; ++r) rf0[r] = row_read(adrB[0][0], r, HW_);
            }
#pragma unroll 2
            for (int kk = 0; kk < TPIX / 32; ++kk) {
                const bool more = kk + 1 < TPIX / 32;
                if constexpr (TAPS == 9) {
                    s16x8 fan[WTI];
                    s16x4 rfn[4];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int j = 0; j < WTJ; ++j) {
                            s16x4 rf[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) rf[r] = (kx == 0 && j == 0) ? rf0[r] : row_read(adrB[kx][j], r, HW_);
                            if (kx == 2 && j == WTJ - 1 && more) {
                                // next K-step = two rows further down both images
#pragma unroll
                                for (int i = 0; i < WTI; ++i) fan[i] = join(row_read(adrA[i], 2, kTW), row_read(adrA[i], 3, kTW));
#pragma unroll
                                for (int r = 0; r < 4; ++r) rfn[r] = row_read(adrB[0][0], r + 2, HW_);
                            }
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky) {
                                const s16x8 fb = join(rf[ky], rf[ky + 1]);
#pragma unroll
                                for (int i = 0; i < WTI; ++i)
                                    acc[ky * 3 + kx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                        __builtin_bit_cast(bf16x8, fb), __builtin_bit_cast(bf16x8, fa[i]), acc[ky * 3 + kx][i][j], 0, 0, 0);
                            }
                        }
                    if (more) {
#pragma unroll
                        for (int i = 0; i < WTI; ++i) fa[i] = fan[i];
#pragma unroll
                        for (int r = 0; r < 4; ++r) rf0[r] = rfn[r];
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < TAPS; ++t)
#pragma unroll
                        for (int j = 0; j < WTJ; ++j) {
                            const s16x8 fb = join(row_read(adrB[t][j], 0, HW_), row_read(adrB[t][j], 1, HW_));
#pragma unroll
                            for (int i = 0; i < WTI; ++i)
                                acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb),
                                                                                      __builtin_bit_cast(bf16x8, fa[i]), acc[t][i][j], 0, 0, 0);
                        }
                    if (more) {
#pragma unroll
                        for (int i = 0; i < WTI; ++i) fa[i] = join(row_read(adrA[i], 2, kTW), row_read(adrA[i], 3, kTW));
                    }
                }
                // advance to the next K-step: two tile rows; after the last one, to the start of the other stage
                const bool last = kk == TPIX / 32 - 1;
                const int back = st == 0 ? (XBYTES + DBYTES) : -(XBYTES + DBYTES);
                const int stepA = last ? back - (TPIX / 32 - 1) * 32 * kRowBytes : 32 * kRowBytes;
                const int stepB = last ? back - (TPIX / 32 - 1) * 2 * HW_ * kRowBytes : 2 * HW_ * kRowBytes;
#pragma unroll
                for (int i = 0; i < WTI; ++i) adrA[i] += stepA;
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int j = 0; j < WTJ; ++j) adrB[v][j] += stepB;
            }
        } else {
            // fp32: K-step = 4 pixels; lane (m = lane&15, k = lane>>4) reads one element of each operand.
#pragma unroll 1
            for (int kk = 0; kk < TPIX / 4; ++kk) {
                const int pix = kk * 4 + g;
                float fa[WTI];
#pragma unroll
                for (int i = 0; i < WTI; ++i) {
                    const int c = (wi * WTI + i) * 16 + l15;
                    fa[i] = *reinterpret_cast<const float*>(&s_dy[pix * kRowBytes + ((((c >> 2) ^ wg_swz<T>(pix)) << 4) | ((c & 3) << 2))]);
                }
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int ky = TAPS == 9 ? t / 3 : (TAPS == 4 ? a.tky[t] : 0), kx = TAPS == 9 ? t % 3 : (TAPS == 4 ? a.tkx[t] : 0);
                    const int hx = (pix & 15) + kx, hp = ((pix >> 4) + ky) * HW_ + hx;
#pragma unroll
                    for (int j = 0; j < WTJ; ++j) {
                        const int c = (wj * WTJ + j) * 16 + l15;
                        const float fb = *reinterpret_cast<const float*>(&s_x[hp * kRowBytes + ((((c >> 2) ^ wg_swz<T>(hx)) << 4) | ((c & 3) << 2))]);
#pragma unroll
                        for (int i = 0; i < WTI; ++i) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, fa[i], acc[t][i][j], 0, 0, 0);
                    }
                }
            }
        }
        // the other stage's DMA must have landed, and everyone must be done reading this stage, before the swap
        __builtin_amdgcn_sched_barrier(0);
        dma_wait_all();
        __syncthreads();
    }
    }

    // partial slab: slab[split][co][tap][ci].  The input-channel tile is the MFMA's row operand, so a lane holds four
    // consecutive input channels of one output channel: one 16-byte store each.
    const int cin_tot = a.nch * KR;
    const int64_t co_tot = (int64_t)a.ncb * KR;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < WTI; ++i)
#pragma unroll
            for (int j = 0; j < WTJ; ++j) {
                const int co = cb * KR + (wi * WTI + i) * 16 + l15;
                const int ci = chunk * KR + (wj * WTJ + j) * 16 + 4 * g;
                *reinterpret_cast<float4*>(&a.slab[((((int64_t)kdi * a.splits + split) * co_tot + co) * TAPS + t) * cin_tot + ci]) =
                    make_float4(acc[t][i][j][0], acc[t][i][j][1], acc[t][i][j][2], acc[t][i][j][3]);
            }
}

// dw[co][ci][tap] (OIHW) (+)= sum_splits slab[s][co][tap][ci]
// P lanes share one output: lane = p * (64/P) + c handles output (wave * 64/P + c) and adds splits p, p+P, ...; a
// butterfly over p finishes the sum (fixed order: deterministic).  P = 8 for the layers with few outputs and hundreds
// of splits (a serial chain of 512 loads per thread otherwise), P = 1 where every thread has only a few splits.
template <int P>
__global__ __launch_bounds__(kConvThreads) void conv_wgrad_reduce_kernel(const float* __restrict__ slab, int splits, int64_t co_tot,
                                                                         int taps, int cin_tot, int cout_real, int cin_real,
                                                                         int seg0, int k0_pad, float* __restrict__ dw,
                                                                         int accumulate, int ct_cout, int ct_k0, int ct_k1,
                                                                         int ct_k2, int ct_k3, int64_t dw_so, int64_t dw_si, int nkd) {
    constexpr int OPW = 64 / P;   // outputs per wavefront
    const int64_t total = (int64_t)cout_real * cin_real * taps * nkd;
    const int lane = threadIdx.x & 63, p = lane / OPW, c = lane % OPW;
    const int64_t waves = (int64_t)gridDim.x * (kConvThreads / 64);
    for (int64_t w = (int64_t)blockIdx.x * (kConvThreads / 64) + (threadIdx.x >> 6); w * OPW < total; w += waves) {
        const int64_t i = w * OPW + c;
        const bool live = i < total;
        // ci fastest so that slab reads are coalesced
        const int ci = live ? (int)(i % cin_real) : 0;
        int64_t r = live ? i / cin_real : 0;
        const int t = (int)(r % taps); r /= taps;
        const int kdi = (int)(r % nkd);          // depth tap of a volume convolution (nkd = 3), else 0
        const int co = (int)(r / nkd);
        const int kc = ci < seg0 ? ci : k0_pad + (ci - seg0);   // position in the K-row-padded channel space
        float s = 0.f;
        if (live)
            for (int k = p; k < splits; k += P) s += slab[((((int64_t)kdi * splits + k) * co_tot + co) * taps + t) * cin_tot + kc];
#pragma unroll
        for (int o = OPW; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
        if (!live || p != 0) continue;
        // ordinary convolution: OIHW [co][ci][tap].  Transposed convolution (ct_cout > 0): the weight is
        // [ci][co][4][4] and tap t of this phase is kernel element ct_k{t}.
        float* d = ct_cout > 0 ? dw + ((int64_t)ci * ct_cout + co) * 16 + (t == 0 ? ct_k0 : t == 1 ? ct_k1 : t == 2 ? ct_k2 : ct_k3)
                               : dw + co * dw_so + ci * dw_si + kdi * taps + t;
        *d = accumulate ? *d + s : s;
    }
}

// The same reduction with four consecutive input channels per lane (16-byte slab reads): for channel counts and segment
// boundaries that are multiples of 4, i.e. every layer but the patch rows of a first convolution.
template <int P>
__global__ __launch_bounds__(kConvThreads) void conv_wgrad_reduce4_kernel(const float* __restrict__ slab, int splits, int64_t co_tot,
                                                                          int taps, int cin_tot, int cout_real, int cin_real,
                                                                          int seg0, int k0_pad, float* __restrict__ dw,
                                                                          int accumulate, int ct_cout, int ct_k0, int ct_k1,
                                                                          int ct_k2, int ct_k3, int64_t dw_so, int64_t dw_si, int nkd) {
    constexpr int OPW = 64 / P;   // output vectors per wavefront
    const int cv = cin_real / 4;
    const int64_t total = (int64_t)cout_real * cv * taps * nkd;
    const int lane = threadIdx.x & 63, p = lane / OPW, c = lane % OPW;
    const int64_t waves = (int64_t)gridDim.x * (kConvThreads / 64);
    const int64_t split_stride = co_tot * taps * cin_tot;
    for (int64_t w = (int64_t)blockIdx.x * (kConvThreads / 64) + (threadIdx.x >> 6); w * OPW < total; w += waves) {
        const int64_t i = w * OPW + c;
        const bool live = i < total;
        const int ci = live ? (int)(i % cv) * 4 : 0;
        int64_t r = live ? i / cv : 0;
        const int t = (int)(r % taps); r /= taps;
        const int kdi = (int)(r % nkd);
        const int co = (int)(r / nkd);
        const int kc = ci < seg0 ? ci : k0_pad + (ci - seg0);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            const float* src = slab + (((int64_t)kdi * splits * co_tot + co) * taps + t) * cin_tot + kc;
            int k = p;
            // two independent chains of loads in flight per lane
            float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (; k + P < splits; k += 2 * P) {
                const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)k * split_stride);
                const float4 b = *reinterpret_cast<const float4*>(src + (int64_t)(k + P) * split_stride);
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
            }
            if (k < splits) {
                const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)k * split_stride);
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            s.x += s2.x; s.y += s2.y; s.z += s2.z; s.w += s2.w;
        }
#pragma unroll
        for (int o = OPW; o < 64; o <<= 1) {
            s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
        }
        if (!live || p != 0) continue;
        const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float* d = ct_cout > 0 ? dw + ((int64_t)(ci + e) * ct_cout + co) * 16 + (t == 0 ? ct_k0 : t == 1 ? ct_k1 : t == 2 ? ct_k2 : ct_k3)
                                   : dw + co * dw_so + (ci + e) * dw_si + kdi * taps + t;
            *d = accumulate ? *d + v[e] : v[e];
        }
    }
}

static void launch_wgrad_reduce(hipStream_t st, const float* slab, int splits, int64_t co_tot, int taps, int cin_tot, int cout_real,
                                int cin_real, int seg0, int k0_pad, float* dw, int accumulate, int ct_cout = 0, int k0 = 0, int k1 = 0,
                                int k2 = 0, int k3 = 0, int64_t dw_so = 0, int64_t dw_si = 0, int nkd = 1) {
    const int64_t total = (int64_t)cout_real * cin_real * taps * nkd;
    if (dw_so == 0) { dw_so = (int64_t)cin_real * taps; dw_si = taps; }   // contiguous OIHW
    static const bool scalar_only = getenv("SEGK_WGRAD_REDUCE_SCALAR") != nullptr;   // experiment switch
    if (cin_real % 4 == 0 && seg0 % 4 == 0 && k0_pad % 4 == 0 && cin_tot % 4 == 0 && !scalar_only) {
        const int64_t vecs = total / 4;
        if (splits >= 32) {
            int64_t grid = (vecs * 8 + kConvThreads - 1) / kConvThreads;
            if (grid > 16384) grid = 16384;
            hipLaunchKernelGGL(conv_wgrad_reduce4_kernel<8>, dim3((unsigned)grid), dim3(kConvThreads), 0, st, slab, splits, co_tot, taps, cin_tot,
                               cout_real, cin_real, seg0, k0_pad, dw, accumulate, ct_cout, k0, k1, k2, k3, dw_so, dw_si, nkd);
        } else {
            int64_t grid = (vecs + kConvThreads - 1) / kConvThreads;
            if (grid > 8192) grid = 8192;
            hipLaunchKernelGGL(conv_wgrad_reduce4_kernel<1>, dim3((unsigned)grid), dim3(kConvThreads), 0, st, slab, splits, co_tot, taps, cin_tot,
                               cout_real, cin_real, seg0, k0_pad, dw, accumulate, ct_cout, k0, k1, k2, k3, dw_so, dw_si, nkd);
        }
        return;
    }
    if (splits >= 32) {
        int64_t grid = (total * 8 + kConvThreads - 1) / kConvThreads;
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel<8>, dim3((unsigned)grid), dim3(kConvThreads), 0, st, slab, splits, co_tot, taps, cin_tot,
                           cout_real, cin_real, seg0, k0_pad, dw, accumulate, ct_cout, k0, k1, k2, k3, dw_so, dw_si, nkd);
    } else {
        int64_t grid = (total + kConvThreads - 1) / kConvThreads;
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel<1>, dim3((unsigned)grid), dim3(kConvThreads), 0, st, slab, splits, co_tot, taps, cin_tot,
                           cout_real, cin_real, seg0, k0_pad, dw, accumulate, ct_cout, k0, k1, k2, k3, dw_so, dw_si, nkd);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
static inline int krow(int dtype) { return dtype == SEGK_BF16 ? 64 : 32; }
// output channels per workgroup: as wide as divides the (32-padded) channel count, so that the input halo is
// staged as few times as possible; 96 serves the 96- and 192-channel data gradients of the concat inputs
static inline int pick_bn(int64_t cout_pad) {
    if (cout_pad % 128 == 0) return 128;
    if (cout_pad % 96 == 0) return 96;
    return cout_pad % 64 == 0 ? 64 : 32;
}

template <typename T, int TAPS>
static void launch_fwd_t(const ConvArgs& a, int bn, hipStream_t st) {
    const unsigned grid = (unsigned)((int64_t)a.N * a.tiles_x * a.tiles_y * (a.cout_pad / bn));
    // (a 16x16-pixel tile on 8 waves, <T, TAPS, 16, 128, 4, 2>, halves the weight traffic into LDS per FLOP but measured
    // 2-5 % slower on the UNet-VGG layers: one 8-wave barrier domain per CU instead of two independent 4-wave ones)
    if constexpr (TAPS == 9 && sizeof(T) == 2) {
        // channel blocks narrower than 128: the high-occupancy variants (one halo buffer, 3-4 workgroups per CU) win at
        // every depth of K (measured up to 5 K-rows: 929 -> 793 us for 320 -> 64 channels at 384^2); a 128-wide block does
        // not fit three times into a CU's LDS (56 KB each) and gains nothing from the single halo buffer alone
        if (bn != 128) {
            if (bn == 96) hipLaunchKernelGGL((conv_fwd_kernel<T, TAPS, 8, 96, 4, 1, 3>), dim3(grid), dim3(kConvThreads), 0, st, a);
            else if (bn == 64) hipLaunchKernelGGL((conv_fwd_kernel<T, TAPS, 8, 64, 2, 2, 4>), dim3(grid), dim3(kConvThreads), 0, st, a);
            else hipLaunchKernelGGL((conv_fwd_kernel<T, TAPS, 8, 32, 4, 1, 4>), dim3(grid), dim3(kConvThreads), 0, st, a);
            return;
        }
    }
    if constexpr (sizeof(T) == 2 && TAPS == 9) {
        static const bool no_pipe = getenv("SEGK_NO_PIPE") != nullptr;
        const bool full_rows = a.C0 % 64 == 0 && a.C1 % 64 == 0;
        if (bn == 128 && full_rows && !no_pipe) {
            hipLaunchKernelGGL((conv_fwd_kernel<T, TAPS, 8, 128, 2, 2, 2, true>), dim3(grid), dim3(kConvThreads), 0, st, a);
            return;
        }
    }
    if (bn == 128)
        hipLaunchKernelGGL((conv_fwd_kernel<T, TAPS, 8, 128, 2, 2>), dim3(grid), dim3(kConvThreads), 0, st, a);
    else if (bn == 96)
        hipLaunchKernelGGL((conv_fwd_kernel<T, TAPS, 8, 96, 4, 1>), dim3(grid), dim3(kConvThreads), 0, st, a);
    else if (bn == 64)
        hipLaunchKernelGGL((conv_fwd_kernel<T, TAPS, 8, 64, 2, 2>), dim3(grid), dim3(kConvThreads), 0, st, a);
    else
        hipLaunchKernelGGL((conv_fwd_kernel<T, TAPS, 8, 32, 4, 1>), dim3(grid), dim3(kConvThreads), 0, st, a);
}

static void launch_fwd_any(int dtype, int taps, const ConvArgs& a, int bn, hipStream_t st) {
    if (dtype == SEGK_BF16) {
        if (taps == 9) launch_fwd_t<bf16, 9>(a, bn, st);
        else if (taps == 4) launch_fwd_t<bf16, 4>(a, bn, st);
        else launch_fwd_t<bf16, 1>(a, bn, st);
    } else {
        if (taps == 9) launch_fwd_t<float, 9>(a, bn, st);
        else if (taps == 4) launch_fwd_t<float, 4>(a, bn, st);
        else launch_fwd_t<float, 1>(a, bn, st);
    }
}

// The thin-layer kernel (conv_direct_kernel) serves bf16 3x3 convolutions whose whole packed weight image fits in LDS and
// whose input has one or two K-rows made of whole 32-channel k-substeps.  Returns false when the layer is not one of its.
static bool launch_direct(const ConvArgs& c, int bn, hipStream_t st) {
    static const bool off = getenv("SEGK_NO_DIRECT") != nullptr;          // experiment switch: always the LDS-staged kernels
    if (off || c.depth > 0 || c.nb_part || c.cout_pad > 128 || c.W % kTW || c.H % 2 || c.C0 % 32 || c.C1 % 32) return false;
    if (c.stat_part && (c.H % 8 || c.stat_rpt != 4)) return false;   // its statistics come one row per pair of pixel rows
    if (c.vin.sy != 1 || c.vin.sx != 1 || c.vout.sy != 1 || c.vout.sx != 1) return false;
    int nk0, nk1;
    if (c.C1 > 0) { nk0 = c.C0 / 32; nk1 = c.C1 / 32; if (nk0 > 2 || nk1 > 2) return false; }
    else if (c.C0 <= 64) { nk0 = c.C0 / 32; nk1 = 0; }
    else if (c.C0 <= 128) { nk0 = 2; nk1 = c.C0 / 32 - 2; }
    else return false;
    // 128 output channels run as two blocks of 64 (the accumulators of one 128-wide block leave no registers for rows in flight)
    const int ncb = c.cout_pad == 128 ? 2 : 1;
    const int ni = c.cout_pad / ncb / 16;
    DirectArgs a;
    a.seg0 = c.x0; a.seg1 = c.C1 > 0 ? c.x1 : nullptr;
    a.ldb0 = c.ld0 * 2; a.ldb1 = c.ld1 * 2; a.off1 = 128;
    a.N = c.N; a.H = c.H; a.W = c.W; a.strips_x = c.W / kTW;
    a.w = c.w; a.Cout = c.Cout; a.bias = c.bias;
    a.y = c.y; a.ldy = c.ldy; a.y1 = c.y1; a.ldy1 = c.ldy1; a.split = c.split; a.accumulate = c.accumulate; a.accumulate1 = c.accumulate1;
    a.mask = c.mask; a.ld_mask = c.ld_mask; a.mask_c = c.mask_c; a.relu = c.relu;
    a.stat_part = c.stat_part; a.tiles_x = c.tiles_x; a.tiles_y = c.tiles_y;
    static const bool no_pairs = getenv("SEGK_DIRECT_NO_PAIRS") != nullptr;   // experiment switch
    a.pair_first = 99;
    if (!no_pairs) {
        if (!c.y1 || c.split % 64 == 0) a.pair_first = 0;
        else if (c.split % 32 == 0) a.pair_first = c.split / 32;
    }
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    // rows per unit: a multiple of 8 (statistics rows never span units) that balances the units over the waves with the
    // least work in the busiest one (two extra input rows per unit)
    const int64_t waves = (int64_t)cus * kDirectWaves;
    int best = 8;
    int64_t best_cost = INT64_MAX;
    for (int seg = 8; seg <= 128; seg += 8) {
        const int64_t units = (int64_t)c.N * ncb * ((c.H + seg - 1) / seg) * a.strips_x;
        const int64_t cost = ((units + waves - 1) / waves) * (seg + 4);
        if (cost < best_cost) { best_cost = cost; best = seg; }
    }
    a.seg_rows = best; a.segs_y = (c.H + best - 1) / best;
    const int64_t units = (int64_t)c.N * ncb * a.segs_y * a.strips_x;
    if (units >= (1ll << 31)) return false;
    a.units = (int)units;
    const dim3 grid((unsigned)cus), block(64 * kDirectWaves);
#define SEGK_DIRECT_CASE(K0, K1, I, CB, P, PS, PM, AB)                                                      \
    if (nk0 == K0 && nk1 == K1 && ni == I && ncb == CB) {                                                     \
        if (a.stat_part && a.mask) return false;                                                              \
        if (a.stat_part) {                                                                                    \
            if constexpr (PS >= 0) hipLaunchKernelGGL((conv_direct_kernel<K0, K1, I, (PS >= 0 ? PS : 1), true, false, AB, CB>), grid, block, 0, st, a); \
            else return false;                                                                                \
        } else if (a.mask) {                                                                                  \
            if constexpr (PM >= 0) hipLaunchKernelGGL((conv_direct_kernel<K0, K1, I, (PM >= 0 ? PM : 1), false, true, AB, CB>), grid, block, 0, st, a); \
            else return false;                                                                                \
        } else hipLaunchKernelGGL((conv_direct_kernel<K0, K1, I, P, false, false, AB, CB>), grid, block, 0, st, a); \
        return true;                                                                                          \
    }
    // (k-substeps of K-row 0, of K-row 1, 16-channel output tiles per block, blocks, steps of prefetch, the same with
    // statistics and with a mask (-1: not served), weight-fragment buffers)
    SEGK_DIRECT_CASE(2, 0, 4, 1, 2, 2, 1, 2)      // 64 -> 64
    SEGK_DIRECT_CASE(2, 0, 4, 2, 1, -1, -1, 2)    // 64 -> 128 (with statistics the LDS-staged kernel is as fast)
    SEGK_DIRECT_CASE(1, 2, 2, 1, 1, 1, -1, 2)     // [32 | 64] -> 32
    SEGK_DIRECT_CASE(2, 1, 2, 1, 1, 1, -1, 2)     // 96 -> 32 from one tensor
    SEGK_DIRECT_CASE(1, 0, 6, 1, 3, -1, 1, 1)     // 32 -> 96   (data gradient of [32 | 64] -> 32)
    SEGK_DIRECT_CASE(1, 0, 4, 2, 3, -1, 3, 2)     // 32 -> 128  (data gradient of 128 -> 32)
    SEGK_DIRECT_CASE(1, 0, 4, 1, 3, 2, 3, 2)      // 32 -> 64
    SEGK_DIRECT_CASE(1, 0, 2, 1, 3, 3, 3, 2)      // 32 -> 32
#undef SEGK_DIRECT_CASE
    return false;
}

static int wgrad_splits(int64_t ntiles, int64_t ncb, int64_t nch) {
    // Two workgroups fit a CU (LDS and registers), so 512 are resident at once.  Exactly one resident round has
    // no tail and keeps the partial-slab traffic (147 KB written and read back per workgroup) at its minimum;
    // `target` can be overridden for experiments.
    static const int64_t target = getenv("SEGK_WGRAD_BLOCKS") ? atoll(getenv("SEGK_WGRAD_BLOCKS")) : 512;
    int64_t s = target / (ncb * nch);
    if (s > ntiles) s = ntiles;
    if (s > 512) s = 512;
    if (s < 1) s = 1;
    // even out the tile ranges
    const int64_t per = (ntiles + s - 1) / s;
    return (int)((ntiles + per - 1) / per);
}

template <int TAPS>
static void launch_wgrad_t(int dtype, const WgradArgs& a, unsigned grid, hipStream_t st, int64_t Cout, int64_t C0) {
    // tiling variant: skip the zero half of a 64-channel bf16 K-row when dy (or a single-K-row input) has <= 32 channels
    const bool thin_co = dtype == SEGK_BF16 && Cout <= 32;
    const bool thin_ci = dtype == SEGK_BF16 && !thin_co && a.nch == 1 && C0 <= 32;
    static const int wave_layout = getenv("SEGK_WGRAD_LAYOUT") ? atoi(getenv("SEGK_WGRAD_LAYOUT")) : 22;   // experiment switch
    if (dtype == SEGK_F32)
        hipLaunchKernelGGL((conv_wgrad_kernel<float, TAPS, 2, 2, 2, 2>), dim3(grid), dim3(kConvThreads), 0, st, a);
    else if (thin_co)
        hipLaunchKernelGGL((conv_wgrad_kernel<bf16, TAPS, 2, 4, 1, 4>), dim3(grid), dim3(kConvThreads), 0, st, a);
    else if (thin_ci)
        hipLaunchKernelGGL((conv_wgrad_kernel<bf16, TAPS, 4, 2, 4, 1>), dim3(grid), dim3(kConvThreads), 0, st, a);
    else if (wave_layout == 14)
        hipLaunchKernelGGL((conv_wgrad_kernel<bf16, TAPS, 4, 4, 1, 4>), dim3(grid), dim3(kConvThreads), 0, st, a);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<bf16, TAPS, 4, 4, 2, 2>), dim3(grid), dim3(kConvThreads), 0, st, a);
}
static void launch_wgrad_any(int dtype, int taps, const WgradArgs& a, unsigned grid, hipStream_t st, int64_t Cout, int64_t C0) {
    if (taps == 9) launch_wgrad_t<9>(dtype, a, grid, st, Cout, C0);
    else if (taps == 4) launch_wgrad_t<4>(dtype, a, grid, st, Cout, C0);
    else launch_wgrad_t<1>(dtype, a, grid, st, Cout, C0);
}

static int check_segments(const char* who, int dtype, const void* x0, int64_t ld0, int64_t C0, const void* x1, int64_t ld1,
                          int64_t C1) {
    const int kr = krow(dtype);
    if (dtype != SEGK_F32 && dtype != SEGK_BF16) return fail(SEGK_EINVAL, "%s: unknown dtype %d", who, dtype);
    (void)kr;
    if (!x0 || C0 <= 0 || C0 % 8) return fail(SEGK_ESHAPE, "%s: segment 0 channels %lld must be a positive multiple of 8", who, (long long)C0);
    if (ld0 < C0 || !aligned16(x0) || (ld0 * esize(dtype)) % 16) return fail(SEGK_EALIGN, "%s: segment 0 pointer/ld misaligned or ld < C", who);
    if (C1 != 0) {
        if (!x1 || C1 < 0 || C1 % 8) return fail(SEGK_ESHAPE, "%s: segment 1 channels %lld must be a multiple of 8", who, (long long)C1);
        if (ld1 < C1 || !aligned16(x1) || (ld1 * esize(dtype)) % 16) return fail(SEGK_EALIGN, "%s: segment 1 pointer/ld misaligned or ld < C", who);
    }
    return SEGK_OK;
}

}  // namespace segk

using namespace segk;

extern "C" int segk_conv_krow(int dtype) { return krow(dtype); }

extern "C" size_t segk_conv_packed_bytes(int dtype, int64_t Cout, int64_t Cin, int taps, int64_t seg0) {
    if (Cout <= 0 || Cin <= 0 || (taps != 1 && taps != 9) || seg0 < 0 || seg0 > Cin) return 0;
    const int kr = krow(dtype);
    if (seg0 == 0) seg0 = Cin;
    const int64_t kf = round_up(seg0, kr) + round_up(Cin - seg0, kr);
    const int64_t f = kf / kr * taps * round_up(Cout, 32) * kRowBytes;
    const int64_t b = round_up(Cout, kr) / kr * taps * round_up(Cin, 32) * kRowBytes;
    return (size_t)(f > b ? f : b);
}

extern "C" int segk_conv_pack(const float* w_oihw, int64_t Cout, int64_t Cin, int taps, int64_t seg0, int dtype, void* wf,
                              void* wb, int dev, void* stream) {
    SEGK_REQUIRE(w_oihw && (wf || wb), SEGK_EINVAL, "conv_pack: null pointer");
    SEGK_REQUIRE(taps == 1 || taps == 9, SEGK_ESHAPE, "conv_pack: taps must be 1 or 9");
    SEGK_REQUIRE(dtype == SEGK_F32 || dtype == SEGK_BF16, SEGK_EINVAL, "conv_pack: unknown dtype");
    SEGK_REQUIRE(Cout > 0 && Cin > 0 && seg0 >= 0 && seg0 <= Cin, SEGK_ESHAPE, "conv_pack: bad channel counts (Cout=%lld Cin=%lld seg0=%lld)",
                 (long long)Cout, (long long)Cin, (long long)seg0);
    SEGK_REQUIRE((!wf || aligned16(wf)) && (!wb || aligned16(wb)), SEGK_EALIGN, "conv_pack: packed buffers must be 16-byte aligned");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    if (seg0 == 0) seg0 = Cin;
    const int64_t kf = round_up(seg0, kr) + round_up(Cin - seg0, kr);
    const int64_t slots = ((wf ? kf / kr * round_up(Cout, 32) : 0) + (wb ? round_up(Cout, kr) / kr * round_up(Cin, 32) : 0)) * taps * 8;
    int64_t grid = (slots + kConvThreads - 1) / kConvThreads;
    if (grid > 4096) grid = 4096;
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((conv_pack_kernel<T>), dim3((unsigned)grid), dim3(kConvThreads), 0, (hipStream_t)stream,
                                                     w_oihw, (int)Cout, (int)Cin, taps, (int)seg0, (char*)wf, (char*)wb, Cin * taps, (int64_t)taps));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_conv_pack_batch(int dtype, const segk_pack_desc* descs_dev, int64_t n, int64_t blocks_per_desc, int dev, void* stream) {
    SEGK_REQUIRE(descs_dev && n > 0 && n < 65536 && blocks_per_desc > 0 && blocks_per_desc <= 1024, SEGK_EINVAL, "conv_pack_batch: bad arguments");
    SEGK_REQUIRE(dtype == SEGK_F32 || dtype == SEGK_BF16, SEGK_EINVAL, "conv_pack_batch: unknown dtype");
    SEGK_USE_DEVICE(dev);
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((conv_pack_batch_kernel<T>), dim3((unsigned)blocks_per_desc, (unsigned)n), dim3(kConvThreads), 0,
                                                     (hipStream_t)stream, descs_dev));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// the norm-backward sums of segk_conv_dgrad_norm (all null for segk_conv_fwd)
struct NormBwdSums { const void* z; int64_t ldz; const float* stats; const float* gamma; const float* beta; float* partials; };

static int conv_fwd_impl(int dtype, int taps, const void* x0, int64_t ld0, int64_t C0, const void* x1, int64_t ld1, int64_t C1,
                         int64_t N, int64_t H, int64_t W, const void* w_packed, int64_t Cout, const float* bias, void* y,
                         int64_t ldy, void* y1, int64_t ldy1, int64_t split, const void* mask, int64_t ld_mask, int64_t mask_channels,
                         float* stat_partials, int flags, const NormBwdSums& nb, int dev, void* stream) {
    if (int rc = check_segments("conv_fwd", dtype, x0, ld0, C0, x1, ld1, C1)) return rc;
    SEGK_REQUIRE(!y1 || (split > 0 && split < Cout && split % 8 == 0 && ldy >= split && ldy1 >= Cout - split && aligned16(y1) &&
                         (ldy1 * esize(dtype)) % 16 == 0 && (!mask || mask_channels <= split)),
                 SEGK_ESHAPE, "conv_fwd: second output needs 0 < split < Cout, split a multiple of 8, aligned y1, and a mask (if any) within the first output");
    SEGK_REQUIRE(taps == 1 || taps == 9, SEGK_ESHAPE, "conv_fwd: taps must be 1 or 9");
    // the halo tables pack (pixel index << 3 | slot) into a signed 32-bit register
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0 && N * H * W < (1ll << 28), SEGK_ESHAPE, "conv_fwd: bad N/H/W (at most 2^28 - 1 pixels per launch)");
    SEGK_REQUIRE(w_packed && y && aligned16(w_packed), SEGK_EINVAL, "conv_fwd: null or misaligned weights/output");
    SEGK_REQUIRE(Cout > 0 && Cout % 8 == 0 && (y1 || ldy >= Cout), SEGK_ESHAPE, "conv_fwd: Cout %lld must be a multiple of 8 and <= ldy", (long long)Cout);
    SEGK_REQUIRE(aligned16(y) && (ldy * esize(dtype)) % 16 == 0, SEGK_EALIGN, "conv_fwd: output pointer/ld misaligned");
    SEGK_REQUIRE(!bias || aligned16(bias), SEGK_EALIGN, "conv_fwd: bias must be 16-byte aligned");
    SEGK_REQUIRE(!mask || (aligned16(mask) && (ld_mask * esize(dtype)) % 16 == 0), SEGK_EALIGN, "conv_fwd: mask misaligned");
    SEGK_REQUIRE(!mask || (mask_channels > 0 && mask_channels <= Cout && mask_channels % 8 == 0 && ld_mask >= mask_channels), SEGK_ESHAPE,
                 "conv_fwd: mask_channels must be a multiple of 8 in (0, Cout] and fit ld_mask");
    SEGK_REQUIRE(!stat_partials || (aligned16(stat_partials) && !y1 && !mask && !(flags & 6)), SEGK_EINVAL,
                 "conv_fwd: output statistics need a single destination, no mask, no accumulation and a 16-byte aligned buffer");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    ConvArgs a;
    a.stat_part = stat_partials;
    a.stat_rpt = (flags & 8) ? 4 : 0;
    a.nb_z = (const char*)nb.z; a.nb_ldz = nb.ldz; a.nb_stats = nb.stats; a.nb_gamma = nb.gamma; a.nb_beta = nb.beta; a.nb_part = nb.partials;
    a.x0 = (const char*)x0; a.x1 = (const char*)x1; a.ld0 = ld0; a.ld1 = ld1;
    a.C0 = (int)C0; a.C1 = (int)C1;
    a.nch0 = (int)(round_up(C0, kr) / kr); a.nch = a.nch0 + (int)(round_up(C1, kr) / kr);
    a.N = (int)N; a.H = (int)H; a.W = (int)W;
    a.vin = a.vout = identity_view((int)H, (int)W);
    a.tiles_x = (int)((W + kTW - 1) / kTW); a.tiles_y = (int)((H + 7) / 8);
    a.w = (const char*)w_packed; a.Cout = (int)Cout; a.cout_pad = (int)round_up(Cout, 32);
    a.bias = bias; a.y = (char*)y; a.ldy = ldy; a.mask = (const char*)mask; a.ld_mask = ld_mask; a.mask_c = (int)mask_channels;
    a.relu = flags & 1; a.accumulate = (flags >> 1) & 1;
    if (y1) { a.y1 = (char*)y1; a.ldy1 = ldy1; a.split = (int)split; a.accumulate1 = (flags >> 2) & 1; }
    static const int remap = getenv("SEGK_XCD_REMAP") ? atoi(getenv("SEGK_XCD_REMAP")) : 1;
    a.xcd_remap = remap;
    const int bn = pick_bn(a.cout_pad);
    SEGK_REQUIRE((int64_t)a.N * a.tiles_x * a.tiles_y * (a.cout_pad / bn) < (1ll << 31), SEGK_ESHAPE, "conv_fwd: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (!(dtype == SEGK_BF16 && taps == 9 && launch_direct(a, bn, st))) launch_fwd_any(dtype, taps, a, bn, st);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_conv_fwd(int dtype, int taps, const void* x0, int64_t ld0, int64_t C0, const void* x1, int64_t ld1, int64_t C1,
                             int64_t N, int64_t H, int64_t W, const void* w_packed, int64_t Cout, const float* bias, void* y,
                             int64_t ldy, void* y1, int64_t ldy1, int64_t split, const void* mask, int64_t ld_mask, int64_t mask_channels,
                             float* stat_partials, int flags, int dev, void* stream) {
    return conv_fwd_impl(dtype, taps, x0, ld0, C0, x1, ld1, C1, N, H, W, w_packed, Cout, bias, y, ldy, y1, ldy1, split, mask, ld_mask,
                         mask_channels, stat_partials, flags, NormBwdSums{nullptr, 0, nullptr, nullptr, nullptr, nullptr}, dev, stream);
}

extern "C" int segk_conv_dgrad_norm(int dtype, int taps, const void* dy, int64_t ld_dy, int64_t C, int64_t N, int64_t H, int64_t W,
                                    const void* w_packed, int64_t Cout, void* g, int64_t ldg, const void* z, int64_t ldz,
                                    const float* stats, const float* gamma, const float* beta, float* partials, int dev, void* stream) {
    SEGK_REQUIRE(dtype == SEGK_F32 || dtype == SEGK_BF16, SEGK_EINVAL, "conv_dgrad_norm: unknown dtype %d", dtype);
    SEGK_REQUIRE(z && stats && partials && aligned16(z) && aligned16(stats) && aligned16(partials), SEGK_EINVAL,
                 "conv_dgrad_norm: z, stats and partials must be non-null and 16-byte aligned");
    SEGK_REQUIRE(ldz >= Cout && (ldz * esize(dtype)) % 16 == 0, SEGK_EALIGN, "conv_dgrad_norm: z pixel stride misaligned or < Cout");
    return conv_fwd_impl(dtype, taps, dy, ld_dy, C, nullptr, 0, 0, N, H, W, w_packed, Cout, nullptr, g, ldg, nullptr, 0, 0, nullptr, 0, 0,
                         nullptr, 0, NormBwdSums{z, ldz, stats, gamma, beta, partials}, dev, stream);
}

// rows of per-wave output statistics segk_conv_fwd writes: (image, tile, pixel-wave); the pixel-wave count follows the
// kernel variant launch_fwd_t picks for this channel-block width
extern "C" int64_t segk_conv_stat_rows(int64_t N, int64_t H, int64_t W, int64_t Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    const int bn = pick_bn(round_up(Cout, 32));
    const int wpx = (bn == 128 || bn == 64) ? 2 : 4;
    return N * ((H + 7) / 8) * ((W + kTW - 1) / kTW) * wpx;
}

// rows of the statistics buffer of a segk_conv_fwd call that sets flag bit 3: four per (image, tile) whatever the kernel
extern "C" int64_t segk_conv_stat_rows_fine(int64_t N, int64_t H, int64_t W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return N * ((H + 7) / 8) * ((W + kTW - 1) / kTW) * 4;
}

extern "C" size_t segk_conv_wgrad_workspace(int dtype, int taps, int64_t C0, int64_t C1, int64_t N, int64_t H, int64_t W,
                                            int64_t Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || C0 <= 0 || C1 < 0) return 0;
    const int kr = krow(dtype);
    const int64_t nch = round_up(C0, kr) / kr + round_up(C1, kr) / kr, ncb = (Cout + kr - 1) / kr;
    const int64_t ntiles = N * ((H + 7) / 8) * ((W + kTW - 1) / kTW);
    const int s = wgrad_splits(ntiles, ncb, nch);
    return align_up((size_t)s * ncb * kr * taps * nch * kr * sizeof(float), 256);
}

extern "C" int segk_conv_wgrad(int dtype, int taps, const void* x0, int64_t ld0, int64_t C0, const void* x1, int64_t ld1,
                               int64_t C1, int64_t N, int64_t H, int64_t W, const void* dy, int64_t ldy, int64_t Cout,
                               int64_t cin_real, float* dw, int flags, void* workspace, size_t workspace_bytes, int dev,
                               void* stream) {
    if (int rc = check_segments("conv_wgrad", dtype, x0, ld0, C0, x1, ld1, C1)) return rc;
    SEGK_REQUIRE(taps == 1 || taps == 9, SEGK_ESHAPE, "conv_wgrad: taps must be 1 or 9");
    SEGK_REQUIRE(N > 0 && H > 0 && W > 0 && N * H * W < (1ll << 31), SEGK_ESHAPE, "conv_wgrad: bad N/H/W");
    SEGK_REQUIRE(dy && dw && workspace, SEGK_EINVAL, "conv_wgrad: null pointer");
    SEGK_REQUIRE(Cout > 0 && Cout % 8 == 0 && ldy >= Cout && cin_real > 0 && cin_real <= C0 + C1 && (C1 == 0 || cin_real == C0 + C1),
                 SEGK_ESHAPE, "conv_wgrad: bad channel counts (Cout=%lld cin_real=%lld)", (long long)Cout, (long long)cin_real);
    SEGK_REQUIRE(aligned16(dy) && (ldy * esize(dtype)) % 16 == 0, SEGK_EALIGN, "conv_wgrad: dy pointer/ld misaligned");
    SEGK_REQUIRE(workspace_bytes >= segk_conv_wgrad_workspace(dtype, taps, C0, C1, N, H, W, Cout), SEGK_EWORKSPACE, "conv_wgrad: workspace too small");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    WgradArgs a;
    a.x0 = (const char*)x0; a.x1 = (const char*)x1; a.ld0 = ld0; a.ld1 = ld1;
    a.C0 = (int)C0; a.C1 = (int)C1;
    a.nch0 = (int)(round_up(C0, kr) / kr); a.nch = a.nch0 + (int)(round_up(C1, kr) / kr);
    a.N = (int)N; a.H = (int)H; a.W = (int)W;
    a.tiles_x = (int)((W + kTW - 1) / kTW); a.tiles_y = (int)((H + 7) / 8);
    a.dy = (const char*)dy; a.ldy = ldy; a.Cout = (int)Cout; a.ncb = (int)((Cout + kr - 1) / kr);
    a.vdy = identity_view((int)H, (int)W);
    const int64_t ntiles = (int64_t)a.N * a.tiles_x * a.tiles_y;
    a.splits = wgrad_splits(ntiles, a.ncb, a.nch);
    a.tiles_per_split = (int)((ntiles + a.splits - 1) / a.splits);
    a.slab = (float*)workspace;
    static const bool no_remap = getenv("SEGK_WGRAD_NO_XCD_REMAP") != nullptr;   // experiment switch: plain block order
    a.xcd_remap = no_remap ? 0 : 1;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((int64_t)a.ncb * a.nch * a.splits);
    launch_wgrad_any(dtype, taps, a, grid, st, Cout, C0);
    SEGK_LAUNCH_CHECK();
    const int seg0 = C1 ? (int)C0 : (int)cin_real;
    launch_wgrad_reduce(st, (const float*)workspace, a.splits, (int64_t)a.ncb * kr, taps, a.nch * kr, (int)Cout, (int)cin_real, seg0,
                        a.nch0 * kr, dw, (flags >> 1) & 1);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose2d(4, stride 2, padding 1): host side
// ------------------------------------------------------------------------------------------------
namespace segk {
static size_t convt_phase_bytes(int dtype, int64_t Cin, int64_t Cout, bool fwd) {
    const int kr = krow(dtype);
    return fwd ? (size_t)(round_up(Cin, kr) / kr * 4 * round_up(Cout, 32) * kRowBytes)
               : (size_t)(round_up(Cout, kr) / kr * 4 * round_up(Cin, 32) * kRowBytes);
}
static int check_convt(const char* who, int dtype, const void* p, int64_t ld, int64_t C) {
    if (dtype != SEGK_F32 && dtype != SEGK_BF16) return fail(SEGK_EINVAL, "%s: unknown dtype %d", who, dtype);
    if (!p || C <= 0 || C % 8 || ld < C) return fail(SEGK_ESHAPE, "%s: channels %lld must be a positive multiple of 8 and <= ld", who, (long long)C);
    if (!aligned16(p) || (ld * esize(dtype)) % 16) return fail(SEGK_EALIGN, "%s: pointer/ld not 16-byte aligned", who);
    return SEGK_OK;
}
}  // namespace segk

extern "C" size_t segk_convt_packed_bytes(int dtype, int64_t Cin, int64_t Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    const size_t f = 4 * convt_phase_bytes(dtype, Cin, Cout, true), b = 4 * convt_phase_bytes(dtype, Cin, Cout, false);
    return f > b ? f : b;
}

extern "C" int segk_convt_pack(const float* w, int64_t Cin, int64_t Cout, int dtype, void* wf, void* wb, int dev, void* stream) {
    SEGK_REQUIRE(w && (wf || wb) && Cin > 0 && Cout > 0, SEGK_EINVAL, "convt_pack: bad arguments");
    SEGK_REQUIRE(dtype == SEGK_F32 || dtype == SEGK_BF16, SEGK_EINVAL, "convt_pack: unknown dtype");
    SEGK_REQUIRE((!wf || aligned16(wf)) && (!wb || aligned16(wb)), SEGK_EALIGN, "convt_pack: packed buffers must be 16-byte aligned");
    SEGK_USE_DEVICE(dev);
    const int64_t slots = (int64_t)((wf ? 4 * convt_phase_bytes(dtype, Cin, Cout, true) : 0) + (wb ? 4 * convt_phase_bytes(dtype, Cin, Cout, false) : 0)) / 16;
    int64_t grid = (slots + kConvThreads - 1) / kConvThreads;
    if (grid > 4096) grid = 4096;
    SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((convt_pack_kernel<T>), dim3((unsigned)grid), dim3(kConvThreads), 0, (hipStream_t)stream, w,
                                                     (int)Cin, (int)Cout, (char*)wf, (char*)wb));
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" int segk_convt_fwd(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t H, int64_t W, const void* wf,
                              int64_t Cout, const float* bias, void* y, int64_t ldy, int flags, int dev, void* stream) {
    if (int rc = check_convt("convt_fwd(x)", dtype, x, ldx, Cin)) return rc;
    if (int rc = check_convt("convt_fwd(y)", dtype, y, ldy, Cout)) return rc;
    SEGK_REQUIRE(wf && aligned16(wf) && N > 0 && H > 0 && W > 0 && N * H * W * 4 < (1ll << 28), SEGK_ESHAPE, "convt_fwd: bad arguments");
    SEGK_REQUIRE(!bias || aligned16(bias), SEGK_EALIGN, "convt_fwd: bias must be 16-byte aligned");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    ConvArgs a;
    a.x0 = (const char*)x; a.x1 = nullptr; a.ld0 = ldx; a.ld1 = 0; a.C0 = (int)Cin; a.C1 = 0;
    a.nch0 = a.nch = (int)(round_up(Cin, kr) / kr);
    a.N = (int)N; a.H = (int)H; a.W = (int)W;
    a.vin = identity_view((int)H, (int)W);
    a.tiles_x = (int)((W + kTW - 1) / kTW); a.tiles_y = (int)((H + 7) / 8);
    a.Cout = (int)Cout; a.cout_pad = (int)round_up(Cout, 32);
    a.bias = bias; a.y = (char*)y; a.ldy = ldy; a.mask = nullptr; a.ld_mask = 0; a.mask_c = 0; a.relu = flags & 1; a.accumulate = 0;
    const int bn = pick_bn(a.cout_pad);
    for (int ph = 0; ph < 4; ++ph) {
        const int pa = ph >> 1, pb = ph & 1;
        a.w = (const char*)wf + (size_t)ph * convt_phase_bytes(dtype, Cin, Cout, true);
        a.vout = View{(int)(2 * H), (int)(2 * W), 2, 2, pa, pb};
        for (int t = 0; t < 4; ++t) { a.tky[t] = ct_off(pa, t >> 1) + 1; a.tkx[t] = ct_off(pb, t & 1) + 1; }
        launch_fwd_any(dtype, 4, a, bn, (hipStream_t)stream);
        SEGK_LAUNCH_CHECK();
    }
    return SEGK_OK;
}

extern "C" int segk_convt_dgrad(int dtype, const void* dy, int64_t ld_dy, int64_t Cout, int64_t N, int64_t H, int64_t W, const void* wb,
                                int64_t Cin, void* dx, int64_t ld_dx, const void* mask, int64_t ld_mask, int flags, int dev, void* stream) {
    if (int rc = check_convt("convt_dgrad(dy)", dtype, dy, ld_dy, Cout)) return rc;
    if (int rc = check_convt("convt_dgrad(dx)", dtype, dx, ld_dx, Cin)) return rc;
    SEGK_REQUIRE(wb && aligned16(wb) && N > 0 && H > 0 && W > 0 && N * H * W * 4 < (1ll << 28), SEGK_ESHAPE, "convt_dgrad: bad arguments");
    SEGK_REQUIRE(!mask || (aligned16(mask) && ld_mask >= Cin && (ld_mask * esize(dtype)) % 16 == 0), SEGK_EALIGN, "convt_dgrad: mask misaligned");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    ConvArgs a;
    a.x0 = (const char*)dy; a.x1 = nullptr; a.ld0 = ld_dy; a.ld1 = 0; a.C0 = (int)Cout; a.C1 = 0;
    a.nch0 = a.nch = (int)(round_up(Cout, kr) / kr);
    a.N = (int)N; a.H = (int)H; a.W = (int)W;                 // the convolution grid is the low-resolution input grid
    a.vout = identity_view((int)H, (int)W);
    a.tiles_x = (int)((W + kTW - 1) / kTW); a.tiles_y = (int)((H + 7) / 8);
    a.Cout = (int)Cin; a.cout_pad = (int)round_up(Cin, 32);
    a.bias = nullptr; a.y = (char*)dx; a.ldy = ld_dx; a.mask = (const char*)mask; a.ld_mask = ld_mask; a.mask_c = (int)Cin; a.relu = 0;
    const int bn = pick_bn(a.cout_pad);
    for (int ph = 0; ph < 4; ++ph) {
        const int pa = ph >> 1, pb = ph & 1;
        a.w = (const char*)wb + (size_t)ph * convt_phase_bytes(dtype, Cin, Cout, false);
        a.vin = View{(int)(2 * H), (int)(2 * W), 2, 2, pa, pb};   // phase (a,b) of the output gradient
        for (int t = 0; t < 4; ++t) { a.tky[t] = -ct_off(pa, t >> 1) + 1; a.tkx[t] = -ct_off(pb, t & 1) + 1; }
        a.accumulate = (ph > 0 || ((flags >> 1) & 1)) ? 1 : 0;
        launch_fwd_any(dtype, 4, a, bn, (hipStream_t)stream);
        SEGK_LAUNCH_CHECK();
    }
    return SEGK_OK;
}

extern "C" size_t segk_convt_wgrad_workspace(int dtype, int64_t Cin, int64_t Cout, int64_t N, int64_t H, int64_t W) {
    return segk_conv_wgrad_workspace(dtype, 4, Cin, 0, N, H, W, Cout);
}

extern "C" int segk_convt_wgrad(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t H, int64_t W, const void* dy,
                                int64_t ld_dy, int64_t Cout, float* dw, int flags, void* workspace, size_t workspace_bytes, int dev,
                                void* stream) {
    if (int rc = check_convt("convt_wgrad(x)", dtype, x, ldx, Cin)) return rc;
    if (int rc = check_convt("convt_wgrad(dy)", dtype, dy, ld_dy, Cout)) return rc;
    SEGK_REQUIRE(dw && workspace && N > 0 && H > 0 && W > 0 && N * H * W * 4 < (1ll << 28), SEGK_ESHAPE, "convt_wgrad: bad arguments");
    SEGK_REQUIRE(workspace_bytes >= segk_convt_wgrad_workspace(dtype, Cin, Cout, N, H, W), SEGK_EWORKSPACE, "convt_wgrad: workspace too small");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a;
    a.x0 = (const char*)x; a.x1 = nullptr; a.ld0 = ldx; a.ld1 = 0; a.C0 = (int)Cin; a.C1 = 0;
    a.nch0 = a.nch = (int)(round_up(Cin, kr) / kr);
    a.N = (int)N; a.H = (int)H; a.W = (int)W;
    a.tiles_x = (int)((W + kTW - 1) / kTW); a.tiles_y = (int)((H + 7) / 8);
    a.dy = (const char*)dy; a.ldy = ld_dy; a.Cout = (int)Cout; a.ncb = (int)((Cout + kr - 1) / kr);
    const int64_t ntiles = (int64_t)a.N * a.tiles_x * a.tiles_y;
    a.splits = wgrad_splits(ntiles, a.ncb, a.nch);
    a.tiles_per_split = (int)((ntiles + a.splits - 1) / a.splits);
    a.slab = (float*)workspace;
    const unsigned grid = (unsigned)((int64_t)a.ncb * a.nch * a.splits);
    for (int ph = 0; ph < 4; ++ph) {
        const int pa = ph >> 1, pb = ph & 1;
        a.vdy = View{(int)(2 * H), (int)(2 * W), 2, 2, pa, pb};
        int kidx[4];
        for (int t = 0; t < 4; ++t) {
            a.tky[t] = ct_off(pa, t >> 1) + 1; a.tkx[t] = ct_off(pb, t & 1) + 1;
            kidx[t] = ct_k(pa, t >> 1) * 4 + ct_k(pb, t & 1);
        }
        launch_wgrad_any(dtype, 4, a, grid, st, Cout, Cin);
        SEGK_LAUNCH_CHECK();
        launch_wgrad_reduce(st, (const float*)workspace, a.splits, (int64_t)a.ncb * kr, 4, a.nch * kr, (int)Cout, (int)Cin, (int)Cin,
                            a.nch0 * kr, dw, (flags >> 1) & 1, (int)Cout, kidx[0], kidx[1], kidx[2], kidx[3]);
        SEGK_LAUNCH_CHECK();
    }
    return SEGK_OK;
}

// ------------------------------------------------------------------------------------------------
// Conv3d(kernel 3, stride 1, padding 1) on volumes stored slice by slice ([N*D, H, W, C]): ONE launch of the 3x3
// kernel whose K dimension runs over the three depth taps -- K-rows [kdi*nch, (kdi+1)*nch) read the channels of the
// neighbouring slice d + kdi - 1 (zero beyond the volume's faces).  Packed images: three 2-D images back to back, one
// per depth tap, so that the kernel's (K-row, tap) step index walks straight through them.
// ------------------------------------------------------------------------------------------------
namespace segk {
// bytes of ONE depth tap's forward (fwd) or data-gradient image
static size_t conv3d_image_bytes(int dtype, int64_t Cout, int64_t Cin, bool fwd) {
    const int kr = krow(dtype);
    return fwd ? (size_t)(round_up(Cin, kr) / kr * 9 * round_up(Cout, 32) * kRowBytes)
               : (size_t)(round_up(Cout, kr) / kr * 9 * round_up(Cin, 32) * kRowBytes);
}
}

extern "C" size_t segk_conv3d_packed_bytes(int dtype, int64_t Cout, int64_t Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    const size_t f = conv3d_image_bytes(dtype, Cout, Cin, true), b = conv3d_image_bytes(dtype, Cout, Cin, false);
    return 3 * (f > b ? f : b);
}

extern "C" int segk_conv3d_pack(const float* w, int64_t Cout, int64_t Cin, int dtype, void* wf, void* wb, int dev, void* stream) {
    SEGK_REQUIRE(w && (wf || wb) && Cin > 0 && Cout > 0, SEGK_EINVAL, "conv3d_pack: bad arguments");
    SEGK_REQUIRE(dtype == SEGK_F32 || dtype == SEGK_BF16, SEGK_EINVAL, "conv3d_pack: unknown dtype");
    SEGK_REQUIRE((!wf || aligned16(wf)) && (!wb || aligned16(wb)), SEGK_EALIGN, "conv3d_pack: packed buffers must be 16-byte aligned");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    const size_t per_f = conv3d_image_bytes(dtype, Cout, Cin, true), per_b = conv3d_image_bytes(dtype, Cout, Cin, false);
    const int64_t slots = ((wf ? round_up(Cin, kr) / kr * round_up(Cout, 32) : 0) + (wb ? round_up(Cout, kr) / kr * round_up(Cin, 32) : 0)) * 9 * 8;
    int64_t grid = (slots + kConvThreads - 1) / kConvThreads;
    if (grid > 4096) grid = 4096;
    for (int kd = 0; kd < 3; ++kd) {
        // forward image kd multiplies slice d + kd - 1; the data gradient reads dy at d - (kd - 1): its image sits at 2 - kd
        char* f = wf ? (char*)wf + kd * per_f : nullptr;
        char* b = wb ? (char*)wb + (2 - kd) * per_b : nullptr;
        SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((conv_pack_kernel<T>), dim3((unsigned)grid), dim3(kConvThreads), 0, (hipStream_t)stream,
                                                         w + kd * 9, (int)Cout, (int)Cin, 9, (int)Cin, f, b, Cin * 27, (int64_t)27));
        SEGK_LAUNCH_CHECK();
    }
    return SEGK_OK;
}

extern "C" int segk_conv3d_fwd(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W,
                               const void* w_packed, int64_t Cout, void* y, int64_t ldy, int flags, int dev, void* stream) {
    if (int rc = check_convt("conv3d_fwd(x)", dtype, x, ldx, Cin)) return rc;
    if (int rc = check_convt("conv3d_fwd(y)", dtype, y, ldy, Cout)) return rc;
    SEGK_REQUIRE(w_packed && aligned16(w_packed) && N > 0 && D > 0 && H > 0 && W > 0 && N * D * H * W < (1ll << 28), SEGK_ESHAPE,
                 "conv3d_fwd: bad arguments");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    ConvArgs a;
    a.x0 = (const char*)x; a.x1 = nullptr; a.ld0 = ldx; a.ld1 = 0; a.C0 = (int)Cin; a.C1 = 0;
    a.nch0 = (int)(round_up(Cin, kr) / kr);
    a.nch = 3 * a.nch0;                                   // the three depth taps are consecutive stretches of K
    a.N = (int)(N * D); a.H = (int)H; a.W = (int)W; a.depth = (int)D;
    a.slice_bytes = H * W * ldx * (int64_t)esize(dtype);
    a.vin = a.vout = identity_view((int)H, (int)W);
    a.tiles_x = (int)((W + kTW - 1) / kTW); a.tiles_y = (int)((H + 7) / 8);
    a.w = (const char*)w_packed;
    a.Cout = (int)Cout; a.cout_pad = (int)round_up(Cout, 32);
    a.bias = nullptr; a.y = (char*)y; a.ldy = ldy; a.mask = nullptr; a.ld_mask = 0; a.mask_c = 0; a.relu = 0;
    a.accumulate = (flags >> 1) & 1;
    const int bn = pick_bn(a.cout_pad);
    SEGK_REQUIRE((int64_t)a.N * a.tiles_x * a.tiles_y * (a.cout_pad / bn) < (1ll << 31), SEGK_ESHAPE, "conv3d_fwd: grid too large");
    launch_fwd_any(dtype, 9, a, bn, (hipStream_t)stream);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

namespace segk {
static int conv3d_wgrad_splits(int dtype, int64_t Cin, int64_t ND, int64_t H, int64_t W, int64_t Cout) {
    const int kr = krow(dtype);
    const int64_t ntiles = ND * ((H + 7) / 8) * ((W + kTW - 1) / kTW);
    // three times the workgroups of the 2-D case for the same resident round: a third of the pixel splits
    return wgrad_splits(ntiles, 3 * ((Cout + kr - 1) / kr), round_up(Cin, kr) / kr);
}
}

extern "C" size_t segk_conv3d_wgrad_workspace(int dtype, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W, int64_t Cout) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin <= 0) return 0;
    const int kr = krow(dtype);
    const int64_t nch = round_up(Cin, kr) / kr, ncb = (Cout + kr - 1) / kr;
    const int s = conv3d_wgrad_splits(dtype, Cin, N * D, H, W, Cout);
    return align_up((size_t)3 * s * ncb * kr * 9 * nch * kr * sizeof(float), 256);
}

extern "C" int segk_conv3d_wgrad(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W,
                                 const void* dy, int64_t ld_dy, int64_t Cout, float* dw, int flags, void* workspace,
                                 size_t workspace_bytes, int dev, void* stream) {
    if (int rc = check_convt("conv3d_wgrad(x)", dtype, x, ldx, Cin)) return rc;
    if (int rc = check_convt("conv3d_wgrad(dy)", dtype, dy, ld_dy, Cout)) return rc;
    SEGK_REQUIRE(dw && workspace && N > 0 && D > 0 && H > 0 && W > 0 && N * D * H * W < (1ll << 28), SEGK_ESHAPE, "conv3d_wgrad: bad arguments");
    SEGK_REQUIRE(workspace_bytes >= segk_conv3d_wgrad_workspace(dtype, Cin, N, D, H, W, Cout), SEGK_EWORKSPACE, "conv3d_wgrad: workspace too small");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a;
    a.x0 = (const char*)x; a.x1 = nullptr; a.ld0 = ldx; a.ld1 = 0; a.C0 = (int)Cin; a.C1 = 0;
    a.nch0 = a.nch = (int)(round_up(Cin, kr) / kr);
    a.N = (int)(N * D); a.H = (int)H; a.W = (int)W; a.depth = (int)D;
    a.tiles_x = (int)((W + kTW - 1) / kTW); a.tiles_y = (int)((H + 7) / 8);
    a.dy = (const char*)dy; a.ldy = ld_dy; a.Cout = (int)Cout; a.ncb = (int)((Cout + kr - 1) / kr);
    a.vdy = identity_view((int)H, (int)W);
    const int64_t ntiles = (int64_t)a.N * a.tiles_x * a.tiles_y;
    a.splits = conv3d_wgrad_splits(dtype, Cin, N * D, H, W, Cout);
    a.tiles_per_split = (int)((ntiles + a.splits - 1) / a.splits);
    a.slab = (float*)workspace;
    // one launch for the three depth taps (slab [3][splits]...), one reduction that writes dw[Cout][Cin][3][3][3]
    const unsigned grid = (unsigned)((int64_t)3 * a.ncb * a.nch * a.splits);
    launch_wgrad_any(dtype, 9, a, grid, st, Cout, Cin);
    SEGK_LAUNCH_CHECK();
    launch_wgrad_reduce(st, (const float*)workspace, a.splits, (int64_t)a.ncb * kr, 9, a.nch * kr, (int)Cout, (int)Cin, (int)Cin,
                        a.nch0 * kr, dw, (flags >> 1) & 1, 0, 0, 0, 0, 0, Cin * 27, 27, 3);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// ------------------------------------------------------------------------------------------------
// Conv3d with kernel (kd, 1, 1), stride 1, padding (kd/2, 0, 0) over slice-major volumes: the depth half of a
// factorised strided stem convolution -- the (kh, kw) window is gathered per slice into patch rows once
// (segk_im2col3d with a depth extent of 1), and this convolution sums the kd neighbouring slices' patch rows.
// One launch of the 1x1 kernel whose K loop runs over the depth taps.  w: fp32 [Cout][Cin][kd].
// ------------------------------------------------------------------------------------------------
namespace segk {
static size_t convd_image_bytes(int dtype, int64_t Cout, int64_t Cin, bool fwd) {
    const int kr = krow(dtype);
    return fwd ? (size_t)(round_up(Cin, kr) / kr * round_up(Cout, 32) * kRowBytes)
               : (size_t)(round_up(Cout, kr) / kr * round_up(Cin, 32) * kRowBytes);
}
static int check_convd(const char* who, int64_t kd) {
    if (kd < 1 || kd > 7 || kd % 2 == 0) return fail(SEGK_ESHAPE, "%s: depth extent %lld must be 1, 3, 5 or 7", who, (long long)kd);
    return SEGK_OK;
}
static int convd_wgrad_splits(int dtype, int64_t Cin, int64_t ND, int64_t H, int64_t W, int64_t Cout, int64_t kd) {
    const int kr = krow(dtype);
    const int64_t ntiles = ND * ((H + 7) / 8) * ((W + kTW - 1) / kTW);
    return wgrad_splits(ntiles, kd * ((Cout + kr - 1) / kr), round_up(Cin, kr) / kr);
}
}

extern "C" size_t segk_convd_packed_bytes(int dtype, int64_t Cout, int64_t Cin, int64_t kd) {
    if (Cout <= 0 || Cin <= 0 || kd < 1) return 0;
    const size_t f = convd_image_bytes(dtype, Cout, Cin, true), b = convd_image_bytes(dtype, Cout, Cin, false);
    return (size_t)kd * (f > b ? f : b);
}

extern "C" int segk_convd_pack(const float* w, int64_t Cout, int64_t Cin, int64_t kd, int dtype, void* wf, void* wb, int dev, void* stream) {
    SEGK_REQUIRE(w && (wf || wb) && Cin > 0 && Cout > 0, SEGK_EINVAL, "convd_pack: bad arguments");
    if (int rc = check_convd("convd_pack", kd)) return rc;
    SEGK_REQUIRE(dtype == SEGK_F32 || dtype == SEGK_BF16, SEGK_EINVAL, "convd_pack: unknown dtype");
    SEGK_REQUIRE((!wf || aligned16(wf)) && (!wb || aligned16(wb)), SEGK_EALIGN, "convd_pack: packed buffers must be 16-byte aligned");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    const size_t per_f = convd_image_bytes(dtype, Cout, Cin, true), per_b = convd_image_bytes(dtype, Cout, Cin, false);
    const int64_t slots = ((wf ? round_up(Cin, kr) / kr * round_up(Cout, 32) : 0) + (wb ? round_up(Cout, kr) / kr * round_up(Cin, 32) : 0)) * 8;
    int64_t grid = (slots + kConvThreads - 1) / kConvThreads;
    if (grid > 4096) grid = 4096;
    for (int a = 0; a < kd; ++a) {
        // forward image a multiplies slice d + a - kd/2; the data gradient reads dy at d - (a - kd/2): its image sits at kd-1-a
        char* f = wf ? (char*)wf + a * per_f : nullptr;
        char* b = wb ? (char*)wb + (kd - 1 - a) * per_b : nullptr;
        SEGK_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((conv_pack_kernel<T>), dim3((unsigned)grid), dim3(kConvThreads), 0, (hipStream_t)stream,
                                                         w + a, (int)Cout, (int)Cin, 1, (int)Cin, f, b, Cin * kd, kd));
        SEGK_LAUNCH_CHECK();
    }
    return SEGK_OK;
}

extern "C" int segk_convd_fwd(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W, int64_t kd,
                              const void* w_packed, int64_t Cout, void* y, int64_t ldy, float* stat_partials, int flags, int dev,
                              void* stream) {
    if (int rc = check_convt("convd_fwd(x)", dtype, x, ldx, Cin)) return rc;
    if (int rc = check_convt("convd_fwd(y)", dtype, y, ldy, Cout)) return rc;
    if (int rc = check_convd("convd_fwd", kd)) return rc;
    SEGK_REQUIRE(w_packed && aligned16(w_packed) && N > 0 && D > 0 && H > 0 && W > 0 && N * D * H * W < (1ll << 28), SEGK_ESHAPE,
                 "convd_fwd: bad arguments");
    SEGK_REQUIRE(!stat_partials || (aligned16(stat_partials) && !(flags & 2)), SEGK_EINVAL, "convd_fwd: statistics need a non-accumulating call");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    ConvArgs a;
    a.x0 = (const char*)x; a.x1 = nullptr; a.ld0 = ldx; a.ld1 = 0; a.C0 = (int)Cin; a.C1 = 0;
    a.nch0 = (int)(round_up(Cin, kr) / kr);
    a.nkd = (int)kd;
    a.nch = a.nkd * a.nch0;
    a.N = (int)(N * D); a.H = (int)H; a.W = (int)W; a.depth = (int)D;
    a.slice_bytes = H * W * ldx * (int64_t)esize(dtype);
    a.vin = a.vout = identity_view((int)H, (int)W);
    a.tiles_x = (int)((W + kTW - 1) / kTW); a.tiles_y = (int)((H + 7) / 8);
    a.w = (const char*)w_packed;
    a.Cout = (int)Cout; a.cout_pad = (int)round_up(Cout, 32);
    a.bias = nullptr; a.y = (char*)y; a.ldy = ldy; a.mask = nullptr; a.ld_mask = 0; a.mask_c = 0; a.relu = 0;
    a.accumulate = (flags >> 1) & 1;
    a.stat_part = stat_partials;
    const int bn = pick_bn(a.cout_pad);
    SEGK_REQUIRE((int64_t)a.N * a.tiles_x * a.tiles_y * (a.cout_pad / bn) < (1ll << 31), SEGK_ESHAPE, "convd_fwd: grid too large");
    launch_fwd_any(dtype, 1, a, bn, (hipStream_t)stream);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" size_t segk_convd_wgrad_workspace(int dtype, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W, int64_t Cout, int64_t kd) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin <= 0 || kd < 1) return 0;
    const int kr = krow(dtype);
    const int64_t nch = round_up(Cin, kr) / kr, ncb = (Cout + kr - 1) / kr;
    const int s = convd_wgrad_splits(dtype, Cin, N * D, H, W, Cout, kd);
    return align_up((size_t)kd * s * ncb * kr * nch * kr * sizeof(float), 256);
}

/* dw: fp32 [Cout][cin_real][kd] (cin_real <= Cin: trailing padding channels of the patch rows are left out) */
extern "C" int segk_convd_wgrad(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W, int64_t kd,
                                const void* dy, int64_t ld_dy, int64_t Cout, int64_t cin_real, float* dw, int flags, void* workspace,
                                size_t workspace_bytes, int dev, void* stream) {
    if (int rc = check_convt("convd_wgrad(x)", dtype, x, ldx, Cin)) return rc;
    if (int rc = check_convt("convd_wgrad(dy)", dtype, dy, ld_dy, Cout)) return rc;
    if (int rc = check_convd("convd_wgrad", kd)) return rc;
    SEGK_REQUIRE(dw && workspace && N > 0 && D > 0 && H > 0 && W > 0 && N * D * H * W < (1ll << 28) && cin_real > 0 && cin_real <= Cin,
                 SEGK_ESHAPE, "convd_wgrad: bad arguments");
    SEGK_REQUIRE(workspace_bytes >= segk_convd_wgrad_workspace(dtype, Cin, N, D, H, W, Cout, kd), SEGK_EWORKSPACE, "convd_wgrad: workspace too small");
    SEGK_USE_DEVICE(dev);
    const int kr = krow(dtype);
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a;
    a.x0 = (const char*)x; a.x1 = nullptr; a.ld0 = ldx; a.ld1 = 0; a.C0 = (int)Cin; a.C1 = 0;
    a.nch0 = a.nch = (int)(round_up(Cin, kr) / kr);
    a.N = (int)(N * D); a.H = (int)H; a.W = (int)W; a.depth = (int)D; a.nkd = (int)kd;
    a.tiles_x = (int)((W + kTW - 1) / kTW); a.tiles_y = (int)((H + 7) / 8);
    a.dy = (const char*)dy; a.ldy = ld_dy; a.Cout = (int)Cout; a.ncb = (int)((Cout + kr - 1) / kr);
    a.vdy = identity_view((int)H, (int)W);
    const int64_t ntiles = (int64_t)a.N * a.tiles_x * a.tiles_y;
    a.splits = convd_wgrad_splits(dtype, Cin, N * D, H, W, Cout, kd);
    a.tiles_per_split = (int)((ntiles + a.splits - 1) / a.splits);
    a.slab = (float*)workspace;
    const unsigned grid = (unsigned)((int64_t)kd * a.ncb * a.nch * a.splits);
    launch_wgrad_any(dtype, 1, a, grid, st, Cout, Cin);
    SEGK_LAUNCH_CHECK();
    launch_wgrad_reduce(st, (const float*)workspace, a.splits, (int64_t)a.ncb * kr, 1, a.nch * kr, (int)Cout, (int)cin_real, (int)cin_real,
                        a.nch0 * kr, dw, (flags >> 1) & 1, 0, 0, 0, 0, 0, cin_real * kd, kd, (int)kd);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

#ifdef SEGK_CLOCK_PROBE
// diagnostic build only: copy the stamps of the last convolution launch to the host ([1024][4] x 8 bytes)
extern "C" int segk_debug_clock_probe(unsigned long long* out_host) {
    SEGK_HIP(hipDeviceSynchronize());
    SEGK_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(segk::g_clock_probe), sizeof(unsigned long long) * 1024 * 4));
    return SEGK_OK;
}
#endif
