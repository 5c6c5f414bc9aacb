/*
 * segk -- C ABI of the MI355X (gfx950) segmentation-training-step kernels.
 *
 * The reference (PraiseThampiV/image-segmentation) is pure Python on stock PyTorch ops; it has no
 * FFI.  Each entry point below therefore replaces the ATen call(s) issued by a reference line, which
 * is cited per function (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success or a negative SEGK_E* code; segk_last_error() gives the
 *     text of the last failure on the calling thread.
 *   - pointers are DEVICE pointers unless the name ends in _host.  The caller (PyTorch) owns every
 *     buffer; the library never allocates, frees, retains a pointer or synchronises.  Work is
 *     enqueued on `stream` (a hipStream_t passed as void*) of device `dev`.
 *   - activations are NHWC ("pixel-major"): element (n,y,x,c) of a tensor view lives at
 *     ptr[((n*H + y)*W + x)*ld + c]; `ld` (elements) may exceed C so that a view can be a channel
 *     slice of a wider concat buffer.  dtype is SEGK_F32 or SEGK_BF16 for activations and packed
 *     weights; statistics, master weights, gradients of weights, biases and logits are fp32.
 *   - ptr and ld*sizeof(elem) must be 16-byte aligned for activations; channel counts are multiples
 *     of 8.
 */
#ifndef SEGK_H
#define SEGK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { SEGK_F32 = 0, SEGK_BF16 = 1 } segk_dtype;

enum {
    SEGK_OK = 0,
    SEGK_EINVAL = -1,     /* bad enum / null pointer / flag combination */
    SEGK_ESHAPE = -2,     /* unsupported or inconsistent shape          */
    SEGK_EALIGN = -3,     /* pointer or leading dimension misaligned    */
    SEGK_EWORKSPACE = -4, /* workspace too small                        */
    SEGK_EHIP = -5        /* a HIP runtime call failed                  */
};

const char* segk_last_error(void);
int segk_version(void);

/* ------------------------------------------------------------------------------------------------
 * Losses.  logits are NCHW (the layout the networks return, models/unetvgg.py:105), fp32 or bf16;
 * labels are int64 [N,H,W] (loss/dice.py:47-52 scatter one-hot; loss/focal.py:102-103) or, for Dice
 * only, an explicit fp32 target `hot` [N,C,H,W] (loss/dice.py:53-54).
 * ---------------------------------------------------------------------------------------------- */

/* Bytes of workspace needed by segk_dice_focal_fwd for this shape. */
size_t segk_dice_focal_workspace(int64_t N, int64_t C, int64_t HW);

/* Fused softmax + squared-Dice sums + focal sum.
 * replaces: loss/dice.py:55-72 (Softmax2d, p*g, p*p, g*g sums) and loss/focal.py:134-136,64-75
 *           (cross_entropy(reduction='none'), softmax, focal weight, product, reduce).
 * out_sums  [N*C*4] fp32: per (n,c) {sum p*g, sum p*p, sum g*g, sum focal_c}            (written)
 * out_flags [N]     int32: 1 if image n has any non-zero target (loss/dice.py:82)        (written)
 * focal_map optional [N,C,H,W] fp32 elementwise focal values (reduction='none')          (written)
 * flags bit0: logits already are probabilities (no_sft, Dice only); bit3 (all three calls): self-adjusting Dice
 * (loss/dice.py:275-325): numerator sum p*g*(1-p), denominator sum p*(1-p) + sum g.                          */
int segk_dice_focal_fwd(const void* logits, int logits_dtype, const int64_t* labels, const float* hot,
                        int64_t N, int64_t C, int64_t HW, float gamma, const float* alpha, int flags,
                        float* out_sums, int32_t* out_flags, float* focal_map, void* workspace,
                        size_t workspace_bytes, int dev, void* stream);

/* segk_dice_focal_fwd for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): identical arguments and results, but the block partials go to a fixed
 * 1 MiB device array inside the library instead of a caller-provided workspace, and the call makes no hipMalloc /
 * hipFree (nor any other allocating call).  No counterpart in the reference.
 * Limit: segk_dice_focal_workspace(N, C, HW) <= 1048576 bytes (the query is the bound: it grows with the number of
 * blocks the kernel splits an image into, not only with N * C), larger shapes return SEGK_EWORKSPACE.  Every GPU the
 * library runs on holds its own copy of the array until the process ends; all calls on one GPU share it, so calls that
 * overlap on one GPU (different streams or threads) race -- serialise them, or use segk_dice_focal_fwd.  Nothing in the
 * array has to be cleared between calls.
 * The FIRST call on a device looks the array's address up (hipGetSymbolAddress, which may load the code object): make
 * one warm-up call per device outside any stream capture; every later call only launches kernels and can be captured. */
int segk_dice_focal_fwd_static(const void* logits, int logits_dtype, const int64_t* labels, const float* hot,
                               int64_t N, int64_t C, int64_t HW, float gamma, const float* alpha, int flags,
                               float* out_sums, int32_t* out_flags, float* focal_map, int dev, void* stream);

/* Reduce the per-(n,c) sums to the two scalars the reference returns.
 * replaces: loss/dice.py:72-87 (dice table, bg drop, fg-image average; loss = 1 - coeff at :23) and
 *           loss/focal.py:24-29 (mean over N*C*H*W or sum).
 * out_scalars[0] = dice coefficient (the loss is 1 - it, loss/dice.py:23), [1] = focal loss, [2] = 1/div (0 when no image is selected),
 * [3] = focal scale (1/(N*C*HW) for mean, 1 for sum).   flags bit1: keep channel 0 (is_all_chann),
 * bit2: focal reduction is 'sum'.                                                                  */
int segk_dice_focal_finalize(const float* sums, const int32_t* img_flags, int64_t N, int64_t C, int64_t HW,
                             int flags, float* out_scalars, int dev, void* stream);

/* dlogits = gscale[0]*d(dice loss)/dlogits + gscale[1]*d(focal loss)/dlogits   (closed forms of the
 * autograd graph of loss/dice.py:55-86 and loss/focal.py:64-75,134-136).  gscale is a DEVICE pointer
 * to two floats (upstream gradients), so no host synchronisation is needed.
 * dlogits has the dtype/layout of logits.                                                          */
int segk_dice_focal_bwd(const void* logits, int logits_dtype, const int64_t* labels, const float* hot,
                        int64_t N, int64_t C, int64_t HW, float gamma, const float* alpha, int flags,
                        const float* sums, const int32_t* img_flags, const float* scalars, const float* gscale,
                        void* dlogits, int dev, void* stream);

/* argmax mask + hard-Dice counts.  replaces loss/dice.py:129-163 (softmax, argmax, one-hot, sums).
 * mask [N,HW] uint8 (optional), counts [N*C*3] fp32 {sum phat*g, sum phat, sum g}, flags as above. */
int segk_argmax_dice(const void* logits, int logits_dtype, const int64_t* labels, int64_t N, int64_t C,
                     int64_t HW, uint8_t* mask, float* counts, int32_t* out_flags, int dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Layout conversion at the network boundary.
 * ---------------------------------------------------------------------------------------------- */

/* NCHW fp32 image [N,Cin,H,W] -> NHWC activation with Cout >= Cin channels (extra channels zero).
 * replaces: nothing in the reference (it stays NCHW); boundary of models/unetvgg.py:90.           */
int segk_nchw_to_nhwc(const float* src, int64_t N, int64_t Cin, int64_t H, int64_t W, void* dst,
                      int dst_dtype, int64_t Cout, int64_t ld, int dev, void* stream);

/* 3x3 patch gather of a few-channel NCHW fp32 image: dst[n,y,x, (ky*3+kx)*Cin + c] = src[n,c,y+ky-1,
 * x+kx-1] (zero outside), channels >= 9*Cin are zero.  Feeds the first VGG convolution
 * (models/unetvgg.py:72, Conv2d(3,64,3,padding=1)) to the MFMA kernel as a 1x1 convolution.       */
int segk_im2col3x3(const float* src, int64_t N, int64_t Cin, int64_t H, int64_t W, void* dst,
                   int dst_dtype, int64_t Cout, int64_t ld, int dev, void* stream);

/* The first convolution computed straight from the image: y = bias + conv3x3(x, w) for an fp32 NCHW image with 1..3
 * channels and the module's fp32 [Cout, Cin, 3, 3] weight, Cout in {32, 64}, bf16 NHWC output (no patch image, no
 * packed weights: the 9*Cin <= 27 products of a pixel are ONE 32-deep matrix step).
 * replaces: nn.Conv2d(3, 64, 3, padding=1) forward, models/unetvgg.py:72 (VGG features[0]); nn.Conv2d(in_ch, 64, 3,
 *           padding=1), models/unetplusplus.py:11,45.
 * flags bit0: ReLU.  stat_partials (optional, fp32 [segk_conv_first_stat_rows(N,H,W)][2][Cout]): per-wave sums and sums
 * of squares of the stored output, consumed by segk_norm_stats_finalize like those of segk_conv_fwd.          */
int64_t segk_conv_first_stat_rows(int64_t N, int64_t H, int64_t W);
int segk_conv_first_fwd(const float* x, int64_t N, int64_t Cin, int64_t H, int64_t W, const float* w, const float* bias,
                        int64_t Cout, void* y, int y_dtype, int64_t ldy, float* stat_partials, int flags, int dev,
                        void* stream);

/* The first convolution together with the BatchNorm2d + ReLU behind it, WITHOUT storing the convolution's output
 * z = bias + conv3x3(x, w): every mode recomputes z from the 1..3-channel image (a few matrix steps per tile) instead of
 * moving a 64-channel tensor.  Cout = 64, bf16 NHWC tensors, batch statistics.
 *   mode 1: partials <- per-wave sum and sum of squares of z (fp32 [segk_conv_first_stat_rows(N,H,W)][2][64];
 *           segk_norm_stats_finalize turns them into `stats` and updates the running buffers)
 *   mode 2: out <- relu(gamma * (z - mean) * rstd + beta)
 *   mode 3: partials <- per-wave sums of g*[pre > 0] and g*[pre > 0]*zhat for g = dL/d out (segk_norm_bwd_sums_finalize
 *           reduces them to `red`, dgamma, dbeta)
 *   mode 4: out <- dL/dz = gamma*rstd*(g*[pre > 0] - red[c][0]/M - zhat*red[c][1]/M), the input of segk_conv_first_wgrad
 * replaces: features[0..2] of models/unetvgg.py:72 (Conv2d(3,64) -> BatchNorm2d -> ReLU) and conv1/bn1/activation of
 *           models/unetplusplus.py:11-12,16-19 for conv0_0, forward and backward.                                   */
int segk_conv_first_norm(int mode, const float* x, int64_t N, int64_t Cin, int64_t H, int64_t W, const float* w,
                         const float* bias, int64_t Cout, const float* stats, const float* gamma, const float* beta,
                         const float* red, const void* g, int64_t ldg, void* out, int64_t ld_out, float* partials, int dev,
                         void* stream);

/* The whole backward of that fused first layer from g = dL/d relu(norm(z)) (bf16 NHWC, 64 channels): dw (fp32
 * [64, Cin, 3, 3]), dgamma, dbeta (flags bit1: all three accumulated).  Two streaming passes over g, each recomputing z
 * per tile: the first reduces the norm's two sums, the second turns g into dL/dz inside the LDS stage of the weight
 * gradient -- neither z nor dL/dz is ever written to memory.  (The bias gradient of a convolution in front of batch
 * statistics is identically zero.)  Deterministic: fixed-order reductions.  flags bit2: the norm's half only (first pass:
 * dgamma, dbeta, the sums kept in the workspace); bit3: the convolution's half only (second pass: dw, from the sums a
 * bit2 call left in the same workspace) -- for callers that time or schedule the two halves separately.
 * replaces: the backward of features[0..2], models/unetvgg.py:72; of conv1/bn1 of conv0_0, models/unetplusplus.py:11-12. */
size_t segk_conv_first_norm_bwd_workspace(int64_t N, int64_t H, int64_t W);
int segk_conv_first_norm_bwd(const float* x, int64_t N, int64_t Cin, int64_t H, int64_t W, const float* w, const float* bias,
                             const float* stats, const float* gamma, const float* beta, const void* g, int64_t ldg, float* dw,
                             float* dgamma, float* dbeta, int flags, void* workspace, size_t workspace_bytes, int dev,
                             void* stream);

/* Weight gradient of that first convolution, again straight from the image: dw (fp32, the module's [64, Cin, 3, 3]
 * layout) = sum over pixels of dy (bf16 NHWC, 64 channels) x the 3x3 neighbourhood of x; per-workgroup partial blocks
 * are added in fixed order (deterministic).  flags bit1: dw += result.
 * replaces: the weight-gradient half of that Conv2d's backward (the image needs no data gradient).             */
size_t segk_conv_first_wgrad_workspace(int64_t N, int64_t H, int64_t W);
int segk_conv_first_wgrad(const float* x, int64_t N, int64_t Cin, int64_t H, int64_t W, const void* dy, int dy_dtype,
                          int64_t ld_dy, int64_t Cout, float* dw, int flags, void* workspace, size_t workspace_bytes, int dev,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolutions on the matrix cores.
 * ---------------------------------------------------------------------------------------------- */

/* Input channels per K-row of the MFMA convolution for a dtype (64 for bf16, 32 for fp32).  Input
 * segments whose channel count is not a multiple of it are zero-filled up to the next K-row.      */
int segk_conv_krow(int dtype);

/* Pack fp32 OIHW weights [Cout,Cin,KH,KW] (KH=KW=3: taps=9, or 1x1: taps=1) into MFMA operand images:
 *   wf : forward image, consumed by segk_conv_fwd for y = conv(x, w).  If the input is the
 *        concatenation of two segments, seg0 = channels of the first one (0 = single segment): each
 *        segment starts on a K-row boundary of the image.
 *   wb : data-gradient image (taps flipped, in/out channels swapped), consumed by segk_conv_fwd
 *        called on dy (Cout channels) to produce dx with Cin channels.          (either may be NULL)
 * replaces: the weight operand of F.conv2d / its backward (models/unetvgg.py:28,34,38,72-76).
 * segk_conv_packed_bytes: size that holds either image.                                           */
size_t segk_conv_packed_bytes(int dtype, int64_t Cout, int64_t Cin, int taps, int64_t seg0);
int segk_conv_pack(const float* w_oihw, int64_t Cout, int64_t Cin, int taps, int64_t seg0, int dtype, void* wf,
                   void* wb, int dev, void* stream);

/* The same packing for many weight tensors in ONE launch (a training step repacks every layer after the optimizer
 * update): `descs_dev` is a DEVICE array of n descriptors; element (co, ci, tap) of tensor i is read at
 * w[co*w_so + ci*w_si + tap] (contiguous OIHW: w_so = Cin*taps, w_si = taps; depth tap kd of a [Cout,Cin,3,3,3]
 * Conv3d weight: w + kd*9, w_so = Cin*27, w_si = 27).  seg0 = Cin for a single-segment input.  wf / wb may be NULL.
 * blocks_per_desc workgroups share the slots of one descriptor.                                              */
typedef struct segk_pack_desc {
    const float* w;
    void* wf;
    void* wb;
    int32_t Cout, Cin, taps, seg0;
    int64_t w_so, w_si;
} segk_pack_desc;
int segk_conv_pack_batch(int dtype, const segk_pack_desc* descs_dev, int64_t n, int64_t blocks_per_desc, int dev, void* stream);

/* y[n,y,x,co] = act( bias[co] + sum_{tap,ci} x[n,y+ky-1,x+kx-1,ci] * w[co,ci,ky,kx] ) * (mask>0)
 * replaces: nn.Conv2d(k=3,padding=1) forward (models/unetvgg.py:28,34,38,72-76,87;
 *           models/unetplusplus.py:11,13) and -- called with the `wb` image on dy -- the
 *           data-gradient half of its backward; taps=1 gives a 1x1 convolution.
 * The input is the channel concatenation of up to two segments (x0 then x1; torch.cat at
 * models/unetvgg.py:99-103, models/unetplusplus.py:64-79); C0, C1, Cout and mask_channels multiples of 8.
 * flags bit0: ReLU epilogue; mask (optional, same dtype, ld_mask): output channels [0, mask_channels) are zeroed
 * where mask<=0: the ReLU backward of the producer(s) of those channels, fused into the data gradient.
 * flags bit1: accumulate into y (y += result) instead of overwriting.
 * y1 (optional): second destination -- output channels [split, Cout) are written to y1[..., c - split] (flags
 * bit2: accumulated there); the data gradient of a two-segment input lands directly in both segments' buffers
 * (a mask then has to lie within the first destination: mask_channels <= split).  */
int segk_conv_fwd(int dtype, int taps, const void* x0, int64_t ld0, int64_t C0, const void* x1, int64_t ld1,
                  int64_t C1, int64_t N, int64_t H, int64_t W, const void* w_packed, int64_t Cout,
                  const float* bias, void* y, int64_t ldy, void* y1, int64_t ldy1, int64_t split, const void* mask,
                  int64_t ld_mask, int64_t mask_channels, float* stat_partials, int flags, int dev, void* stream);
/* stat_partials (optional, fp32 [segk_conv_stat_rows(N,H,W,Cout)][2][Cout]): per (image, tile, pixel-wave) the sum and
 * the sum of squares of the output as stored, per channel -- the batch / instance statistics of the BatchNorm or
 * InstanceNorm that follows (models/unetvgg.py:29-30,72-76; models/unetplusplus.py:11-14) without another pass over
 * y; segk_norm_stats_finalize turns them into the stats vector.  Needs one destination, no mask, no accumulation.  */
int64_t segk_conv_stat_rows(int64_t N, int64_t H, int64_t W, int64_t Cout);
/* flags bit3 (with stat_partials): the buffer has segk_conv_stat_rows_fine(N,H,W) rows instead -- four per (image, 8x16
 * tile), one for each pair of pixel rows, whatever kernel runs (rows a kernel does not own are written as zeros).  Only
 * with this layout can the thin-layer kernel, which keeps no sums across its steps, serve a layer that wants statistics;
 * segk_norm_stats_finalize takes either layout (it only sums rows).  */
int64_t segk_conv_stat_rows_fine(int64_t N, int64_t H, int64_t W);

/* Data gradient of a convolution whose input was h = relu(batchnorm(z)) and had no other reader
 * (conv -> BatchNorm -> ReLU -> conv: models/unetvgg.py:72-76 through the VGG stages, models/unetplusplus.py:20-25):
 * g = segk_conv_fwd(dy, data-gradient image) [N,H,W,Cout], and in the same epilogue the two sums the backward of that
 * BatchNorm needs -- partials[row][0][c] = sum g*[norm(z)>0], partials[row][1][c] = sum g*[norm(z)>0]*zhat over the
 * pixels of (image, tile, pixel-wave) `row`, segk_conv_stat_rows(N,H,W,Cout) rows -- so that segk_norm_bwd_finish
 * needs no reduction pass over g and z.  z: the norm's input (same dtype, pixel stride ldz), stats {mean, 1/std}
 * interleaved as segk_norm_stats writes them, gamma/beta may be NULL.
 * replaces: half of the first pass of autograd's batch_norm backward, fused into the convolution backward in front of it. */
int segk_conv_dgrad_norm(int dtype, int taps, const void* dy, int64_t ld_dy, int64_t C, int64_t N, int64_t H, int64_t W,
                         const void* w_packed, int64_t Cout, void* g, int64_t ldg, const void* z, int64_t ldz,
                         const float* stats, const float* gamma, const float* beta, float* partials, int dev,
                         void* stream);

/* dw[co,ci,ky,kx] (fp32 OIHW, [Cout, cin_real, taps]) = sum_{n,y,x} dy[n,y,x,co] * x[n,y+ky-1,x+kx-1,ci]
 * replaces: the weight-gradient half of Conv2d's backward.  The pixels are split over workgroups
 * whose partial slabs are summed in fixed order (deterministic).  flags bit1: dw += result.
 * cin_real: number of input channels present in dw; C0+C1 unless a single segment carries trailing
 * padding channels (the im2col image of the first convolution).                                   */
size_t segk_conv_wgrad_workspace(int dtype, int taps, int64_t C0, int64_t C1, int64_t N, int64_t H, int64_t W,
                                 int64_t Cout);
int segk_conv_wgrad(int dtype, int taps, const void* x0, int64_t ld0, int64_t C0, const void* x1, int64_t ld1,
                    int64_t C1, int64_t N, int64_t H, int64_t W, const void* dy, int64_t ldy, int64_t Cout,
                    int64_t cin_real, float* dw, int flags, void* workspace, size_t workspace_bytes, int dev,
                    void* stream);

/* ConvTranspose2d(kernel 4, stride 2, padding 1): [N,H,W,Cin] -> [N,2H,2W,Cout], run as four phase convolutions
 * (2x2 taps each) on the same MFMA kernels.  w is the module's fp32 [Cin,Cout,4,4] weight.
 * replaces: nn.ConvTranspose2d(mid, out, kernel_size=4, stride=2, padding=1) forward/backward of the
 *           `if_convtrans=True` decoder tail (models/unetvgg.py:31-32).
 * segk_convt_pack writes the per-phase forward images into wf and data-gradient images into wb (either may be
 * NULL; each buffer segk_convt_packed_bytes large).  fwd flags bit0: ReLU epilogue.  dgrad: dx (H x W, Cin) from
 * dy (2H x 2W, Cout), optional ReLU-backward mask of the producer of x, flags bit1: accumulate.  wgrad: dw in the
 * module's [Cin,Cout,4,4] layout, flags bit1: accumulate.                                                  */
size_t segk_convt_packed_bytes(int dtype, int64_t Cin, int64_t Cout);
int segk_convt_pack(const float* w, int64_t Cin, int64_t Cout, int dtype, void* wf, void* wb, int dev, void* stream);
int segk_convt_fwd(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t H, int64_t W, const void* wf,
                   int64_t Cout, const float* bias, void* y, int64_t ldy, int flags, int dev, void* stream);
int segk_convt_dgrad(int dtype, const void* dy, int64_t ld_dy, int64_t Cout, int64_t N, int64_t H, int64_t W,
                     const void* wb, int64_t Cin, void* dx, int64_t ld_dx, const void* mask, int64_t ld_mask, int flags,
                     int dev, void* stream);
size_t segk_convt_wgrad_workspace(int dtype, int64_t Cin, int64_t Cout, int64_t N, int64_t H, int64_t W);
int segk_convt_wgrad(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t H, int64_t W, const void* dy,
                     int64_t ld_dy, int64_t Cout, float* dw, int flags, void* workspace, size_t workspace_bytes, int dev,
                     void* stream);

/* ---- volumes (the 3-D classifier of models/resnt.py:26-45 = DetectMid over models/densenet3d.py:71-159) ----------
 * A batch of volumes [N, C, D, H, W] is stored slice by slice: [N*D, H, W, ld], channels innermost, so the
 * channel-wise entry points above (norm_*, conv with taps=1, colsum, copy_slice) take it with N*D as the batch.
 *
 * Conv3d(kernel 3, stride 1, padding 1, bias=False): ONE launch of the 3x3 MFMA kernel whose K dimension runs over the
 * three depth taps, each reading the neighbouring slice of the same volume (zero beyond its faces); fp32 accumulation
 * over all 27 taps.  The weight gradient is one launch for the three depth taps plus one reduction.
 * replaces: nn.Conv3d(bn_size*growth_rate, growth_rate, kernel_size=3, stride=1, padding=1, bias=False) forward and
 *           backward (models/densenet3d.py:27-32).  w: the module's fp32 [Cout, Cin, 3, 3, 3] weight.
 * segk_conv3d_fwd with the `wb` image on dy (Cout channels) yields dx (Cin channels).  flags bit1: accumulate.
 * dw: fp32 [Cout, Cin, 3, 3, 3]; flags bit1: dw += result.                                                    */
size_t segk_conv3d_packed_bytes(int dtype, int64_t Cout, int64_t Cin);
int segk_conv3d_pack(const float* w, int64_t Cout, int64_t Cin, int dtype, void* wf, void* wb, int dev, void* stream);
int segk_conv3d_fwd(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W,
                    const void* w_packed, int64_t Cout, void* y, int64_t ldy, int flags, int dev, void* stream);
size_t segk_conv3d_wgrad_workspace(int dtype, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W, int64_t Cout);
int segk_conv3d_wgrad(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W,
                      const void* dy, int64_t ld_dy, int64_t Cout, float* dw, int flags, void* workspace,
                      size_t workspace_bytes, int dev, void* stream);

/* Patch gather for the strided stem convolutions: patches[n,od,oh,ow, ((a*kh+b)*kw+e)*C+c] =
 * x[n, od*sd-pd+a, oh*sh-ph+b, ow*sw-pw+e, c] for the first C (<= 8) channels of x, zero outside the volume and in
 * columns [C*kd*kh*kw, Kp); kernel/stride/pad are int[3] = (depth, height, width).  Columns are tap-major,
 * channel-minor: with the Conv3d weight permuted to [Cout, kd, kh, kw, C] and flattened, the convolution is
 * segk_conv_fwd(taps=1) on the patches.
 * replaces: the im2col half of nn.Conv3d(in_classes, 3, (3,7,7), stride (1,2,2), padding (1,3,3)) (models/resnt.py:30)
 *           and nn.Conv3d(3, 64, (7,7,7), stride (1,2,2), padding (3,3,3)) (models/densenet3d.py:98-104).
 * segk_col2im3d is its adjoint (the data gradient w.r.t. x from the patch gradient); flags bit1: accumulate.     */
int segk_im2col3d(int dtype, const void* x, int64_t ldx, int64_t C, int64_t N, int64_t D, int64_t H, int64_t W,
                  const int* kernel, const int* stride, const int* pad, void* patches, int64_t ld_patches, int64_t Kp,
                  int dev, void* stream);
int segk_col2im3d(int dtype, const void* dpatches, int64_t ld_patches, int64_t C, int64_t N, int64_t D, int64_t H, int64_t W,
                  const int* kernel, const int* stride, const int* pad, void* dx, int64_t ld_dx, int flags, int dev,
                  void* stream);

/* Conv3d with kernel (kd, 1, 1), stride 1, padding (kd/2, 0, 0), bias=False, kd in {1,3,5,7}: the depth half of a
 * factorised stem convolution.  The strided (kh, kw) window of every slice is gathered once into patch rows
 * (segk_im2col3d with kernel (1,kh,kw)), and this convolution sums the kd neighbouring slices' rows: together they
 * equal nn.Conv3d(C, Cout, (kd,kh,kw), stride (1,sh,sw), padding (kd/2,ph,pw)) with a patch tensor kd times smaller than a
 * full 3-D gather (models/resnt.py:30; models/densenet3d.py:98-104).  w: fp32 [Cout][Cin][kd] with Cin = patch columns.
 * segk_convd_fwd with the `wb` image on dy gives the patch gradient (-> segk_col2im3d).  stat_partials as segk_conv_fwd.
 * dw: fp32 [Cout][cin_real][kd].  flags bit1: accumulate.                                                            */
size_t segk_convd_packed_bytes(int dtype, int64_t Cout, int64_t Cin, int64_t kd);
int segk_convd_pack(const float* w, int64_t Cout, int64_t Cin, int64_t kd, int dtype, void* wf, void* wb, int dev, void* stream);
int segk_convd_fwd(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W, int64_t kd,
                   const void* w_packed, int64_t Cout, void* y, int64_t ldy, float* stat_partials, int flags, int dev,
                   void* stream);
size_t segk_convd_wgrad_workspace(int dtype, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W, int64_t Cout, int64_t kd);
int segk_convd_wgrad(int dtype, const void* x, int64_t ldx, int64_t Cin, int64_t N, int64_t D, int64_t H, int64_t W, int64_t kd,
                     const void* dy, int64_t ld_dy, int64_t Cout, int64_t cin_real, float* dw, int flags, void* workspace,
                     size_t workspace_bytes, int dev, void* stream);

/* MaxPool3d(kernel 3, stride 2, padding 1): [N,D,H,W,C] -> [N,(D-1)/2+1,...]; argmax receives one byte per output
 * element (winning window position, first maximum in scan order) and drives the backward gather.
 * replaces: nn.MaxPool3d(kernel_size=3, stride=2, padding=1) (models/densenet3d.py:107-109).  flags bit1: accumulate. */
int segk_maxpool3d_fwd(int dtype, const void* x, int64_t ldx, int64_t N, int64_t D, int64_t H, int64_t W, int64_t C,
                       void* y, int64_t ldy, unsigned char* argmax, int dev, void* stream);
int segk_maxpool3d_bwd(int dtype, const void* dy, int64_t ld_dy, const unsigned char* argmax, int64_t N, int64_t D,
                       int64_t H, int64_t W, int64_t C, void* dx, int64_t ld_dx, int flags, int dev, void* stream);

/* AvgPool3d(kernel 2, stride 2) and its backward.
 * replaces: nn.AvgPool3d(kernel_size=2, stride=2) of the transitions (models/densenet3d.py:68).                  */
int segk_avgpool3d_fwd(int dtype, const void* x, int64_t ldx, int64_t N, int64_t D, int64_t H, int64_t W, int64_t C,
                       void* y, int64_t ldy, int dev, void* stream);
int segk_avgpool3d_bwd(int dtype, const void* dy, int64_t ld_dy, int64_t N, int64_t D, int64_t H, int64_t W, int64_t C,
                       void* dx, int64_t ld_dx, int flags, int dev, void* stream);

/* 1x1 head: logits[n,co,y,x] (NCHW fp32) = bias[co] + sum_ci x[n,y,x,ci]*w[co,ci].
 * replaces: models/unetvgg.py:88,105 (final = Conv2d(32,C,1)); models/unetplusplus.py:58,81.      */
int segk_head1x1_fwd(int dtype, const void* x, int64_t ld, int64_t Cin, int64_t N, int64_t HW, const float* w,
                     const float* bias, int64_t Cout, float* logits, int dev, void* stream);
/* backward of the head: dx (NHWC, optionally masked by relu_out>0), dw [Cout,Cin], db [Cout].     */
size_t segk_head1x1_bwd_workspace(int64_t Cin, int64_t Cout, int64_t N, int64_t HW);
int segk_head1x1_bwd(int dtype, const void* x, int64_t ld, int64_t Cin, int64_t N, int64_t HW, const float* w,
                     int64_t Cout, const float* dlogits, void* dx, int64_t ld_dx, int relu_mask, float* dw,
                     float* db, void* workspace, size_t workspace_bytes, int dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation / activation / resampling streams (HBM-bound).
 * `groups` statistics groups of `M` pixels each: BatchNorm2d -> groups=1, M=N*H*W;
 * InstanceNorm2d -> groups=N, M=H*W.   C/8 must be a power of two <= 256.
 * ---------------------------------------------------------------------------------------------- */
size_t segk_norm_workspace(int64_t groups, int64_t M, int64_t C);

/* mean / rstd (biased variance, eps) per (group, channel); optionally the BatchNorm running-buffer
 * update running = (1-momentum)*running + momentum*stat with the UNBIASED variance.
 * replaces: statistics half of F.batch_norm(training=True) / F.instance_norm
 *           (models/unetvgg.py:29,35,70-76; models/unetplusplus.py:12,14).
 * stats [groups*C*2] fp32 {mean, rstd} interleaved per channel.                                    */
int segk_norm_stats(int dtype, const void* x, int64_t ld, int64_t groups, int64_t M, int64_t C, float eps,
                    float* stats, float* running_mean, float* running_var, float momentum, void* workspace,
                    size_t workspace_bytes, int dev, void* stream);

/* y = act( (x-mean)*rstd*gamma + beta ); gamma/beta may be NULL (InstanceNorm2d(affine=False)).
 * flags bit0: ReLU.  `stats` as produced above, or built from running buffers for eval mode.
 * replaces: normalise half of batch_norm / instance_norm + nn.ReLU(inplace=True).                  */
int segk_norm_apply(int dtype, const void* x, int64_t ldx, int64_t groups, int64_t M, int64_t C,
                    const float* stats, const float* gamma, const float* beta, int flags, void* y, int64_t ldy,
                    int dev, void* stream);

/* Same result as segk_norm_stats from partial sums produced elsewhere (segk_conv_fwd's stat_partials):
 * partials [groups][rows_per_group][2][C] = plain sums and sums of squares over disjoint pixel sets of a group of
 * M pixels.                                                                                          */
size_t segk_norm_stats_finalize_workspace(int64_t groups, int64_t rows_per_group, int64_t C);
int segk_norm_stats_finalize(const float* partials, int64_t groups, int64_t rows_per_group, int64_t M, int64_t C, float eps,
                             float* stats, float* running_mean, float* running_var, float momentum, void* workspace,
                             size_t workspace_bytes, int dev, void* stream);

/* stats for eval mode: {running_mean, 1/sqrt(running_var+eps)} interleaved.                        */
int segk_norm_eval_stats(const float* running_mean, const float* running_var, int64_t C, float eps,
                         float* stats, int dev, void* stream);

/* Backward of y = act(norm(x)):  g = dy * (y_pre>0 if ReLU);  dgamma = sum g*xhat, dbeta = sum g,
 * dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)).  Two launches inside: reduce, then dx.
 * dgamma/dbeta (fp32 [C], BatchNorm only) may be NULL; dx may alias dy.  flags bit0: ReLU,
 * bit1: dgamma/dbeta += , bit2: frozen (running) statistics, bit3: dx += (x has other consumers whose share is
 * already in dx; dx must then not alias dy).   dbias_out (optional fp32 [C]): column sum of dx (bias gradient of the
 * convolution in front of the norm).
 * replaces: autograd backward of batch_norm / instance_norm + ReLU.                                */
int segk_norm_bwd(int dtype, const void* dy, int64_t ld_dy, const void* x, int64_t ldx, int64_t groups,
                  int64_t M, int64_t C, const float* stats, const float* gamma, const float* beta, int flags,
                  void* dx, int64_t ld_dx, float* dgamma, float* dbeta, float* dbias_out, void* workspace,
                  size_t workspace_bytes, int dev, void* stream);

/* segk_norm_bwd for one group (BatchNorm, batch statistics) whose two sums arrive as the partial rows of
 * segk_conv_dgrad_norm ([rows][2][C]): fixed-order reduction of the rows, dgamma/dbeta, then the dx pass.
 * flags as segk_norm_bwd (bit2 not allowed).                                                        */
size_t segk_norm_bwd_finish_workspace(int64_t rows, int64_t C);
int segk_norm_bwd_finish(int dtype, const float* partials, int64_t rows, const void* dy, int64_t ld_dy, const void* x,
                         int64_t ldx, int64_t M, int64_t C, const float* stats, const float* gamma, const float* beta,
                         int flags, void* dx, int64_t ld_dx, float* dgamma, float* dbeta, void* workspace,
                         size_t workspace_bytes, int dev, void* stream);

/* The reduction half of segk_norm_bwd_finish alone: partial rows [rows][2][C] of (sum g, sum g*xhat) -> red[C][2], dgamma,
 * dbeta (flags bit1: accumulated), for a producer that computes dx itself (segk_conv_first_norm modes 3 and 4).
 * workspace: segk_norm_stats_finalize_workspace(1, rows, C) bytes (none needed for rows <= 128).
 * replaces: the per-channel sums of BatchNorm2d's backward (models/unetvgg.py:72 features[1]).                 */
int segk_norm_bwd_sums_finalize(const float* partials, int64_t rows, int64_t C, float* red, float* dgamma, float* dbeta,
                                int flags, void* workspace, size_t workspace_bytes, int dev, void* stream);

/* BatchNorm (batch or running statistics in `stats`) + ReLU + the 2x2/stride-2 max pool behind it, one pass:
 * y = relu(norm(x)) [N,H,W,C] and pool = maxpool2x2(y) [N,H/2,W/2,C].
 * replaces: the last BatchNorm2d/ReLU of an encoder stage together with the MaxPool2d that opens the next one
 *           (models/unetvgg.py:72-79; models/unetplusplus.py:35,45-58).
 * segk_norm_pool_bwd: dx = d loss / d x from dy (gradient reaching y from its other consumers, e.g. the skip
 * connection; may be NULL) and dpool (gradient of the pooled tensor, routed to each window's first maximum);
 * dgamma/dbeta as segk_norm_bwd.  flags bit1: accumulate dgamma/dbeta, bit2: frozen (running) statistics.
 * workspace: segk_norm_workspace(1, N*H*W, C) bytes.                                               */
int segk_norm_pool_apply(int dtype, const void* x, int64_t ldx, int64_t N, int64_t H, int64_t W, int64_t C,
                         const float* stats, const float* gamma, const float* beta, void* y, int64_t ldy, void* pool,
                         int64_t ld_pool, int dev, void* stream);
int segk_norm_pool_bwd(int dtype, const void* dy, int64_t ld_dy, const void* dpool, int64_t ld_dpool, const void* x,
                       int64_t ldx, int64_t N, int64_t H, int64_t W, int64_t C, const float* stats, const float* gamma,
                       const float* beta, int flags, void* dx, int64_t ld_dx, float* dgamma, float* dbeta,
                       void* workspace, size_t workspace_bytes, int dev, void* stream);

/* 2x2/stride-2 max pool (models/unetvgg.py:73-79 VGG pools; models/unetplusplus.py:35).           */
int segk_maxpool2x2_fwd(int dtype, const void* x, int64_t ldx, int64_t N, int64_t H, int64_t W, int64_t C,
                        void* y, int64_t ldy, int dev, void* stream);
/* dx (+)= route(dy) to the first maximum in scan order.  flags bit1: accumulate into dx.          */
int segk_maxpool2x2_bwd(int dtype, const void* x, int64_t ldx, const void* dy, int64_t ld_dy, int64_t N,
                        int64_t H, int64_t W, int64_t C, void* dx, int64_t ld_dx, int flags, int dev,
                        void* stream);

/* y[n, 2h x 2w, c] = act( (bilinear_x2(x) - mean)*rstd ), align_corners selectable; stats NULL means
 * no normalisation.  replaces: InstanceNorm2d + nn.Upsample(scale_factor=2, mode='bilinear') + ReLU
 * of the decoder block (models/unetvgg.py:35-36,41) and nn.Upsample(align_corners=True)
 * (models/unetplusplus.py:36).   flags bit0: ReLU, bit2: align_corners=True.                       */
int segk_upsample2x_fwd(int dtype, const void* x, int64_t ldx, int64_t N, int64_t H, int64_t W, int64_t C,
                        const float* stats, int flags, void* y, int64_t ldy, int dev, void* stream);
/* adjoint: dxhat[n,h,w,c] = sum over the output pixels that read (h,w) of weight * dy * (y>0 if ReLU);
 * y (the saved forward output) is only read when bit0 is set.                                      */
int segk_upsample2x_bwd(int dtype, const void* dy, int64_t ld_dy, const void* y, int64_t ldy, int64_t N,
                        int64_t H, int64_t W, int64_t C, int flags, void* dx, int64_t ld_dx, int dev,
                        void* stream);

/* dx = dy where y > 0 else 0 over a [P,C] slice: backward of a ReLU whose producer has no norm to fuse it into
 * (the ReLU behind the transposed convolution, models/unetvgg.py:41).                              */
int segk_relu_bwd(int dtype, const void* dy, int64_t ld_dy, const void* y, int64_t ldy, int64_t P, int64_t C, void* dx,
                  int64_t ld_dx, int dev, void* stream);

/* out[c] = sum over pixels of x[p,c] (fp32, deterministic two-stage); flags bit1: out += .
 * replaces: bias gradient of Conv2d/ConvTranspose2d backward.                                      */
size_t segk_colsum_workspace(int64_t P, int64_t C);
int segk_colsum(int dtype, const void* x, int64_t ld, int64_t P, int64_t C, float* out, int flags,
                void* workspace, size_t workspace_bytes, int dev, void* stream);

/* dst[p, c] (+)= src[p, c] for a [P,C] channel slice (gradient fan-in of a tensor with several
 * consumers, e.g. the dense skips of models/unetplusplus.py:64-79).  flags bit1: accumulate.
 * src == NULL: the slice is filled with zeros.                                                     */
int segk_copy_slice(int dtype, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t P,
                    int64_t C, int flags, int dev, void* stream);

/* Bilinear resize by any factor, align_corners=False: y[n, Ho x Wo, c] = act((bilinear(x) - mean) * rstd), source coordinate
 * max(ratio * (o + 0.5) - 0.5, 0) with ratio = 1 / scale_factor as ATen forms it; stats NULL means no normalisation.
 * flags bit0: ReLU.   _bwd: dx (+)= adjoint of the interpolation applied to dy (gather form, deterministic); flags bit1: dx += .
 * replaces: InstanceNorm2d + nn.Upsample(scale_factor=upsc, mode='bilinear') + ReLU of make_decoder_block for upsc other
 * than 2 (models/unetvgg.py:36, used with upsc = 8, 4, 0.5 by SpiderNet :955-960).                    */
int segk_resize_bilinear_fwd(int dtype, const void* x, int64_t ldx, int64_t N, int64_t H, int64_t W, int64_t C,
                             int64_t Ho, int64_t Wo, float ratio_h, float ratio_w, const float* stats, int flags,
                             void* y, int64_t ldy, int dev, void* stream);
int segk_resize_bilinear_bwd(int dtype, const void* dy, int64_t ld_dy, int64_t N, int64_t H, int64_t W, int64_t C,
                             int64_t Ho, int64_t Wo, float ratio_h, float ratio_w, void* dx, int64_t ld_dx, int flags,
                             int dev, void* stream);

/* 2x2 pixel blocks <-> channels: wide[n,oh,ow,(b*2+e)*C + c] = x[n,2*oh+b,2*ow+e,c] ([N,H,W,C] -> [N,H/2,W/2,4C]).
 * With the weight read as [Cout][(b,e,c)] a Conv2d(kernel_size=2, stride=2) becomes a 1x1 convolution over
 * `wide` (segk_conv_fwd with taps = 1).  flags bit0: run backwards (x from wide: the data gradient);
 * bit1 (with bit0): x += .
 * replaces: the strided down-sampling Conv2d of make_encoder_block, models/unetvgg.py:305.          */
int segk_block2(int dtype, void* x, int64_t ldx, void* wide, int64_t ld_wide, int64_t N, int64_t H, int64_t W,
                int64_t C, int flags, int dev, void* stream);

/* Edge map of an fp32 [planes, H, W] tensor (planes = N*C): per plane the two Sobel derivatives (zero padding), each
 * instance-normalised (biased variance, eps), rectified in both directions and summed:
 *     edge = |IN(sobel_x)| + |IN(sobel_y)|;   stats[plane] = (mean_x, rstd_x, mean_y, rstd_y) is kept for the backward.
 * segk_edge_bwd: dx = d loss / d x from dedge = d loss / d edge (same shape).
 * replaces: get_edg_c, loss/hausdorff.py:71-108 (four grouped F.conv2d + nn.InstanceNorm2d + ReLU), the operator
 * behind edge_conv_loss (:110-128), whose Dice part is segk_dice_focal_* with a float target.        */
size_t segk_edge_workspace(int64_t planes, int64_t H, int64_t W);
size_t segk_edge_bwd_workspace(int64_t planes, int64_t H, int64_t W);
int segk_edge_fwd(const float* x, int64_t planes, int64_t H, int64_t W, float eps, float* edge, float* stats,
                  void* workspace, size_t workspace_bytes, int dev, void* stream);
int segk_edge_bwd(const float* x, const float* dedge, const float* stats, int64_t planes, int64_t H, int64_t W,
                  float* dx, void* workspace, size_t workspace_bytes, int dev, void* stream);

/* Fused multi-tensor Adam step (torch.optim.Adam defaults; perlosstrain/naive_train.py:397-398,
 * pl_utils/pl_mod.py:109-110).  One flat fp32 parameter / gradient / moment buffer.                */
int segk_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                   int dev, void* stream);

/* ---- classifier tail of the 3-D middle-slice network ------------------------------------------
 * out[n][c] (fp32) = mean over the V voxels of sample n of x[(n*V + v)*ld + c]: the global average pool behind the
 * last BatchNorm3d + ReLU (F.adaptive_avg_pool3d(.., 1), models/densenet3d.py:153-157); _bwd is its adjoint
 * (dx = dout / V at every voxel, overwriting).                                                     */
int segk_rows_mean_fwd(int dtype, const void* x, int64_t ld, int64_t N, int64_t V, int64_t C, float* out, int dev,
                       void* stream);
int segk_rows_mean_bwd(int dtype, const float* dout, int64_t N, int64_t V, int64_t C, void* dx, int64_t ld, int dev,
                       void* stream);

/* y[N][O] = x[N][K] w[O][K]^T + bias (fp32, contiguous; bias may be NULL)
 * replaces: nn.Linear forward (models/densenet3d.py:139,158; models/resnt.py:34-35,41-44) on the pooled features.  */
int segk_linear_fwd(const float* x, const float* w, const float* bias, int64_t N, int64_t K, int64_t O, float* y, int dev,
                    void* stream);
/* dw[O][K] = dy^T x, dbias[O] = column sums of dy (optional), dx[N][K] = dy w (optional); flags bit1: dw/dbias += .
 * Fixed summation order (deterministic).   replaces: nn.Linear backward.                           */
int segk_linear_bwd(const float* x, const float* w, const float* dy, int64_t N, int64_t K, int64_t O, float* dx, float* dw,
                    float* dbias, int flags, int dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGK_H */
