"""Tensor-level wrappers over the C ABI: shape bookkeeping only, no arithmetic.

Activations are `Act` views: channels [c0, c0+C) of an NHWC buffer [N,H,W,ld].  Producers write
straight into a channel slice of a wider buffer, which is how torch.cat of the reference
(models/unetvgg.py:99-103) disappears.
"""
import contextlib

import torch

from . import _lib
from ._lib import SEGK_BF16, SEGK_F32, call, query

_DT = {torch.float32: SEGK_F32, torch.bfloat16: SEGK_BF16}
RELU, ACCUM, ALIGN, FROZEN = 1, 2, 4, 4   # flag bits (ALIGN: upsample; FROZEN: norm_bwd)


def dt(dtype):
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"segk computes in float32 or bfloat16, not {dtype}") from None


def krow(dtype):
    return 64 if dtype == torch.bfloat16 else 32


class Act:
    """View of channels [c0, c0+C) of the NHWC buffer `buf` ([N,H,W,ld], contiguous)."""
    __slots__ = ("buf", "c0", "C")

    def __init__(self, buf, c0=0, C=None):
        assert buf.dim() == 4 and buf.is_contiguous()
        self.buf, self.c0 = buf, c0
        self.C = buf.shape[3] - c0 if C is None else C
        assert 0 <= c0 and c0 + self.C <= buf.shape[3]

    N = property(lambda s: s.buf.shape[0])
    H = property(lambda s: s.buf.shape[1])
    W = property(lambda s: s.buf.shape[2])
    ld = property(lambda s: s.buf.shape[3])
    dtype = property(lambda s: s.buf.dtype)
    device = property(lambda s: s.buf.device)
    P = property(lambda s: s.buf.shape[0] * s.buf.shape[1] * s.buf.shape[2])

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.c0 * self.buf.element_size()

    def slice(self, c0, C):
        return Act(self.buf, self.c0 + c0, C)

    def nchw(self):
        """Logical NCHW tensor (a strided view) for tests and debugging."""
        return self.buf[..., self.c0:self.c0 + self.C].permute(0, 3, 1, 2)

    @staticmethod
    def empty(N, H, W, C, dtype, device, zero=False):
        mk = torch.zeros if zero else torch.empty
        return Act(mk((N, H, W, C), dtype=dtype, device=device))

    @staticmethod
    def from_nchw(x, dtype=None):
        """Test helper: copy a logical NCHW tensor into a fresh NHWC buffer."""
        return Act(x.permute(0, 2, 3, 1).contiguous().to(dtype or x.dtype))


def _env(t):
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    return dev, torch.cuda.current_stream(dev).cuda_stream


_workspaces = {}

# Optional per-launch timing of the MFMA convolutions (bench.py's roofline leg): a list that receives
# (kind, start_event, end_event, algorithmic FLOP, kernel variant, shape text) for every conv launch, recorded on the
# stream the kernel runs on.  The HBM-bound passes (normalisation, up-sampling, loss) are recorded the same way with kind
# "stream", their algorithmic bytes in place of the FLOP and their family name as the variant.
CONV_TIMER = None


class _Timed:
    def __init__(self, kind, flops=0.0, variant="", desc=""):
        self.kind, self.flops, self.variant, self.desc = kind, flops, variant, desc

    def __enter__(self):
        if CONV_TIMER is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if CONV_TIMER is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            CONV_TIMER.append((self.kind, self.e0, e1, self.flops, self.variant, self.desc))


STREAM_TIMING = True     # with CONV_TIMER set: also record the HBM-bound passes (bench.py turns it on for a few steps only)


def _streamed(family, nbytes, desc=""):
    return _Timed("stream", float(nbytes), family, desc) if (CONV_TIMER is not None and STREAM_TIMING) else contextlib.nullcontext()


def workspace(nbytes, device):
    """Scratch buffer per (device, stream): every kernel that uses it runs on that stream, so successive uses are
    ordered."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _p(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------ layout
def im2col3x3(x, dtype):
    """[N,Cin,H,W] fp32 -> NHWC patch image with 32 channels (9*Cin real ones, the rest zero): half a bf16
    K-row, so the convolution multiplies it with a single k-substep."""
    N, Cin, H, W = x.shape
    C = 32
    assert 9 * Cin <= C, "im2col path is for few-channel inputs"
    x = x.contiguous().float()
    out = Act.empty(N, H, W, C, dtype, x.device)
    dev, st = _env(x)
    call("segk_im2col3x3", x.data_ptr(), N, Cin, H, W, out.ptr, dt(dtype), C, out.ld, dev, st)
    return out


def conv_first_direct_ok(x, Cout, dtype):
    """The direct first-layer kernel serves bf16 outputs of 32 or 64 channels from images with up to 3 channels."""
    return dtype == torch.bfloat16 and x.shape[1] <= 3 and Cout in (32, 64)


def conv_first_fwd(x, w, bias, out, relu=False, want_stats=False):
    """out (bf16 NHWC Act) = bias + conv3x3(x, w) straight from the fp32 NCHW image x; -> statistics partials or None."""
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    assert out.C == Cout and out.dtype == torch.bfloat16 and (out.N, out.H, out.W) == (N, H, W)
    x = x.contiguous().float()
    part = None
    if want_stats:
        part = torch.empty(query("segk_conv_first_stat_rows", N, H, W), 2, Cout, dtype=torch.float32, device=x.device)
    dev, st = _env(x)
    with _Timed("conv_fwd", 2.0 * N * H * W * Cout * 32, "first layer"):
        call("segk_conv_first_fwd", x.data_ptr(), N, Cin, H, W, w.detach().contiguous().data_ptr(), _p(bias), Cout, out.ptr,
             dt(out.dtype), out.ld, _p(part), RELU if relu else 0, dev, st)
    return part


def conv_first_norm(mode, x, w, bias, stats=None, gamma=None, beta=None, red=None, g=None, out=None):
    """The first convolution fused with the BatchNorm2d + ReLU behind it, its output recomputed instead of stored
    (segk_conv_first_norm): mode 1 -> statistics partials, 2 -> out = relu(norm(conv)), 3 -> partials of the norm
    backward's sums from g, 4 -> out = gradient of the convolution's output."""
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    x = x.contiguous().float()
    part = None
    if mode in (1, 3):
        part = torch.empty(query("segk_conv_first_stat_rows", N, H, W), 2, Cout, dtype=torch.float32, device=x.device)
    dev, st = _env(x)
    # only the pass that produces the layer's output counts as the convolution's forward in the roofline timer; the
    # others stand where BatchNorm's streaming passes stood
    with (_Timed("conv_fwd", 2.0 * N * H * W * Cout * 32, "first layer") if mode == 2 else contextlib.nullcontext()):
        call("segk_conv_first_norm", mode, x.data_ptr(), N, Cin, H, W, w.detach().contiguous().data_ptr(), _p(bias), Cout, _p(stats),
             _p(gamma), _p(beta), _p(red), g.ptr if g is not None else None, g.ld if g is not None else 0,
             out.ptr if out is not None else None, out.ld if out is not None else 0, _p(part), dev, st)
    return part


def conv_first_norm_bwd(x, w, bias, stats, gamma, beta, g, dw, dgamma, dbeta):
    """Backward of the fused first layer (conv_first_norm): weight gradient of the convolution and dgamma/dbeta of the norm
    from g = gradient of relu(norm(conv(x))), in two passes over g that recompute the convolution."""
    N, Cin, H, W = x.shape
    assert g.C == 64 and g.dtype == torch.bfloat16 and tuple(dw.shape) == (64, Cin, 3, 3) and dw.is_contiguous()
    x = x.contiguous().float()
    dev, st = _env(x)
    ws = workspace(query("segk_conv_first_norm_bwd_workspace", N, H, W), x.device)
    args = (x.data_ptr(), N, Cin, H, W, w.detach().contiguous().data_ptr(), _p(bias), stats.data_ptr(), _p(gamma), _p(beta), g.ptr, g.ld,
            dw.data_ptr(), _p(dgamma), _p(dbeta))
    # two calls so that the roofline timer brackets the convolution's half only (the first pass stands where BatchNorm's
    # backward passes stood); the sums travel in the workspace
    call("segk_conv_first_norm_bwd", *args, 4, ws.data_ptr(), ws.numel(), dev, st)
    with _Timed("conv_wgrad", 2.0 * N * H * W * 64 * 32):
        call("segk_conv_first_norm_bwd", *args, 8, ws.data_ptr(), ws.numel(), dev, st)
    return dw


def norm_bwd_sums_finalize(partials, red, dgamma=None, dbeta=None, accumulate=False):
    """Partial rows [rows][2][C] of the norm backward's two sums -> red [C][2] (+ dgamma, dbeta)."""
    rows, _, C = partials.shape
    dev, st = _env(partials)
    ws = workspace(query("segk_norm_stats_finalize_workspace", 1, rows, C), partials.device)
    call("segk_norm_bwd_sums_finalize", partials.data_ptr(), rows, C, red.data_ptr(), _p(dgamma), _p(dbeta), ACCUM if accumulate else 0,
         ws.data_ptr(), ws.numel(), dev, st)
    return red


def conv_first_wgrad(x, dy, dw, accumulate=False):
    """dw (fp32 [64, Cin, 3, 3]) = weight gradient of the first convolution from the fp32 NCHW image x and dy (bf16 Act)."""
    N, Cin, H, W = x.shape
    assert dy.C == 64 and dy.dtype == torch.bfloat16 and tuple(dw.shape) == (64, Cin, 3, 3) and dw.is_contiguous()
    x = x.contiguous().float()
    dev, st = _env(x)
    ws = workspace(query("segk_conv_first_wgrad_workspace", N, H, W), x.device)
    with _Timed("conv_wgrad", 2.0 * N * H * W * 64 * 32, "first layer"):
        call("segk_conv_first_wgrad", x.data_ptr(), N, Cin, H, W, dy.ptr, dt(dy.dtype), dy.ld, 64, dw.data_ptr(),
             ACCUM if accumulate else 0, ws.data_ptr(), ws.numel(), dev, st)
    return dw


def nchw_to_nhwc(x, dtype, C=None):
    N, Cin, H, W = x.shape
    C = C or (Cin + 7) // 8 * 8
    x = x.contiguous().float()
    out = Act.empty(N, H, W, C, dtype, x.device)
    dev, st = _env(x)
    call("segk_nchw_to_nhwc", x.data_ptr(), N, Cin, H, W, out.ptr, dt(dtype), C, out.ld, dev, st)
    return out


# ------------------------------------------------------------------------------------ convolution
def conv_pack(w, dtype, seg0=0, want_fwd=True, want_bwd=True):
    """fp32 [Cout,Cin,k,k] -> (forward image, data-gradient image) byte tensors."""
    Cout, Cin, kh, kw = w.shape
    taps = kh * kw
    assert taps in (1, 9) and w.dtype == torch.float32 and w.is_contiguous()
    nbytes = query("segk_conv_packed_bytes", dt(dtype), Cout, Cin, taps, seg0)
    wf = torch.empty(nbytes, dtype=torch.uint8, device=w.device) if want_fwd else None
    wb = torch.empty(nbytes, dtype=torch.uint8, device=w.device) if want_bwd else None
    dev, st = _env(w)
    call("segk_conv_pack", w.data_ptr(), Cout, Cin, taps, seg0, dt(dtype), _p(wf), _p(wb), dev, st)
    return wf, wb


def _segs(x):
    if isinstance(x, Act):
        return x, None
    a, b = x
    return a, b


def conv_stat_partials(out, fine=False):
    """Buffer for the per-wave output statistics of conv_fwd(..., stat_partials=...) writing `out`.
    fine: four rows per (image, 8x16 tile) whatever kernel runs, which lets the weights-in-LDS kernel serve the layer."""
    rows = query("segk_conv_stat_rows_fine", out.N, out.H, out.W) if fine else query("segk_conv_stat_rows", out.N, out.H, out.W, out.C)
    return torch.empty(rows, 2, out.C, dtype=torch.float32, device=out.device)


def norm_stats_from_partials(partials, groups, M, eps, stats, running_mean=None, running_var=None, momentum=0.1):
    rows, _, C = partials.shape
    dev, st = _env(partials)
    ws = workspace(query("segk_norm_stats_finalize_workspace", groups, rows // groups, C), partials.device)
    call("segk_norm_stats_finalize", partials.data_ptr(), groups, rows // groups, M, C, eps, stats.data_ptr(), _p(running_mean),
         _p(running_var), momentum, ws.data_ptr(), ws.numel(), dev, st)
    return stats


def conv_fwd(x, w_packed, Cout, out, taps=9, bias=None, relu=False, mask=None, accumulate=False, accumulate1=False,
             stat_partials=None):
    """out = conv(x) with the packed image `w_packed`; x is an Act or a pair of Acts (channel concat).
    mask: Act over the first mask.C output channels; those are zeroed where mask <= 0.
    out may be a pair of Acts: channels [0, out[0].C) go to the first, the rest to the second (accumulate1)."""
    a, b = _segs(x)
    o0, o1 = (out, None) if isinstance(out, Act) else out
    assert o0.C + (o1.C if o1 else 0) == Cout and (o0.N, o0.H, o0.W) == (a.N, a.H, a.W) and o0.dtype == a.dtype
    dev, st = _env(a.buf)
    flags = (RELU if relu else 0) | (ACCUM if accumulate else 0) | (4 if accumulate1 else 0)
    if stat_partials is not None and stat_partials.shape[0] == query("segk_conv_stat_rows_fine", a.N, a.H, a.W):
        flags |= 8                                     # four statistics rows per tile
    cin = a.C + (b.C if b else 0)
    with _Timed("conv_fwd", 2.0 * a.P * Cout * cin * taps, "128-wide" if (Cout + 31) // 32 * 32 % 128 == 0 else "narrow",
                f"{a.H}^2 {cin}->{Cout} k{taps}"):
        call("segk_conv_fwd", dt(a.dtype), taps, a.ptr, a.ld, a.C, b.ptr if b else None, b.ld if b else 0, b.C if b else 0,
             a.N, a.H, a.W, w_packed.data_ptr(), Cout, _p(bias), o0.ptr, o0.ld, o1.ptr if o1 else None, o1.ld if o1 else 0,
             o0.C if o1 else 0, mask.ptr if mask else None, mask.ld if mask else 0, mask.C if mask else 0, _p(stat_partials), flags,
             dev, st)
    return out


def conv_dgrad_norm(dy, w_packed, Cout, out, z, stats, gamma, beta, taps=9):
    """out = data gradient (conv of dy with the packed data-gradient image) = gradient of relu(batchnorm(z)); returns the
    per-wave partial rows of the two sums that batchnorm's backward needs (-> norm_bwd_finish)."""
    assert out.C == Cout == z.C and (out.N, out.H, out.W) == (dy.N, dy.H, dy.W) == (z.N, z.H, z.W) and out.dtype == dy.dtype == z.dtype
    dev, st = _env(dy.buf)
    partials = conv_stat_partials(out)
    with _Timed("conv_fwd", 2.0 * dy.P * Cout * dy.C * taps, "128-wide" if (Cout + 31) // 32 * 32 % 128 == 0 else "narrow"):
        call("segk_conv_dgrad_norm", dt(dy.dtype), taps, dy.ptr, dy.ld, dy.C, dy.N, dy.H, dy.W, w_packed.data_ptr(), Cout, out.ptr,
             out.ld, z.ptr, z.ld, stats.data_ptr(), _p(gamma), _p(beta), partials.data_ptr(), dev, st)
    return partials


def norm_bwd_finish(partials, dy, x, stats, gamma, beta, relu, dx, dgamma=None, dbeta=None, accumulate=False, add_dx=False):
    """norm_bwd for a BatchNorm whose sums were produced by conv_dgrad_norm."""
    rows = partials.shape[0]
    dev, st = _env(x.buf)
    ws = workspace(query("segk_norm_bwd_finish_workspace", rows, x.C), x.device)
    flags = (RELU if relu else 0) | (ACCUM if accumulate else 0) | (8 if add_dx else 0)
    call("segk_norm_bwd_finish", dt(x.dtype), partials.data_ptr(), rows, dy.ptr, dy.ld, x.ptr, x.ld, x.P, x.C, stats.data_ptr(),
         _p(gamma), _p(beta), flags, dx.ptr, dx.ld, _p(dgamma), _p(dbeta), ws.data_ptr(), ws.numel(), dev, st)
    return dx


def conv_wgrad(x, dy, dw, taps=9, cin_real=None, accumulate=False):
    """dw (fp32 [Cout, cin_real, taps]) = sum_pixels dy (x) shifted x."""
    a, b = _segs(x)
    Cout = dy.C
    cin = a.C + (b.C if b else 0)
    cin_real = cin_real or cin
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == Cout * cin_real * taps
    dev, st = _env(a.buf)
    d = dt(a.dtype)
    nbytes = query("segk_conv_wgrad_workspace", d, taps, a.C, b.C if b else 0, a.N, a.H, a.W, Cout)
    ws = workspace(nbytes, a.device)
    with _Timed("conv_wgrad", 2.0 * a.P * Cout * cin * taps, "", f"{a.H}^2 {cin}->{Cout} k{taps}"):
        call("segk_conv_wgrad", d, taps, a.ptr, a.ld, a.C, b.ptr if b else None, b.ld if b else 0, b.C if b else 0, a.N, a.H,
             a.W, dy.ptr, dy.ld, Cout, cin_real, dw.data_ptr(), ACCUM if accumulate else 0, ws.data_ptr(), ws.numel(), dev, st)
    return dw


def convt_pack(w, dtype, want_bwd=True):
    """fp32 ConvTranspose2d weight [Cin,Cout,4,4] -> (forward phase images, data-gradient phase images)."""
    Cin, Cout, kh, kw = w.shape
    assert (kh, kw) == (4, 4) and w.dtype == torch.float32 and w.is_contiguous()
    nbytes = query("segk_convt_packed_bytes", dt(dtype), Cin, Cout)
    wf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    wb = torch.empty(nbytes, dtype=torch.uint8, device=w.device) if want_bwd else None
    dev, st = _env(w)
    call("segk_convt_pack", w.data_ptr(), Cin, Cout, dt(dtype), _p(wf), _p(wb), dev, st)
    return wf, wb


def convt_fwd(x, wf, Cout, out, bias=None, relu=False):
    """out [N,2H,2W,Cout] = ConvTranspose2d(4, stride 2, padding 1)(x) (+bias, +ReLU)."""
    assert out.C == Cout and (out.N, out.H, out.W) == (x.N, 2 * x.H, 2 * x.W) and out.dtype == x.dtype
    dev, st = _env(x.buf)
    with _Timed("conv_fwd", 2.0 * x.P * x.C * Cout * 16, "transposed"):
        call("segk_convt_fwd", dt(x.dtype), x.ptr, x.ld, x.C, x.N, x.H, x.W, wf.data_ptr(), Cout, _p(bias), out.ptr, out.ld,
             RELU if relu else 0, dev, st)
    return out


def convt_dgrad(dy, wb, dx, mask=None, accumulate=False):
    """dx [N,H,W,Cin] from dy [N,2H,2W,Cout]; mask: forward activation whose ReLU gates dx."""
    assert (dy.N, dy.H, dy.W) == (dx.N, 2 * dx.H, 2 * dx.W)
    dev, st = _env(dy.buf)
    with _Timed("conv_fwd", 2.0 * dx.P * dx.C * dy.C * 16, "transposed"):
        call("segk_convt_dgrad", dt(dy.dtype), dy.ptr, dy.ld, dy.C, dx.N, dx.H, dx.W, wb.data_ptr(), dx.C, dx.ptr, dx.ld,
             mask.ptr if mask else None, mask.ld if mask else 0, ACCUM if accumulate else 0, dev, st)
    return dx


def convt_wgrad(x, dy, dw, accumulate=False):
    """dw fp32 [Cin,Cout,4,4]."""
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == x.C * dy.C * 16
    dev, st = _env(x.buf)
    d = dt(x.dtype)
    ws = workspace(query("segk_convt_wgrad_workspace", d, x.C, dy.C, x.N, x.H, x.W), x.device)
    with _Timed("conv_wgrad", 2.0 * x.P * x.C * dy.C * 16):
        call("segk_convt_wgrad", d, x.ptr, x.ld, x.C, x.N, x.H, x.W, dy.ptr, dy.ld, dy.C, dw.data_ptr(),
             ACCUM if accumulate else 0, ws.data_ptr(), ws.numel(), dev, st)
    return dw


def relu_bwd(dy, y, dx):
    dev, st = _env(dy.buf)
    call("segk_relu_bwd", dt(dy.dtype), dy.ptr, dy.ld, y.ptr, y.ld, dy.P, dy.C, dx.ptr, dx.ld, dev, st)
    return dx


def head_fwd(x, w, bias, logits):
    Cout = w.shape[0]
    dev, st = _env(x.buf)
    call("segk_head1x1_fwd", dt(x.dtype), x.ptr, x.ld, x.C, x.N, x.H * x.W, w.data_ptr(), _p(bias), Cout, logits.data_ptr(),
         dev, st)
    return logits


def head_bwd(x, w, dlogits, dx, relu_mask, dw, db):
    Cout = w.shape[0]
    dev, st = _env(x.buf)
    nbytes = query("segk_head1x1_bwd_workspace", x.C, Cout, x.N, x.H * x.W)
    ws = workspace(nbytes, x.device)
    call("segk_head1x1_bwd", dt(x.dtype), x.ptr, x.ld, x.C, x.N, x.H * x.W, w.data_ptr(), Cout, dlogits.data_ptr(), dx.ptr,
         dx.ld, 1 if relu_mask else 0, dw.data_ptr(), _p(db), ws.data_ptr(), ws.numel(), dev, st)


# ------------------------------------------------------------------------------------ normalisation
def norm_stats(x, groups, eps, stats, running_mean=None, running_var=None, momentum=0.1):
    M = x.P // groups
    dev, st = _env(x.buf)
    ws = workspace(query("segk_norm_workspace", groups, M, x.C), x.device)
    with _streamed("norm statistics", x.P * x.C * x.buf.element_size()):
        call("segk_norm_stats", dt(x.dtype), x.ptr, x.ld, groups, M, x.C, eps, stats.data_ptr(), _p(running_mean),
             _p(running_var), momentum, ws.data_ptr(), ws.numel(), dev, st)
    return stats


def norm_eval_stats(running_mean, running_var, eps, stats):
    dev, st = _env(running_mean)
    call("segk_norm_eval_stats", running_mean.data_ptr(), running_var.data_ptr(), running_mean.numel(), eps,
         stats.data_ptr(), dev, st)
    return stats


def norm_apply(x, groups, stats, gamma, beta, relu, out):
    M = x.P // groups
    dev, st = _env(x.buf)
    with _streamed("norm forward", 2 * x.P * x.C * x.buf.element_size()):
        call("segk_norm_apply", dt(x.dtype), x.ptr, x.ld, groups, M, x.C, stats.data_ptr(), _p(gamma), _p(beta),
             RELU if relu else 0, out.ptr, out.ld, dev, st)
    return out


def norm_bwd(dy, x, groups, stats, gamma, beta, relu, dx, dgamma=None, dbeta=None, dbias=None, accumulate=False, frozen=False,
             add_dx=False):
    """accumulate: dgamma/dbeta += ...; add_dx: dx += ... (x has other consumers that already wrote their share)."""
    M = x.P // groups
    dev, st = _env(x.buf)
    ws = workspace(query("segk_norm_workspace", groups, M, x.C), x.device)
    flags = (RELU if relu else 0) | (ACCUM if accumulate else 0) | (FROZEN if frozen else 0) | (8 if add_dx else 0)
    # two passes over (dy, x), one result written (read first when it is added to)
    with _streamed("norm backward", (6 if add_dx else 5) * x.P * x.C * x.buf.element_size()):
        call("segk_norm_bwd", dt(x.dtype), dy.ptr, dy.ld, x.ptr, x.ld, groups, M, x.C, stats.data_ptr(), _p(gamma), _p(beta),
             flags, dx.ptr, dx.ld, _p(dgamma), _p(dbeta), _p(dbias), ws.data_ptr(), ws.numel(), dev, st)
    return dx


def norm_pool_apply(x, stats, gamma, beta, out, pool):
    """out = relu(batchnorm(x)), pool = maxpool2x2(out) in one pass."""
    dev, st = _env(x.buf)
    with _streamed("norm forward", 2.25 * x.P * x.C * x.buf.element_size()):
        call("segk_norm_pool_apply", dt(x.dtype), x.ptr, x.ld, x.N, x.H, x.W, x.C, stats.data_ptr(), _p(gamma), _p(beta),
             out.ptr, out.ld, pool.ptr, pool.ld, dev, st)
    return out, pool


def norm_pool_bwd(dy, dpool, x, stats, gamma, beta, dx, dgamma=None, dbeta=None, accumulate=False, frozen=False):
    """Backward of norm_pool_apply: dy (gradient of `out` from its other consumers) may be None."""
    dev, st = _env(x.buf)
    ws = workspace(query("segk_norm_workspace", 1, x.P, x.C), x.device)
    flags = (ACCUM if accumulate else 0) | (FROZEN if frozen else 0)
    # two passes over (x, pooled gradient [, skip gradient]), one result written
    with _streamed("norm backward", ((5.5 if dy is not None else 3.5)) * x.P * x.C * x.buf.element_size()):
        call("segk_norm_pool_bwd", dt(x.dtype), dy.ptr if dy is not None else None, dy.ld if dy is not None else 0, dpool.ptr,
             dpool.ld, x.ptr, x.ld, x.N, x.H, x.W, x.C, stats.data_ptr(), _p(gamma), _p(beta), flags, dx.ptr, dx.ld, _p(dgamma),
             _p(dbeta), ws.data_ptr(), ws.numel(), dev, st)
    return dx


# ------------------------------------------------------------------------------------ resampling
def maxpool_fwd(x, out):
    dev, st = _env(x.buf)
    call("segk_maxpool2x2_fwd", dt(x.dtype), x.ptr, x.ld, x.N, x.H, x.W, x.C, out.ptr, out.ld, dev, st)
    return out


def maxpool_bwd(x, dy, dx, accumulate=False):
    dev, st = _env(x.buf)
    call("segk_maxpool2x2_bwd", dt(x.dtype), x.ptr, x.ld, dy.ptr, dy.ld, x.N, x.H, x.W, x.C, dx.ptr, dx.ld,
         ACCUM if accumulate else 0, dev, st)
    return dx


def upsample_fwd(x, out, stats=None, relu=False, align=False):
    dev, st = _env(x.buf)
    with _streamed("bilinear x2", 5 * x.P * x.C * x.buf.element_size()):
        call("segk_upsample2x_fwd", dt(x.dtype), x.ptr, x.ld, x.N, x.H, x.W, x.C, _p(stats),
             (RELU if relu else 0) | (ALIGN if align else 0), out.ptr, out.ld, dev, st)
    return out


def upsample_bwd(dy, y, dx, relu=False, align=False):
    """dx has the low-resolution shape; y is the saved forward output (only read when relu)."""
    dev, st = _env(dy.buf)
    with _streamed("bilinear x2", 5 * dx.P * dx.C * dx.buf.element_size()):
        call("segk_upsample2x_bwd", dt(dy.dtype), dy.ptr, dy.ld, y.ptr if y is not None else None, y.ld if y is not None else 0,
             dx.N, dx.H, dx.W, dx.C, (RELU if relu else 0) | (ALIGN if align else 0), dx.ptr, dx.ld, dev, st)
    return dx


def resize_fwd(x, out, scale, stats=None, relu=False):
    """Bilinear resize of x into out (its size fixes Ho, Wo) as nn.Upsample(scale_factor=scale, align_corners=False)."""
    dev, st = _env(x.buf)
    call("segk_resize_bilinear_fwd", dt(x.dtype), x.ptr, x.ld, x.N, x.H, x.W, x.C, out.H, out.W, 1.0 / scale, 1.0 / scale, _p(stats),
         RELU if relu else 0, out.ptr, out.ld, dev, st)
    return out


def resize_bwd(dy, dx, scale, accumulate=False):
    """dx (low- or high-resolution input shape) (+)= adjoint of resize_fwd applied to dy."""
    dev, st = _env(dy.buf)
    call("segk_resize_bilinear_bwd", dt(dy.dtype), dy.ptr, dy.ld, dx.N, dx.H, dx.W, dx.C, dy.H, dy.W, 1.0 / scale, 1.0 / scale,
         dx.ptr, dx.ld, ACCUM if accumulate else 0, dev, st)
    return dx


def colsum(x, out, accumulate=False):
    dev, st = _env(x.buf)
    ws = workspace(query("segk_colsum_workspace", x.P, x.C), x.device)
    call("segk_colsum", dt(x.dtype), x.ptr, x.ld, x.P, x.C, out.data_ptr(), ACCUM if accumulate else 0, ws.data_ptr(),
         ws.numel(), dev, st)
    return out


def copy_slice(src, dst, accumulate=False):
    assert src.C == dst.C and src.P == dst.P
    dev, st = _env(src.buf)
    call("segk_copy_slice", dt(src.dtype), src.ptr, src.ld, dst.ptr, dst.ld, src.P, src.C, ACCUM if accumulate else 0, dev, st)
    return dst


def zero_slice(dst):
    """Zero the channel slice `dst` of its buffer (16-byte stores per pixel; a strided torch fill runs at half the rate)."""
    dev, st = _env(dst.buf)
    call("segk_copy_slice", dt(dst.dtype), None, 0, dst.ptr, dst.ld, dst.P, dst.C, 0, dev, st)
    return dst


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    dev, st = _env(param)
    call("segk_adam_step", param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), param.numel(), lr,
         beta1, beta2, eps, weight_decay, step, grad_scale, dev, st)


# ------------------------------------------------------------------------------------ volumes
# A volume batch [N,C,D,H,W] is an Act over [N*D, H, W, ld]; D travels next to it.
def _i3(t):
    import ctypes
    return (ctypes.c_int * 3)(*[int(v) for v in t])


def conv3d_pack(w, dtype, want_fwd=True, want_bwd=True):
    """fp32 Conv3d weight [Cout,Cin,3,3,3] -> (forward images, data-gradient images), one 2-D image per depth tap."""
    Cout, Cin = w.shape[:2]
    assert tuple(w.shape[2:]) == (3, 3, 3) and w.dtype == torch.float32 and w.is_contiguous()
    nbytes = query("segk_conv3d_packed_bytes", dt(dtype), Cout, Cin)
    wf = torch.empty(nbytes, dtype=torch.uint8, device=w.device) if want_fwd else None
    wb = torch.empty(nbytes, dtype=torch.uint8, device=w.device) if want_bwd else None
    dev, st = _env(w)
    call("segk_conv3d_pack", w.data_ptr(), Cout, Cin, dt(dtype), _p(wf), _p(wb), dev, st)
    return wf, wb


def conv3d_fwd(x, D, w_packed, out, accumulate=False):
    """out = Conv3d(3, padding 1)(x) with the forward images, or dx from dy with the data-gradient images."""
    assert x.N % D == 0 and (out.N, out.H, out.W) == (x.N, x.H, x.W) and out.dtype == x.dtype
    dev, st = _env(x.buf)
    with _Timed("conv_fwd", 2.0 * x.P * x.C * out.C * 27, "128-wide" if (out.C + 31) // 32 * 32 % 128 == 0 else "narrow"):
        call("segk_conv3d_fwd", dt(x.dtype), x.ptr, x.ld, x.C, x.N // D, D, x.H, x.W, w_packed.data_ptr(), out.C, out.ptr, out.ld,
             ACCUM if accumulate else 0, dev, st)
    return out


def conv3d_wgrad(x, D, dy, dw, accumulate=False):
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == dy.C * x.C * 27
    dev, st = _env(x.buf)
    d = dt(x.dtype)
    ws = workspace(query("segk_conv3d_wgrad_workspace", d, x.C, x.N // D, D, x.H, x.W, dy.C), x.device)
    with _Timed("conv_wgrad", 2.0 * x.P * x.C * dy.C * 27):
        call("segk_conv3d_wgrad", d, x.ptr, x.ld, x.C, x.N // D, D, x.H, x.W, dy.ptr, dy.ld, dy.C, dw.data_ptr(),
             ACCUM if accumulate else 0, ws.data_ptr(), ws.numel(), dev, st)
    return dw


def convd_pack(w, dtype, want_fwd=True, want_bwd=True):
    """fp32 depth-convolution weight [Cout,Cin,kd] -> (forward images, data-gradient images), one 1x1 image per depth tap."""
    Cout, Cin, kd = w.shape
    assert w.dtype == torch.float32 and w.is_contiguous()
    nbytes = query("segk_convd_packed_bytes", dt(dtype), Cout, Cin, kd)
    wf = torch.empty(nbytes, dtype=torch.uint8, device=w.device) if want_fwd else None
    wb = torch.empty(nbytes, dtype=torch.uint8, device=w.device) if want_bwd else None
    dev, st = _env(w)
    call("segk_convd_pack", w.data_ptr(), Cout, Cin, kd, dt(dtype), _p(wf), _p(wb), dev, st)
    return wf, wb


def convd_fwd(x, D, kd, w_packed, out, accumulate=False, stat_partials=None):
    """out = Conv3d((kd,1,1), padding (kd/2,0,0))(x) with the forward images, or dx from dy with the data-gradient images."""
    assert x.N % D == 0 and (out.N, out.H, out.W) == (x.N, x.H, x.W) and out.dtype == x.dtype
    dev, st = _env(x.buf)
    with _Timed("conv_fwd", 2.0 * x.P * x.C * out.C * kd, "128-wide" if (out.C + 31) // 32 * 32 % 128 == 0 else "narrow"):
        call("segk_convd_fwd", dt(x.dtype), x.ptr, x.ld, x.C, x.N // D, D, x.H, x.W, kd, w_packed.data_ptr(), out.C, out.ptr, out.ld,
             _p(stat_partials), ACCUM if accumulate else 0, dev, st)
    return out


def convd_wgrad(x, D, kd, dy, dw, cin_real=None, accumulate=False):
    cin_real = cin_real or x.C
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == dy.C * cin_real * kd
    dev, st = _env(x.buf)
    d = dt(x.dtype)
    ws = workspace(query("segk_convd_wgrad_workspace", d, x.C, x.N // D, D, x.H, x.W, dy.C, kd), x.device)
    with _Timed("conv_wgrad", 2.0 * x.P * x.C * dy.C * kd):
        call("segk_convd_wgrad", d, x.ptr, x.ld, x.C, x.N // D, D, x.H, x.W, kd, dy.ptr, dy.ld, dy.C, cin_real, dw.data_ptr(),
             ACCUM if accumulate else 0, ws.data_ptr(), ws.numel(), dev, st)
    return dw


def block2(x, wide, backward=False, accumulate=False):
    """[N,H,W,C] <-> [N,H/2,W/2,4C] (2x2 pixel blocks as channels, tap-major); backward: x (+)= from wide."""
    assert wide.C == 4 * x.C and wide.N == x.N and 2 * wide.H == x.H and 2 * wide.W == x.W
    dev, st = _env(x.buf)
    call("segk_block2", dt(x.dtype), x.ptr, x.ld, wide.ptr, wide.ld, x.N, x.H, x.W, x.C,
         (1 if backward else 0) | (ACCUM if accumulate and backward else 0), dev, st)
    return x if backward else wide


def out_extent(size, kernel, stride, pad):
    return tuple((s + 2 * p - k) // st + 1 for s, k, st, p in zip(size, kernel, stride, pad))


def im2col3d(x, D, C, kernel, stride, pad):
    """Patch image of the first C channels of x: -> (Act [N*Do, Ho, Wo, Kp], Do)."""
    N = x.N // D
    Do, Ho, Wo = out_extent((D, x.H, x.W), kernel, stride, pad)
    K = C * kernel[0] * kernel[1] * kernel[2]
    Kp = (K + 7) // 8 * 8
    out = Act.empty(N * Do, Ho, Wo, Kp, x.dtype, x.device)
    dev, st = _env(x.buf)
    call("segk_im2col3d", dt(x.dtype), x.ptr, x.ld, C, N, D, x.H, x.W, _i3(kernel), _i3(stride), _i3(pad), out.ptr, out.ld, Kp, dev, st)
    return out, Do


def col2im3d(dpatches, dx, D, C, kernel, stride, pad, accumulate=False):
    """Adjoint of im2col3d: dx (first 8-channel group of an Act over the input volume) from the patch gradient."""
    dev, st = _env(dx.buf)
    call("segk_col2im3d", dt(dx.dtype), dpatches.ptr, dpatches.ld, C, dx.N // D, D, dx.H, dx.W, _i3(kernel), _i3(stride), _i3(pad),
         dx.ptr, dx.ld, ACCUM if accumulate else 0, dev, st)
    return dx


def maxpool3d_fwd(x, D, out, argmax):
    dev, st = _env(x.buf)
    call("segk_maxpool3d_fwd", dt(x.dtype), x.ptr, x.ld, x.N // D, D, x.H, x.W, x.C, out.ptr, out.ld, argmax.data_ptr(), dev, st)
    return out


def maxpool3d_bwd(dy, argmax, dx, D, accumulate=False):
    dev, st = _env(dx.buf)
    call("segk_maxpool3d_bwd", dt(dx.dtype), dy.ptr, dy.ld, argmax.data_ptr(), dx.N // D, D, dx.H, dx.W, dx.C, dx.ptr, dx.ld,
         ACCUM if accumulate else 0, dev, st)
    return dx


def avgpool3d_fwd(x, D, out):
    dev, st = _env(x.buf)
    call("segk_avgpool3d_fwd", dt(x.dtype), x.ptr, x.ld, x.N // D, D, x.H, x.W, x.C, out.ptr, out.ld, dev, st)
    return out


def avgpool3d_bwd(dy, dx, D, accumulate=False):
    dev, st = _env(dx.buf)
    call("segk_avgpool3d_bwd", dt(dx.dtype), dy.ptr, dy.ld, dx.N // D, D, dx.H, dx.W, dx.C, dx.ptr, dx.ld,
         ACCUM if accumulate else 0, dev, st)
    return dx


# ------------------------------------------------------------------------------------ classifier tail
def rows_mean_fwd(x, N):
    """fp32 [N, C] mean over the x.P // N pixels (voxels) of every sample of the slice-major Act x."""
    dev, st = _env(x.buf)
    out = torch.empty(N, x.C, dtype=torch.float32, device=x.device)
    call("segk_rows_mean_fwd", dt(x.dtype), x.ptr, x.ld, N, x.P // N, x.C, out.data_ptr(), dev, st)
    return out


def rows_mean_bwd(dout, dx):
    """Adjoint of rows_mean_fwd into the Act dx (overwrites)."""
    N = dout.shape[0]
    dev, st = _env(dx.buf)
    call("segk_rows_mean_bwd", dt(dx.dtype), dout.data_ptr(), N, dx.P // N, dx.C, dx.ptr, dx.ld, dev, st)
    return dx


def linear_fwd(x, w, bias):
    """x [N, K] @ w [O, K]^T + bias, fp32."""
    N, K = x.shape
    O = w.shape[0]
    assert x.dtype == w.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous() and w.shape[1] == K
    dev, st = _env(x)
    y = torch.empty(N, O, dtype=torch.float32, device=x.device)
    call("segk_linear_fwd", x.data_ptr(), w.data_ptr(), _p(bias), N, K, O, y.data_ptr(), dev, st)
    return y


def linear_bwd(x, w, dy, dw, dbias=None, want_dx=True, accumulate=False):
    """-> dx [N, K] (or None); dw [O, K] and dbias [O] are written (or added to)."""
    N, K = x.shape
    O = w.shape[0]
    assert dy.shape == (N, O) and dy.dtype == torch.float32 and dy.is_contiguous() and dw.is_contiguous()
    dev, st = _env(x)
    dx = torch.empty(N, K, dtype=torch.float32, device=x.device) if want_dx else None
    call("segk_linear_bwd", x.data_ptr(), w.data_ptr(), dy.data_ptr(), N, K, O, _p(dx), dw.data_ptr(), _p(dbias),
         ACCUM if accumulate else 0, dev, st)
    return dx

