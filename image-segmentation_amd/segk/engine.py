"""Layer-granular tape that runs a network's forward and backward as explicit kernel launches.

A model's `forward` describes itself once per call through the helpers below (conv, bn_relu, pool,
in_up_relu, ...).  Each helper launches the forward kernels immediately and, when gradients are
needed, records a closure that launches the matching backward kernels; `Tape.backward` replays the
closures in reverse.  There is no graph compiler: the tape is the schedule.

Gradient buffers mirror activation buffers (same NHWC shape), so a tensor that lives in a channel
slice of a concat buffer receives its gradient in the same slice of the mirrored buffer and the
data-gradient of the convolution that reads the whole concat buffer fills all slices at once.
"""
import contextlib
import os

import torch

from . import ops
from .ops import Act


_NO_FIRST_DIRECT = bool(os.environ.get("SEGK_NO_FIRST_DIRECT"))   # experiments: first convolution as a 1x1 convolution over a patch image
_NO_FIRST_NORM_FUSE = bool(os.environ.get("SEGK_NO_FIRST_NORM_FUSE"))   # experiments: first convolution's output stored, BatchNorm as separate passes
_NO_CONV_STATS = bool(os.environ.get("SEGK_NO_CONV_STATS"))   # experiments: separate statistics pass after every convolution
# Weight gradients on a second stream: a convolution's weight and data gradients only share their input dy, so the two
# kernels can run side by side and fill each other's tails and the under-occupied small layers (gradients are bitwise
# the same either way).  On by default with one rank; SEGK_WGRAD_STREAM=0 turns it off.
_WGRAD_STREAM = os.environ.get("SEGK_WGRAD_STREAM", "1") == "1"
# With more than one rank the side stream is used as well (SEGK_WGRAD_STREAM_DP=0 keeps the weight gradients on the main stream
# there): a bucket's all-reduce waits for BOTH streams (segk/flat.py), and a one-rank RCCL group with forced collectives runs
# exactly this three-stream schedule with bitwise unchanged gradients (tests/rccl_worker.py).  One setting for every rank count
# also keeps the N = 1 point of a scaling curve like for like with the others.
_WGRAD_STREAM_DP = os.environ.get("SEGK_WGRAD_STREAM_DP", "1") == "1"
_side_streams = {}


def _data_parallel():
    """More than one rank (bucket all-reduces are then launched from inside the backward pass)."""
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def side_stream(device, create=True):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _side_streams and create:
        _side_streams[key] = torch.cuda.Stream(device)
    return _side_streams.get(key)


# Opt-in: the data gradient of the convolution behind a BatchNorm + ReLU also reduces the two sums of that norm's backward
# (one pass over the gradient and the norm's input less).  Measured on UNet-VGG, batch 32: -0.5 ms of norm passes, +0.45 ms of
# convolution epilogues per step, so it stays off by default.
_NORM_FUSE = os.environ.get("SEGK_NORM_FUSE", "0") == "1"


class Var:
    """An activation view plus the bookkeeping of its gradient."""
    __slots__ = ("act", "_ready", "writes", "relu_fused", "stat_part", "norm", "norm_part")

    def __init__(self, act, relu_fused=False):
        self.act = act
        self.stat_part = None         # per-wave channel sums written by the producing convolution (for the norm that follows)
        self._ready = False
        self.writes = 0               # gradient contributions received so far
        self.relu_fused = relu_fused  # act = relu(conv): the producer's ReLU backward is applied by the consumer
        # act = relu(batchnorm(z)) read by exactly one convolution: (z, stats, gamma, beta).  That convolution's data
        # gradient then also reduces the sums of the norm's backward (norm_part: the partial rows it left).
        self.norm = None
        self.norm_part = None

    @property
    def ready(self):
        """The gradient slice has been written at least once."""
        return self._ready

    @ready.setter
    def ready(self, value):
        self._ready = value
        if value:
            self.writes += 1


class Runtime:
    def __init__(self, params, buffers, dtype, device, training, need_grad, grad_sink=None, grad_alloc=None, zero_state=None):
        self.p, self.b = params, buffers        # name -> tensor
        self.dtype, self.device = dtype, device
        self.training, self.need_grad = training, need_grad
        self.tape = []
        self.grads = {}
        self._gradbufs = {}
        self._keep = []                         # keeps activation buffers alive until backward
        self.grad_sink = grad_sink              # optional callback(name, tensor) fired as soon as a gradient is final
        self.grad_alloc = grad_alloc            # optional callback(name, like) -> tensor to write the gradient into
        # names whose persistent gradient tensor is known to hold zeros already (only meaningful with grad_alloc, whose
        # tensors outlive the step): an analytically-zero gradient is then not rewritten every step
        self.zero_state = zero_state if grad_alloc is not None else None
        self.counters = []                      # BatchNorm num_batches_tracked buffers to bump once, together
        self.pack_plan = None                   # segk.packplan.PackPlan of the network (None: pack layer by layer)
        self.exports = []                       # intermediate tensors handed to the caller next to the logits
        self._packed = set()
        self._uses, self._shared, self._left = {}, {}, {}   # parameters used by several layers (models/unetvgg.py:1090-1093)

    # ---------------------------------------------------------------- buffers
    def new(self, N, H, W, C, zero=False):
        return Act.empty(N, H, W, C, self.dtype, self.device, zero=zero)

    def grad_of(self, act):
        """Gradient view mirroring `act` (allocated per underlying buffer on first use)."""
        key = id(act.buf)
        g = self._gradbufs.get(key)
        if g is None:
            g = torch.empty_like(act.buf)
            self._gradbufs[key] = g
            self._keep.append(act.buf)
        return Act(g, act.c0, act.C)

    def alloc_grad(self, name, like):
        """fp32 tensor that receives the gradient of parameter `name` (a view into a flat all-reduce
        bucket when a data-parallel wrapper installed `grad_alloc`)."""
        if self.grad_alloc is not None:
            return self.grad_alloc(name, like)
        return torch.empty_like(like, dtype=torch.float32)

    def _zero_grad(self, name, tensor):
        """Write an analytically zero gradient (skipped when the persistent tensor already holds zeros)."""
        if self.zero_state is not None:
            if name in self.zero_state:
                return
            self.zero_state.add(name)
        tensor.zero_()

    def finish_forward(self, owner=None):
        """Bump the BatchNorm step counters of this forward.  With an owning module they are re-homed once as views of
        ONE int64 tensor (kept on the module), so a step bumps all of them with a single launch."""
        cs, self.counters = self.counters, []
        if not cs:
            return
        if owner is None or len(cs) < 4:
            torch._foreach_add_(cs, 1)
            return
        flat = getattr(owner, "_segk_counters", None)
        base = flat.data_ptr() if flat is not None else 0
        if flat is None or flat.numel() != len(cs) or any(c.data_ptr() != base + 8 * i for i, c in enumerate(cs)):
            flat = torch.stack([c.detach() for c in cs])
            for i, c in enumerate(cs):
                c.data = flat[i]
            owner._segk_counters = flat
        flat.add_(1)

    def _emit(self, name, tensor):
        self.grads[name] = tensor
        if self.grad_sink is not None:
            self.grad_sink(name, tensor)

    def record(self, fn):
        if self.need_grad:
            self.tape.append(fn)

    def backward_keep(self):
        for fn in reversed(self.tape):
            fn()
        self.tape = []
        self.join_side()

    @contextlib.contextmanager
    def side(self):
        """Run the enclosed launches on the side stream, after everything enqueued so far on the current stream."""
        if not _WGRAD_STREAM or self.device.type != "cuda" or (_data_parallel() and not _WGRAD_STREAM_DP):
            yield
            return
        s = side_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))     # the tape's device, which need not be the thread's current one
        s.wait_event(ev)
        with torch.cuda.stream(s):
            yield
        self._side_used = True

    def join_side(self):
        """The current stream waits for the side stream (before gradients are consumed or buffers are released)."""
        if getattr(self, "_side_used", False):
            ev = torch.cuda.Event()
            ev.record(side_stream(self.device))
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._side_used = False

    def release(self):
        self._gradbufs = {}
        self._keep = []

    def backward(self):
        self.backward_keep()
        self.release()

    # ---------------------------------------------------------------- extra outputs
    def export(self, var):
        """fp32 NCHW copy of an intermediate tensor for networks that return feature maps next to their logits
        (models/unetvgg.py:574,646,726); its gradient re-enters the tape through `import_gradients`."""
        self.exports.append(var)
        return var.act.nchw().float().contiguous()

    def import_gradients(self, grads):
        """Add the callers' gradients of the exported tensors (None: unused) to the tape's gradient buffers.  Runs after
        the head's backward and before the recorded closures."""
        for var, g in zip(self.exports, grads):
            if isinstance(var, tuple):           # a second head: its backward seeds the gradient of its own branch
                self.head_backward(g if g is not None else torch.zeros_like(var[4]), record=var)
                continue
            if g is None:
                continue
            tmp = ops.nchw_to_nhwc(g, self.dtype, C=var.act.C)
            if var.relu_fused:                   # the buffer holds d/d(pre-ReLU): mask with the saved output
                ops.relu_bwd(tmp, var.act, tmp)
            ops.copy_slice(tmp, self.grad_of(var.act), accumulate=var.ready)
            var.ready = True

    def input(self, x):
        """NCHW tensor -> NHWC activation Var (layout conversion by a torch copy: boundary plumbing)."""
        return Var(Act.from_nchw(x.detach(), self.dtype))

    # ---------------------------------------------------------------- packed weights
    def _pack(self, wname, seg0=0, view=None):
        """Packed images of parameter `wname` (view: shape to read it as, e.g. [Cout,Cin,1,1] for a 1x1x1 Conv3d weight).
        With a replayed PackPlan the images were already written by this forward's single batched launch."""
        key = (wname, seg0)
        plan = self.pack_plan
        if plan is not None and key in plan.entries and (plan.ready or key in self._packed):
            return plan.entries[key]              # replayed plan, or a second layer using the same weight in this forward
        self._packed.add(key)
        w = self.p[wname].detach()
        if view is not None:
            w = w.reshape(view)
        if w.dim() == 5:
            wf, wb = ops.conv3d_pack(w, self.dtype, want_bwd=self.need_grad)
        else:
            wf, wb = ops.conv_pack(w, self.dtype, seg0=seg0, want_fwd=True, want_bwd=self.need_grad)
        if plan is not None and not plan.ready:
            plan.record(key, wname, w, self.dtype, wf, wb, seg0=seg0)
        return wf, wb

    # ---------------------------------------------------------------- layers
    def _stat_buffer(self, y, stats, norm_groups=None, cin=None):
        """Partial-sum buffer for the convolution producing `y` when a normalisation with batch (training) or
        instance (always) statistics follows.  cin: input channels of a 3x3 convolution; thin layers (the ones the
        weights-in-LDS kernel serves) get one row per pair of pixel rows."""
        if not stats or (norm_groups is None and not self.training) or _NO_CONV_STATS:
            return None
        fine = cin is not None and self.dtype == torch.bfloat16 and cin <= 128 and y.act.C <= 128
        y.stat_part = ops.conv_stat_partials(y.act, fine=fine)
        return y.stat_part

    def _norm_stats(self, y, groups, eps, stats, rmean=None, rvar=None, momentum=0.1):
        if y.stat_part is not None:
            ops.norm_stats_from_partials(y.stat_part, groups, y.act.P // groups, eps, stats, rmean, rvar, momentum)
            y.stat_part = None
        else:
            ops.norm_stats(y.act, groups, eps, stats, rmean, rvar, momentum)

    def conv_first(self, x, wname, bname, bias_grad_zero=False, stats=False):
        """3x3 convolution of a few-channel NCHW fp32 image, run as a 1x1 convolution over its 3x3 patch
        image (one K-row of 9*Cin gathered values).  The output carries Cout rounded up to 8 channels
        (extra ones are exactly zero).  No data gradient: the image does not require one."""
        w = self.p[wname]
        Cout, Cin = w.shape[0], w.shape[1]
        cs = (Cout + 7) // 8 * 8
        N, _, H, W = x.shape
        y = Var(self.new(N, H, W, cs))
        direct = ops.conv_first_direct_ok(x, Cout, self.dtype) and not _NO_FIRST_DIRECT
        if direct:
            # straight from the image: no patch image in the forward pass (the weight gradient gathers it when it runs)
            want = bool(stats) and (stats == "instance" or self.training) and not _NO_CONV_STATS
            y.stat_part = ops.conv_first_fwd(x, w, self._bias(bname), y.act, want_stats=want)
            patches = None
        else:
            patches = ops.im2col3x3(x, self.dtype)
            w1 = torch.zeros(cs, 9 * Cin, 1, 1, dtype=torch.float32, device=self.device)
            w1[:Cout] = w.detach().permute(0, 2, 3, 1).reshape(Cout, 9 * Cin, 1, 1)
            wf, _ = ops.conv_pack(w1, self.dtype, want_bwd=False)
            ops.conv_fwd(patches, wf, cs, y.act, taps=1, bias=self._bias(bname, cs),
                         stat_partials=self._stat_buffer(y, stats, norm_groups=N if stats == "instance" else None))

        def bwd():
            nonlocal patches
            dy = self.grad_of(y.act)
            g = self.alloc_grad(wname, w)
            if direct and Cout == 64 and g.is_contiguous():
                ops.conv_first_wgrad(x, dy, g)               # straight from the image into the gradient tensor
            else:
                if patches is None:
                    patches = ops.im2col3x3(x, self.dtype)
                dw = torch.empty(cs, 9 * Cin, dtype=torch.float32, device=self.device)
                ops.conv_wgrad(patches, dy, dw, taps=1, cin_real=9 * Cin)
                g.copy_(dw[:Cout].view(Cout, 3, 3, Cin).permute(0, 3, 1, 2))
            self._emit(wname, g)
            if bname is not None:
                gb = self.alloc_grad(bname, self.p[bname])
                if bias_grad_zero:
                    self._zero_grad(bname, gb)
                else:
                    if self.zero_state is not None:
                        self.zero_state.discard(bname)
                    gb.copy_(ops.colsum(dy, torch.empty(cs, dtype=torch.float32, device=self.device))[:Cout])
                self._emit(bname, gb)
        self.record(bwd)
        return y

    def can_fuse_first_norm(self, x, Cout):
        """Whether conv_first_bn_relu serves this first layer: the bf16 path on an image of up to three channels, 64 output
        channels, batch statistics (training mode)."""
        return (self.training and Cout == 64 and ops.conv_first_direct_ok(x, Cout, self.dtype) and not _NO_FIRST_DIRECT
                and not _NO_FIRST_NORM_FUSE)

    def conv_first_bn_relu(self, x, wname, bname, prefix, eps=1e-5, momentum=0.1):
        """Conv2d(Cin <= 3, 64, 3, padding=1) -> BatchNorm2d (batch statistics) -> ReLU on the fp32 NCHW image
        (models/unetvgg.py:72 features[0..2]; models/unetplusplus.py:11-12 of conv0_0).  The convolution's 64-channel output is
        never stored: statistics, the normalised output, the backward's two sums and the gradient that the weight gradient
        reads are each computed from the image again (a few matrix steps per tile), which costs far less than moving a
        64-channel tensor through HBM four more times."""
        w = self.p[wname]
        Cout = w.shape[0]
        N, _, H, W = x.shape
        bias = self._bias(bname)
        gamma, beta = self.p[prefix + ".weight"].detach(), self.p[prefix + ".bias"].detach()
        stats = torch.empty(2 * Cout, dtype=torch.float32, device=self.device)
        part = ops.conv_first_norm(1, x, w, bias)
        ops.norm_stats_from_partials(part, 1, N * H * W, eps, stats, self.b[prefix + ".running_mean"], self.b[prefix + ".running_var"], momentum)
        self.counters.append(self.b[prefix + ".num_batches_tracked"])
        a = Var(self.new(N, H, W, Cout))
        ops.conv_first_norm(2, x, w, bias, stats=stats, gamma=gamma, beta=beta, out=a.act)

        def bwd():
            da = self.grad_of(a.act)
            dg, db = self.alloc_grad(prefix + ".weight", gamma), self.alloc_grad(prefix + ".bias", beta)
            gw = self.alloc_grad(wname, w)
            # two passes over da: the norm's sums, then dL/dz formed inside the weight gradient's LDS stage
            if gw.is_contiguous():
                ops.conv_first_norm_bwd(x, w, bias, stats, gamma, beta, da, gw, dg, db)
            else:
                tmp = torch.empty_like(w, dtype=torch.float32)
                ops.conv_first_norm_bwd(x, w, bias, stats, gamma, beta, da, tmp, dg, db)
                gw.copy_(tmp)
            self._emit(wname, gw)
            if bname is not None:                 # the gradient reaching a bias in front of batch statistics sums to zero
                gb = self.alloc_grad(bname, self.p[bname])
                self._zero_grad(bname, gb)
                self._emit(bname, gb)
            self._emit(prefix + ".weight", dg)
            self._emit(prefix + ".bias", db)
        self.record(bwd)
        return a

    def _bias(self, bname, cs=None):
        if bname is None:
            return None
        b = self.p[bname].detach()
        if cs is not None and cs != b.numel():
            pb = torch.zeros(cs, dtype=torch.float32, device=self.device)
            pb[:b.numel()] = b
            return pb
        return b

    def conv(self, x, wname, bname=None, relu=False, out=None, bias_grad_zero=False, seg_vars=None, cin_real=None, stats=False):
        """3x3 convolution.  x: Var, or a list of Vars that tile one buffer contiguously (in-place concat),
        or a pair (list_of_vars, Var) for a two-segment concat.  Returns the output Var.
        stats: a BatchNorm ("batch") or InstanceNorm ("instance") follows -- the convolution's epilogue also produces
        the per-channel partial sums, so the normalisation needs no statistics pass over the output."""
        w = self.p[wname]
        Cout, Cin = w.shape[0], w.shape[1]
        segs = self._segments(x)
        seg0 = segs[0][0].C if len(segs) == 2 else 0
        wf, wb = self._pack(wname, seg0=seg0)
        a0 = segs[0][0]
        N, H, W = a0.N, a0.H, a0.W
        y = Var(out or self.new(N, H, W, Cout), relu_fused=relu)
        xin = a0 if len(segs) == 1 else (segs[0][0], segs[1][0])
        ops.conv_fwd(xin, wf, Cout, y.act, taps=9, bias=self._bias(bname), relu=relu,
                     stat_partials=self._stat_buffer(y, stats, norm_groups=y.act.N if stats == "instance" else None, cin=Cin))

        for name in (wname, bname):
            if name is not None:
                self._uses[name] = self._uses.get(name, 0) + 1

        def bwd():
            dy = self.grad_of(y.act)
            dw, more_w = self._param_grad(wname, w)
            db, more_b = self._param_grad(bname, self.p[bname]) if bname is not None else (None, False)
            with self.side():
                ops.conv_wgrad(xin, dy, dw, taps=9, cin_real=Cin, accumulate=more_w)
                self._param_done(wname, dw)
                if bname is not None:
                    if bias_grad_zero:
                        assert not more_b
                        self._zero_grad(bname, db)
                    else:
                        if self.zero_state is not None:
                            self.zero_state.discard(bname)
                        ops.colsum(dy, db, accumulate=more_b)
                    self._param_done(bname, db)
            self._dgrad(segs, dy, wb, Cin)
        self.record(bwd)
        return y

    def _param_grad(self, name, like):
        """-> (gradient tensor of parameter `name`, whether an earlier layer of this backward already wrote into it)."""
        if name in self._shared:
            return self._shared[name], True
        g = self.alloc_grad(name, like)
        if self._uses.get(name, 1) > 1:
            self._shared[name] = g
        return g, False

    def _param_done(self, name, g):
        """Publish the gradient once the last layer that uses the parameter has added its share."""
        left = self._left.get(name, self._uses.get(name, 1)) - 1
        self._left[name] = left
        if left == 0:
            self._emit(name, g)

    def copy(self, a, out):
        """Second copy of a tensor in another concat buffer (a skip tensor read by two decoder branches)."""
        assert not a.relu_fused
        v = Var(out)
        ops.copy_slice(a.act, out)

        def bwd():
            ops.copy_slice(self.grad_of(out), self.grad_of(a.act), accumulate=a.ready)
            a.ready = True
        self.record(bwd)
        return v

    def _segments(self, x):
        """-> list of (covering Act, [member Vars])."""
        def one(v):
            if isinstance(v, Var):
                return (v.act, [v])
            members = list(v)
            a = members[0].act
            c = a.c0
            for m in members:
                assert m.act.buf is a.buf and m.act.c0 == c, "concat members must tile one buffer in order"
                c += m.act.C
            return (Act(a.buf, a.c0, c - a.c0), members)
        if isinstance(x, tuple):
            return [one(x[0]), one(x[1])]
        return [one(x)]

    def _dgrad(self, segs, dy, wb, Cin, taps=9):
        """Data gradient of a convolution into the gradient slices of its input segments."""
        def mask_of(members):
            """Act covering the leading members whose ReLU backward is left to this consumer (or None)."""
            k = 0
            while k < len(members) and members[k].relu_fused:
                k += 1
            assert not any(m.relu_fused for m in members[k:]), "fused-ReLU tensors must lead their concat segment"
            if k == 0:
                return None
            a = members[0].act
            return Act(a.buf, a.c0, sum(m.act.C for m in members[:k]))

        def state(members):
            r = [m.ready for m in members]
            if any(r) and not all(r):
                # some members already hold a gradient (an exported tensor's, from the caller): zero the others and add
                assert self.exports, "mixed gradient readiness inside one concat segment"
                for m in members:
                    if not m.ready:
                        ops.zero_slice(self.grad_of(m.act))
                        m.ready = True
                return True
            return r[0]

        if len(segs) == 1:
            act, members = segs[0]
            v = members[0]
            if len(members) == 1 and v.norm is not None and not v.ready and taps == 9:
                z, stats, gamma, beta = v.norm
                v.norm_part = ops.conv_dgrad_norm(dy, wb, act.C, self.grad_of(act), z, stats, gamma, beta, taps=taps)
                v.ready = True
                return
            ops.conv_fwd(dy, wb, act.C, self.grad_of(act), taps=taps, mask=mask_of(members), accumulate=state(members))
            for m in members:
                m.ready = True
            return
        # two segments: the kernel writes each segment's gradient straight into its own buffer
        (act0, mem0), (act1, mem1) = segs
        assert mask_of(mem1) is None, "a tensor whose ReLU backward is deferred must be (in) the first segment"
        ops.conv_fwd(dy, wb, Cin, (self.grad_of(act0), self.grad_of(act1)), taps=taps, mask=mask_of(mem0), accumulate=state(mem0),
                     accumulate1=state(mem1))
        for m in mem0 + mem1:
            m.ready = True

    def conv_down2(self, x, wname, bname=None):
        """Conv2d(kernel_size=2, stride=2) (models/unetvgg.py:305): the 2x2 pixel blocks are gathered into channels and
        a 1x1 convolution on the matrix cores does the arithmetic."""
        assert not x.relu_fused, "the block gather cannot consume a tensor whose ReLU backward is deferred"
        w = self.p[wname]
        Cout, Cin = w.shape[0], w.shape[1]
        a = x.act
        assert a.C == Cin and tuple(w.shape[2:]) == (2, 2)
        wide = self.new(a.N, a.H // 2, a.W // 2, 4 * Cin)
        ops.block2(a, wide)
        w1 = w.detach().permute(0, 2, 3, 1).reshape(Cout, 4 * Cin, 1, 1).contiguous()      # [co][(b, e, c)]
        wf, wb = ops.conv_pack(w1, self.dtype, want_fwd=True, want_bwd=self.need_grad)
        y = Var(self.new(a.N, a.H // 2, a.W // 2, Cout))
        ops.conv_fwd(wide, wf, Cout, y.act, taps=1, bias=self._bias(bname))

        def bwd():
            dy = self.grad_of(y.act)
            dw = torch.empty(Cout, 4 * Cin, dtype=torch.float32, device=self.device)
            ops.conv_wgrad(wide, dy, dw, taps=1)
            g = self.alloc_grad(wname, w)
            g.copy_(dw.view(Cout, 2, 2, Cin).permute(0, 3, 1, 2))
            self._emit(wname, g)
            if bname is not None:
                db = self.alloc_grad(bname, self.p[bname])
                if self.zero_state is not None:
                    self.zero_state.discard(bname)
                ops.colsum(dy, db)
                self._emit(bname, db)
            dwide = self.new(a.N, a.H // 2, a.W // 2, 4 * Cin)
            ops.conv_fwd(dy, wb, 4 * Cin, dwide, taps=1)
            ops.block2(self.grad_of(a), dwide, backward=True, accumulate=x.ready)
            x.ready = True
        self.record(bwd)
        return y

    def conv_transpose_relu(self, h, wname, bname, out=None):
        """ConvTranspose2d(4, stride 2, padding 1) + ReLU (models/unetvgg.py:31-32,41)."""
        w = self.p[wname]
        Cin, Cout = w.shape[0], w.shape[1]
        x = h.act
        wf, wb = ops.convt_pack(w.detach(), self.dtype, want_bwd=self.need_grad)
        u = Var(out or self.new(x.N, 2 * x.H, 2 * x.W, Cout), relu_fused=True)   # consumer applies the ReLU backward
        ops.convt_fwd(x, wf, Cout, u.act, bias=self._bias(bname), relu=True)

        def bwd():
            dpre = self.grad_of(u.act)
            dw = self.alloc_grad(wname, w)
            ops.convt_wgrad(x, dpre, dw)
            self._emit(wname, dw)
            if bname is not None:
                db = self.alloc_grad(bname, self.p[bname])
                ops.colsum(dpre, db)
                self._emit(bname, db)
            ops.convt_dgrad(dpre, wb, self.grad_of(x), mask=x if h.relu_fused else None, accumulate=h.ready)
            h.ready = True
        self.record(bwd)
        return u

    def bn_relu(self, y, prefix, out=None, eps=1e-5, momentum=0.1, pool=False, one_reader=False):
        """BatchNorm2d (+ReLU).  Training: batch statistics and running-buffer update; eval: running buffers.
        pool=True also returns the 2x2 max pool of the result, computed (and back-propagated) in the same passes.
        one_reader=True: the caller guarantees that exactly one 3x3 convolution reads the result and nothing else does;
        that convolution's data gradient then reduces the two sums of this norm's backward in its epilogue."""
        C = y.act.C
        gamma, beta = self.p[prefix + ".weight"].detach(), self.p[prefix + ".bias"].detach()
        stats = torch.empty(2 * C, dtype=torch.float32, device=self.device)
        if self.training:
            self._norm_stats(y, 1, eps, stats, self.b[prefix + ".running_mean"], self.b[prefix + ".running_var"], momentum)
            self.counters.append(self.b[prefix + ".num_batches_tracked"])
        else:
            ops.norm_eval_stats(self.b[prefix + ".running_mean"], self.b[prefix + ".running_var"], eps, stats)
        a = Var(out or self.new(y.act.N, y.act.H, y.act.W, C))
        p = None
        if pool:
            p = Var(self.new(y.act.N, y.act.H // 2, y.act.W // 2, C))
            ops.norm_pool_apply(y.act, stats, gamma, beta, a.act, p.act)
        else:
            ops.norm_apply(y.act, 1, stats, gamma, beta, True, a.act)
        frozen = not self.training
        if one_reader and not pool and not frozen and self.need_grad and _NORM_FUSE:
            a.norm = (y.act, stats, gamma, beta)

        def bwd():
            dy = self.grad_of(y.act)
            dg, db = self.alloc_grad(prefix + ".weight", gamma), self.alloc_grad(prefix + ".bias", beta)
            if pool:
                assert p.ready, "the pooled tensor received no gradient"
                da = self.grad_of(a.act) if a.ready else None
                ops.norm_pool_bwd(da, self.grad_of(p.act), y.act, stats, gamma, beta, dy, dgamma=dg, dbeta=db, frozen=frozen)
            elif a.norm_part is not None:
                assert a.writes == 1, "a tensor declared one_reader received several gradient contributions"
                ops.norm_bwd_finish(a.norm_part, self.grad_of(a.act), y.act, stats, gamma, beta, True, dy, dgamma=dg, dbeta=db, add_dx=y.ready)
                a.norm_part = None
            else:
                ops.norm_bwd(self.grad_of(a.act), y.act, 1, stats, gamma, beta, True, dy, dgamma=dg, dbeta=db, frozen=frozen, add_dx=y.ready)
            y.ready = True
            self._emit(prefix + ".weight", dg)
            self._emit(prefix + ".bias", db)
        self.record(bwd)
        return (a, p) if pool else a

    def pool(self, a, out=None):
        assert not a.relu_fused, "max pool cannot consume a tensor whose ReLU backward is deferred"
        x = a.act
        p = Var(out or self.new(x.N, x.H // 2, x.W // 2, x.C))
        ops.maxpool_fwd(x, p.act)

        def bwd():
            ops.maxpool_bwd(x, self.grad_of(p.act), self.grad_of(x), accumulate=a.ready)
            a.ready = True
        self.record(bwd)
        return p

    def in_up_relu(self, y, out=None, eps=1e-5, upsample=True, relu=True, scale=2):
        """InstanceNorm2d(affine=False) -> bilinear resize by `scale` (align_corners=False) -> ReLU, one pass over the
        output (scale 2: the specialised kernels; 8, 4, 0.5 ...: the general ones).
        upsample=False: InstanceNorm2d (-> ReLU unless relu=False) at the input's size."""
        assert relu or not upsample
        x = y.act
        stats = torch.empty(2 * x.N * x.C, dtype=torch.float32, device=self.device)
        self._norm_stats(y, x.N, eps, stats)
        if upsample:
            # u = relu(...): the ReLU backward is applied by the consumer's data gradient (relu_fused)
            u = Var(out or self.new(x.N, int(x.H * scale), int(x.W * scale), x.C), relu_fused=True)
            if scale == 2:
                ops.upsample_fwd(x, u.act, stats=stats, relu=True, align=False)
            else:
                ops.resize_fwd(x, u.act, scale, stats=stats, relu=True)
        else:
            u = Var(out or self.new(x.N, x.H, x.W, x.C))
            ops.norm_apply(x, x.N, stats, None, None, relu, u.act)

        def bwd():
            du = self.grad_of(u.act)
            dy = self.grad_of(x)
            if upsample:
                dxh = self.new(x.N, x.H, x.W, x.C)
                if scale == 2:
                    ops.upsample_bwd(du, None, dxh, relu=False, align=False)
                else:
                    ops.resize_bwd(du, dxh, scale)
                ops.norm_bwd(dxh, x, x.N, stats, None, None, False, dy)
            else:
                ops.norm_bwd(du, x, x.N, stats, None, None, relu, dy)
            y.ready = True
        self.record(bwd)
        return u

    def up(self, a, out=None, align=True, relu=False):
        """Plain bilinear x2 (models/unetplusplus.py:36); relu=True: followed by a ReLU (models/unetvgg.py:370-371) whose
        backward is left to the consumer's data gradient."""
        assert not a.relu_fused, "up-sampling cannot consume a tensor whose ReLU backward is deferred"
        x = a.act
        u = Var(out or self.new(x.N, 2 * x.H, 2 * x.W, x.C), relu_fused=relu)
        ops.upsample_fwd(x, u.act, stats=None, relu=relu, align=align)

        def bwd():
            du = self.grad_of(u.act)
            if a.ready:
                tmp = self.new(x.N, x.H, x.W, x.C)
                ops.upsample_bwd(du, None, tmp, relu=False, align=align)
                ops.copy_slice(tmp, self.grad_of(x), accumulate=True)
            else:
                ops.upsample_bwd(du, None, self.grad_of(x), relu=False, align=align)
            a.ready = True
        self.record(bwd)
        return u

    def head(self, e, wname, bname):
        """1x1 classification head -> NCHW fp32 logits (a torch tensor, the network output)."""
        w, b = self.p[wname].detach(), self.p[bname].detach()
        Cout = w.shape[0]
        x = e.act
        logits = torch.empty(x.N, Cout, x.H, x.W, dtype=torch.float32, device=self.device)
        ops.head_fwd(x, w.reshape(Cout, -1).contiguous(), b, logits)
        record = (e, w, wname, bname, logits)
        if getattr(self, "_head", None) is None:
            self._head = record
        else:                                   # a second head (models/unetvgg.py:1095): its logits leave as an extra output
            self.exports.append(record)
        return logits

    def head_wide(self, x, wname, bname):
        """1x1 classification head over a wide concatenated input (models/unetvgg.py:962-967,993: 6.25 x channelscale
        channels): a 1x1 convolution on the matrix cores with the class count padded to 8 output channels -> NCHW fp32
        logits.  x: list of Vars tiling one buffer."""
        w = self.p[wname].detach()
        Cout = w.shape[0]
        segs = self._segments(x)
        act = segs[0][0]
        assert len(segs) == 1 and Cout <= 8
        w1 = torch.zeros(8, act.C, 1, 1, dtype=torch.float32, device=self.device)
        w1[:Cout] = w.reshape(Cout, act.C, 1, 1)
        wf, wb = ops.conv_pack(w1, self.dtype, want_fwd=True, want_bwd=self.need_grad)
        y = self.new(act.N, act.H, act.W, 8)
        ops.conv_fwd(act, wf, 8, y, taps=1, bias=self._bias(bname, 8))
        logits = y.nchw()[:, :Cout].float().contiguous()
        self._head = (None, w, wname, bname, logits, segs, wb)
        return logits

    def _head_wide_backward(self, dlogits):
        _, w, wname, bname, _, segs, wb = self._head
        Cout, act = w.shape[0], segs[0][0]
        dy = ops.nchw_to_nhwc(dlogits.contiguous().float(), self.dtype, C=8)       # channels past Cout are zero
        dw8 = torch.empty(8, act.C, dtype=torch.float32, device=self.device)
        ops.conv_wgrad(act, dy, dw8, taps=1)
        dw = self.alloc_grad(wname, w)
        dw.copy_(dw8[:Cout].view_as(w))
        self._emit(wname, dw)
        db8 = ops.colsum(dy, torch.empty(8, dtype=torch.float32, device=self.device))
        db = self.alloc_grad(bname, self.p[bname])
        db.copy_(db8[:Cout])
        self._emit(bname, db)
        self._dgrad(segs, dy, wb, act.C, taps=1)

    def head_backward(self, dlogits, record=None):
        if record is None and self._head[0] is None:
            return self._head_wide_backward(dlogits)
        e, w, wname, bname, _ = record or self._head
        Cout = w.shape[0]
        dl = dlogits.contiguous().float()
        dw = self.alloc_grad(wname, w)
        db = self.alloc_grad(bname, self.p[bname])
        ops.head_bwd(e.act, w.reshape(Cout, -1).contiguous(), dl, self.grad_of(e.act), e.relu_fused, dw, db)
        e.ready = True
        self._emit(wname, dw)
        self._emit(bname, db)

    def unused(self, names):
        """Parameters outside the graph of this call: their gradient is zero where a flat gradient buffer expects one."""
        def bwd():
            if self.grad_alloc is None:
                return
            for n in names:
                g = self.alloc_grad(n, self.p[n])
                self._zero_grad(n, g)
                self._emit(n, g)
        self.record(bwd)
