"""Flat parameter / gradient storage with bucketed, overlapped gradient all-reduce.

`FlatModel(net)` re-homes every parameter of a SegkNet into one fp32 buffer and gives the tape a matching
flat gradient buffer: weight-gradient kernels write straight into it (no per-tensor copies), the fused
Adam kernel updates everything in one launch, and -- with a process group -- contiguous ~25 MB buckets of it
are all-reduced on a side stream as soon as the backward pass has produced their last gradient, so the
collective overlaps the rest of backward.  The reference has no data parallelism
(pl_utils/pl_train.py:39-42 runs one device); semantics chosen here are the faithful extension: per-rank
BatchNorm statistics and loss normalisation, gradients averaged over ranks.

The flat layout follows the order in which backward completes gradients (reverse registration order), so
buckets fill front to back and the last, exposed bucket holds the small first-layer tensors.
"""
import os

import torch


class GradBuckets:
    """Bookkeeping of which flat-gradient ranges are complete; launches one all-reduce per finished bucket."""

    def __init__(self, sizes, bucket_bytes, group=None, device=None):
        # sizes: list of (name, numel) in flat order
        self.group = group
        self.offsets, self.bucket_of, self.ranges = {}, {}, []
        off, start, b = 0, 0, 0
        for name, n in sizes:
            self.offsets[name] = (off, n)
            self.bucket_of[name] = b
            off += (n + 63) // 64 * 64         # every tensor starts on a 256-byte boundary (the kernels need 16; padding stays zero)
            if (off - start) * 4 >= bucket_bytes:
                self.ranges.append((start, off))
                start, b = off, b + 1
        if off > start:
            self.ranges.append((start, off))
        self.total = off
        self.expected = [0] * len(self.ranges)
        for name in self.offsets:
            self.expected[self.bucket_of[name]] += 1
        self.world = 1
        initialised = torch.distributed.is_available() and torch.distributed.is_initialized()
        if group is not None or initialised:
            self.world = torch.distributed.get_world_size(group)
        # SEGK_FORCE_COLLECTIVES=1: issue the collectives even in a one-rank group (a one-GPU box can then run the RCCL
        # branch -- event, communication stream, all-reduce, scaling -- end to end; the result is the identity)
        self.active = self.world > 1 or (initialised and os.environ.get("SEGK_FORCE_COLLECTIVES") == "1")
        self.comm_stream = torch.cuda.Stream(device) if (device is not None and device.type == "cuda" and self.active) else None
        self.exposed = None      # set to a list to collect (start, end) events around the compute stream's wait for the collectives
        self.reset()

    def reset(self):
        self.pending = list(self.expected)
        self.works, self.events = [], []
        self.launched = []
        self.marked = set()

    def mark(self, name, flat_grad):
        b = self.bucket_of[name]
        self.pending[b] -= 1
        assert self.pending[b] >= 0 and name not in self.marked, f"gradient of {name} produced twice in one backward"
        self.marked.add(name)
        if self.pending[b] == 0:
            self._launch(b, flat_grad)

    def complete_missing(self, flat_grad):
        """Parameters whose gradient no tape produced in this backward (a network whose output did not reach the loss):
        their gradient is zero.  Write it and mark them, so that every bucket is reduced on every rank.  -> their names."""
        missing = [n for n in self.offsets if n not in self.marked]
        for n in missing:
            off, cnt = self.offsets[n]
            flat_grad[off:off + cnt].zero_()
            self.mark(n, flat_grad)
        return missing

    def _launch(self, b, flat_grad):
        self.launched.append(b)
        if not self.active:
            return
        import torch.distributed as dist
        lo, hi = self.ranges[b]
        view = flat_grad[lo:hi]
        if self.comm_stream is not None and dist.get_backend(self.group) != "nccl":
            # a CPU-only backend (gloo) under CUDA tensors, e.g. a rehearsal with several ranks on one card: reduce a host copy.
            # (gloo's own path for device tensors stalled for minutes per bucket when two ranks shared a GPU)
            torch.cuda.synchronize(flat_grad.device)          # gradients may have been produced on two streams
            host = view.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(host.mul_(1.0 / self.world))
        elif self.comm_stream is not None:
            from .engine import side_stream
            ready = torch.cuda.Event()
            ready.record()                                    # on the current stream, after the last gradient kernel
            other = side_stream(flat_grad.device, create=False)   # gradients are produced on two streams: wait for both
            pending = [ready]
            for st in (other, torch.cuda.default_stream(flat_grad.device)) if other is not None else ():
                if st != torch.cuda.current_stream(flat_grad.device):
                    ev = torch.cuda.Event()
                    ev.record(st)
                    pending.append(ev)
            with torch.cuda.stream(self.comm_stream):
                for ev in pending:
                    self.comm_stream.wait_event(ev)
                dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group)   # RCCL averages in the collective: no scaling launch
                done = torch.cuda.Event()
                done.record()
            self.events.append(done)
        else:                                                 # CPU process groups (gloo): asynchronous work handles
            self.works.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True), view))

    def finish(self):
        """Make the reduced gradients visible to whatever runs next on the current stream."""
        assert all(p == 0 for p in self.pending), f"backward left buckets incomplete: {self.pending}"
        if self.events and self.exposed is not None:
            # how long the compute stream stands still for the collectives: events around the waits, read by the caller later
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for ev in self.events:
                torch.cuda.current_stream().wait_event(ev)
            e1.record()
            self.exposed.append((e0, e1))
            self.events = []
        for ev in self.events:
            torch.cuda.current_stream().wait_event(ev)
        for work, view in self.works:
            work.wait()
            view.mul_(1.0 / self.world)
        self.events, self.works = [], []


class FlatModel:
    """Flat storage for the trainable parameters of `net`: a SegkNet, or a wrapper module (UNetVgg256, UNetVggTwo,
    SegmModelPL ...) whose trainable parameters all live inside SegkNet sub-modules.  The tape of every such network
    gets hooks that translate its own parameter names to the wrapper's (prefixed) names.  Parameters with
    requires_grad=False stay where they are: they are neither bucketed, all-reduced nor touched by the optimizer."""

    def __init__(self, net, bucket_mb=25.0, group=None, broadcast=True):
        from .net import SegkNet
        self.net = net
        if isinstance(net, SegkNet):
            targets = [("", net)]
        else:
            targets, seen = [], []
            for mname, m in net.named_modules():
                if isinstance(m, SegkNet) and not any(mname.startswith(s + ".") for s in seen):
                    seen.append(mname)
                    targets.append((mname + ".", m))
            if not targets:
                raise TypeError(f"FlatModel needs a SegkNet or a module that contains one, got {type(net).__name__}")
            outside = [n for n, p in net.named_parameters()
                       if p.requires_grad and not any(n.startswith(pre) for pre, _ in targets)]
            if outside:
                raise TypeError(f"FlatModel: trainable parameters outside the tape networks are not supported: {outside[:4]}")
        named = [(n, p) for n, p in list(net.named_parameters())[::-1] if p.requires_grad]   # backward completion order
        if not named:
            raise ValueError("FlatModel: the module has no trainable parameter")
        device = named[0][1].device
        sizes = [(n, p.numel()) for n, p in named]
        self.buckets = GradBuckets(sizes, int(bucket_mb * (1 << 20)), group=group, device=device)
        total = self.buckets.total
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.views = {}
        for n, p in named:
            off, cnt = self.buckets.offsets[n]
            self.flat_param[off:off + cnt].copy_(p.detach().reshape(-1))
            p.data = self.flat_param[off:off + cnt].view_as(p)
            gv = self.flat_grad[off:off + cnt].view_as(p)
            p.grad = gv
            self.views[n] = gv
        if broadcast and self.buckets.active:
            import torch.distributed as dist
            dist.broadcast(self.flat_param, src=0, group=group)
            # frozen parameters and BatchNorm buffers: one coalesced broadcast per dtype instead of one per tensor
            frozen = [p for p in net.parameters() if not p.requires_grad]
            for dtype in (torch.float32, torch.int64):
                bufs = [b for b in list(net.buffers()) + frozen if b.dtype == dtype]
                if not bufs:
                    continue
                flat = torch.cat([b.detach().reshape(-1) for b in bufs])
                dist.broadcast(flat, src=0, group=group)
                off = 0
                for b in bufs:
                    b.data.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()
        # Several tape networks under one wrapper do their backward passes one after the other within ONE autograd backward.
        # The bucket state is reset when the first of them starts.  It is closed (gradients nobody produced zeroed, every
        # bucket reduced, the compute stream made to wait for the collectives) by a callback queued on the autograd engine,
        # which runs when that backward pass is over -- however many of the networks took part in it: one whose parameters are
        # all frozen records no backward, and one whose output did not reach the loss is never called.  Hooks driven by hand
        # outside a backward pass (tests) cannot queue a callback; then the pass closes when every network that has a
        # trainable parameter has finished.
        self._done = 0
        self._open = False
        self._queued = False
        self._ntargets = sum(1 for prefix, m in targets if any(p.requires_grad for p in m.parameters()))
        for prefix, m in targets:
            m._segk_grad_hooks = (self._make_sink(prefix), self._make_alloc(prefix), self._begin, self._finish, self._on_forward)

    # hooks called by the tape
    def _on_forward(self):
        """A forward pass starts a new step: a backward pass that was cut short by an exception must not leave the state open."""
        self._open = False

    def _begin(self):
        if self._open:
            return
        self._open = True
        self._done = 0
        self.buckets.reset()
        try:
            torch.autograd.Variable._execution_engine.queue_callback(self._close)
            self._queued = True
        except RuntimeError:                 # not inside a backward pass
            self._queued = False

    def _close(self):
        if not self._open:
            return
        self._open = False
        self.buckets.complete_missing(self.flat_grad)
        self.buckets.finish()

    def _make_alloc(self, prefix):
        def alloc(name, like):
            view = self.views.get(prefix + name)
            if view is None:                      # frozen parameter: the tape's gradient goes to a throw-away tensor
                return torch.empty_like(like, dtype=torch.float32)
            return view
        return alloc

    def _make_sink(self, prefix):
        def sink(name, tensor):
            if prefix + name in self.views:
                self.buckets.mark(prefix + name, self.flat_grad)
        return sink

    # single-network spellings kept for callers that drive the hooks by hand
    def _alloc(self, name, like):
        return self._make_alloc("")(name, like)

    def _sink(self, name, tensor):
        self._make_sink("")(name, tensor)

    def _finish(self):
        self._done += 1
        if not self._queued and self._done >= self._ntargets:
            self._close()
