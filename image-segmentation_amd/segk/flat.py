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
        if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(group)
        self.comm_stream = torch.cuda.Stream(device) if (device is not None and device.type == "cuda" and self.world > 1) else None
        self.reset()

    def reset(self):
        self.pending = list(self.expected)
        self.works, self.events = [], []
        self.launched = []

    def mark(self, name, flat_grad):
        b = self.bucket_of[name]
        self.pending[b] -= 1
        assert self.pending[b] >= 0, f"gradient of {name} produced twice in one backward"
        if self.pending[b] == 0:
            self._launch(b, flat_grad)

    def _launch(self, b, flat_grad):
        self.launched.append(b)
        if self.world == 1:
            return
        import torch.distributed as dist
        lo, hi = self.ranges[b]
        view = flat_grad[lo:hi]
        if self.comm_stream is not None and dist.get_backend(self.group) != "nccl":
            # a CPU-only backend (gloo) under CUDA tensors, e.g. a rehearsal with several ranks on one card: reduce a host copy.
            # (gloo's own path for device tensors stalled for minutes per bucket when two ranks shared a GPU)
            host = view.cpu()                                 # waits for the gradient kernels of the current stream
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
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                view.mul_(1.0 / self.world)
                done = torch.cuda.Event()
                done.record()
            self.events.append(done)
        else:                                                 # CPU process groups (gloo): asynchronous work handles
            self.works.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True), view))

    def finish(self):
        """Make the reduced gradients visible to whatever runs next on the current stream."""
        assert all(p == 0 for p in self.pending), f"backward left buckets incomplete: {self.pending}"
        for ev in self.events:
            torch.cuda.current_stream().wait_event(ev)
        for work, view in self.works:
            work.wait()
            view.mul_(1.0 / self.world)
        self.events, self.works = [], []


class FlatModel:
    def __init__(self, net, bucket_mb=25.0, group=None, broadcast=True):
        self.net = net
        named = list(net.named_parameters())[::-1]            # backward completion order
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
        if broadcast and self.buckets.world > 1:
            import torch.distributed as dist
            dist.broadcast(self.flat_param, src=0, group=group)
            # BatchNorm buffers: one coalesced broadcast per dtype instead of one per tensor
            for dtype in (torch.float32, torch.int64):
                bufs = [b for b in net.buffers() if b.dtype == dtype]
                if not bufs:
                    continue
                flat = torch.cat([b.detach().reshape(-1) for b in bufs])
                dist.broadcast(flat, src=0, group=group)
                off = 0
                for b in bufs:
                    b.data.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()
        net._segk_grad_hooks = (self._sink, self._alloc, self._begin, self._finish)

    # hooks called by the tape
    def _begin(self):
        self.buckets.reset()

    def _alloc(self, name, like):
        return self.views[name]

    def _sink(self, name, tensor):
        self.buckets.mark(name, self.flat_grad)

    def _finish(self):
        self.buckets.finish()
