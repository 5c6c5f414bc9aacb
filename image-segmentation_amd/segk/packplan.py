"""Replay of a network's weight packing as one kernel launch.

Every training step repacks every convolution weight (the optimizer changed them) into the swizzled images the MFMA
kernels read.  The first forward of a network packs layer by layer and records what it packed; later forwards with the
same parameters storage replay the whole list through `segk_conv_pack_batch` into the same persistent buffers.
"""
import struct

import torch

from . import ops
from ._lib import call, query


class PackPlan:
    def __init__(self):
        self.entries = {}        # key -> (wf, wb)
        self.descs = []          # (w_ptr, wf_ptr, wb_ptr, Cout, Cin, taps, seg0, so, si)
        self.sources = []        # (parameter name, data_ptr) the descriptors point into
        self.ready = False
        self.table = None
        self.dtype = None

    # ---------------------------------------------------------------- recording
    def record(self, key, wname, w, dtype, wf, wb, seg0=0):
        """Remember a pack of parameter `wname` (w: its storage viewed as [Cout,Cin,k,k] or [Cout,Cin,3,3,3])."""
        self.dtype = dtype
        self.entries[key] = (wf, wb)
        self.sources.append((wname, w.data_ptr()))
        Cout, Cin = w.shape[0], w.shape[1]
        p = lambda t, off=0: 0 if t is None else t.data_ptr() + off   # noqa: E731
        if w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3):
            # the three depth-tap images lie back to back (forward: K = Cin, rows = Cout; data gradient: the other way round)
            kr = query("segk_conv_krow", ops.dt(dtype))
            up = lambda v, m: (v + m - 1) // m * m   # noqa: E731
            per_f = up(Cin, kr) // kr * 9 * up(Cout, 32) * 128
            per_b = up(Cout, kr) // kr * 9 * up(Cin, 32) * 128
            for kd in range(3):
                self.descs.append((w.data_ptr() + kd * 9 * 4, p(wf, kd * per_f), p(wb, (2 - kd) * per_b), Cout, Cin, 9, Cin, Cin * 27, 27))
        else:
            taps = w[0, 0].numel()
            self.descs.append((w.data_ptr(), p(wf), p(wb), Cout, Cin, taps, seg0 or Cin, Cin * taps, taps))

    def finalize(self, device):
        if self.descs:
            raw = b"".join(struct.pack("<QQQiiiiqq", *d) for d in self.descs)
            self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.ready = True

    # ---------------------------------------------------------------- replay
    def valid(self, params):
        return all(n in params and params[n].data_ptr() == ptr for n, ptr in self.sources)

    def run(self, device):
        if not self.descs:
            return
        dev = device.index if device.index is not None else torch.cuda.current_device()
        n = len(self.descs)
        # workgroups per descriptor: enough that the largest tensor (a few 100 K slots) is a handful of trips per thread
        call("segk_conv_pack_batch", ops.dt(self.dtype), self.table.data_ptr(), n, 256 if n <= 64 else 32, dev,
             torch.cuda.current_stream(dev).cuda_stream)
