"""nn.Module base that routes CUDA tensors through the segk tape inside one autograd.Function.

Subclasses provide `_emit(rt, x) -> logits` (the network expressed with segk.engine.Runtime helpers) and
`_forward_host(x)` (the plain torch.nn path that serves CPU tensors, e.g. the reference's
`saved_model.cpu()` inference in result_utils/save_img.py:122-143).  GPU tensors never take the torch
path: without libsegk.so the forward raises.
"""
import os

import torch
from torch import nn

from . import _lib, ops
from .engine import Runtime
from .packplan import PackPlan

_DEFAULT_DTYPE = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}[
    os.environ.get("SEGK_DTYPE", "fp32").lower()]


def set_default_compute_dtype(dtype):
    """Arithmetic type of the HIP kernels for modules that did not choose one: torch.float32 (parity with
    the fp32 reference, default) or torch.bfloat16 (matrix-core rate; fp32 accumulation and statistics)."""
    global _DEFAULT_DTYPE
    assert dtype in (torch.float32, torch.bfloat16)
    _DEFAULT_DTYPE = dtype


class _TapeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, need_grad, *params):
        names = net._segk_param_names
        dtype = net.compute_dtype()
        hooks = getattr(net, "_segk_grad_hooks", None) or (None, None, None, None, None)
        if need_grad and len(hooks) > 4 and hooks[4] is not None:
            hooks[4]()
        if not hasattr(net, "_segk_zero_grads"):
            net._segk_zero_grads = set()
        rt = net._runtime_cls(dict(zip(names, params)), dict(net.named_buffers()), dtype, x.device, net.training, need_grad,
                              grad_sink=hooks[0], grad_alloc=hooks[1], zero_state=net._segk_zero_grads)
        # weight packing: replay the recorded plan of this configuration in one launch, or record one
        plans = net.__dict__.setdefault("_segk_pack_plans", {})
        key = (dtype, need_grad, net.training)
        plan = plans.get(key)
        if plan is not None and plan.ready and plan.valid(rt.p):
            plan.run(x.device)
        else:
            plan = plans[key] = PackPlan()
        rt.pack_plan = plan
        out = net._emit(rt, x)
        if not plan.ready:
            plan.finalize(x.device)
        rt.finish_forward(net)
        ctx.rt, ctx.names, ctx.hooks = rt, names, hooks
        ctx.set_materialize_grads(False)
        if isinstance(out, tuple):               # (logits, feature maps made with rt.export)
            return (out[0],) + tuple(out[1])
        return out

    @staticmethod
    def backward(ctx, dlogits, *dfeats):
        rt = ctx.rt
        if dlogits is None and all(g is None for g in dfeats):
            return (None, None, None) + tuple(None for _ in ctx.names)
        begin, finish = ctx.hooks[2], ctx.hooks[3]
        if begin is not None:
            begin()
        if dlogits is None:
            dlogits = torch.zeros_like(rt._head[4])
        rt.head_backward(dlogits)
        rt.import_gradients(dfeats)
        rt.backward()
        ctx.rt = None
        if finish is not None:
            # flat mode: gradients were written (and all-reduced) in place in the flat buffer that p.grad views
            finish()
            return (None, None, None) + tuple(None for _ in ctx.names)
        grads = tuple(rt.grads.get(n) if ctx.needs_input_grad[3 + i] else None for i, n in enumerate(ctx.names))
        return (None, None, None) + grads


class SegkNet(nn.Module):
    _segk_dtype = None
    _runtime_cls = Runtime      # tape flavour; volume networks use segk.engine3d.Runtime3D

    def set_compute_dtype(self, dtype):
        assert dtype in (None, torch.float32, torch.bfloat16)
        self._segk_dtype = dtype
        return self

    def compute_dtype(self):
        if self._segk_dtype is not None:
            return self._segk_dtype
        if torch.is_autocast_enabled():
            return torch.bfloat16
        return _DEFAULT_DTYPE

    def forward(self, x):
        if not x.is_cuda:
            return self._forward_host(x)
        _lib.lib()   # raises if the HIP library is absent
        named = list(self.named_parameters())
        self._segk_param_names = [n for n, _ in named]
        params = [p for _, p in named]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        out = _TapeFn.apply(self, x, need_grad, *params)
        if isinstance(out, tuple):               # networks with extra outputs: (logits, feature maps) in the reference's nesting
            return self._pack_outputs(out[0], list(out[1:]))
        return out

    def _pack_outputs(self, logits, feats):
        return logits, feats


class _BlockFn(torch.autograd.Function):
    """A stand-alone building block (decoder block, double-conv block) on NCHW tensors: layout conversion at
    both ends, data gradient returned for the input."""

    @staticmethod
    def forward(ctx, block, x, need_grad, *params):
        names = block._segk_param_names
        rt = Runtime(dict(zip(names, params)), dict(block.named_buffers()), block.compute_dtype(), x.device, block.training,
                     need_grad)
        v = rt.input(x)
        out = block._emit_block(rt, v)
        rt.finish_forward()
        ctx.rt, ctx.names, ctx.v, ctx.out = rt, names, v, out
        return out.act.nchw().float().contiguous()

    @staticmethod
    def backward(ctx, dout):
        rt = ctx.rt
        g = rt.grad_of(ctx.out.act)
        g.nchw().copy_(dout)
        if ctx.out.relu_fused:     # nobody downstream applies the output ReLU's backward: do it here
            ops.relu_bwd(g, ctx.out.act, g)
        rt.backward_keep()
        dx = rt.grad_of(ctx.v.act).nchw().float().contiguous() if (ctx.needs_input_grad[1] and ctx.v.ready) else None
        rt.release()
        grads = tuple(rt.grads.get(n) if ctx.needs_input_grad[3 + i] else None for i, n in enumerate(ctx.names))
        ctx.rt = None
        return (None, dx, None) + grads


class SegkBlock(SegkNet):
    """Base of blocks that are also usable on their own: CUDA tensors run `_emit_block` on a private tape."""

    def forward(self, x):
        if not x.is_cuda:
            return self._forward_host(x)
        _lib.lib()
        if x.shape[1] % 8:
            raise NotImplementedError("stand-alone HIP blocks take a channel count that is a multiple of 8")
        named = list(self.named_parameters())
        self._segk_param_names = [n for n, _ in named]
        params = [p for _, p in named]
        need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _BlockFn.apply(self, x, need_grad, *params)
