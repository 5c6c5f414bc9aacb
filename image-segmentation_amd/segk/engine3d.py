"""Tape helpers for volumes: the layers of the 3-D DenseNet classifier (models/densenet3d.py, models/resnt.py).

A batch of volumes [N,C,D,H,W] lives slice by slice in an NHWC buffer [N*D, H, W, C]; `D` travels next to the
Var.  Channel-wise layers (BatchNorm3d, ReLU, 1x1x1 convolution) are the 2-D kernels with N*D as the batch; the
layers that look across slices have their own kernels (csrc/volume.hip, segk_conv3d_*).
"""
import torch

from . import ops
from .engine import Runtime, Var
from .ops import Act


class Runtime3D(Runtime):
    # ---------------------------------------------------------------- input
    def input_volume(self, x):
        """[N,Cin,D,H,W] fp32 -> (Var over [N*D,H,W,8], D); the image needs no gradient."""
        N, Cin, D, H, W = x.shape
        assert Cin <= 8, "the patch gather takes at most 8 input channels"
        x = x.detach().float()
        x2 = x.reshape(N * D, 1, H, W) if Cin == 1 else x.permute(0, 2, 1, 3, 4).reshape(N * D, Cin, H, W)
        return Var(ops.nchw_to_nhwc(x2, self.dtype, C=8)), D

    # ---------------------------------------------------------------- strided stem convolutions
    def conv_patches(self, v, D, wname, kernel, stride, pad, need_dx, stats=False):
        """Conv3d(bias=False) with kernel (kd,kh,kw), stride (1,sh,sw), padding (kd/2,ph,pw) on a few-channel input,
        factorised: the (kh,kw) window of every slice is gathered once into patch rows (kd times smaller than a full 3-D
        gather), and a depth-only convolution on the MFMA kernel sums the kd neighbouring slices' rows.
        Output channels are stored rounded up to 8 (extra ones exactly zero).  -> (Var, D)"""
        w = self.p[wname]
        Cout, Cin = w.shape[0], w.shape[1]
        kd, kh, kw = kernel
        assert tuple(w.shape[2:]) == tuple(kernel)
        if stride[0] != 1 or pad[0] != kd // 2 or kd % 2 == 0 or kd > 7:
            raise NotImplementedError("the HIP stem convolution takes depth stride 1 with 'same' depth padding and kd in {1,3,5,7}")
        cs = (Cout + 7) // 8 * 8
        k2, s2, p2 = (1, kh, kw), (1, stride[1], stride[2]), (0, pad[1], pad[2])
        patches, _ = ops.im2col3d(v.act, D, Cin, k2, s2, p2)            # [N*D, Ho, Wo, Qp], columns (b, e, c)
        Q, Qp = Cin * kh * kw, patches.C
        w1 = torch.zeros(cs, Qp, kd, dtype=torch.float32, device=self.device)
        w1[:Cout, :Q] = w.detach().permute(0, 3, 4, 1, 2).reshape(Cout, Q, kd)   # [co][(b,e,c)][a]
        wf, wb = ops.convd_pack(w1, self.dtype, want_bwd=self.need_grad and need_dx)
        y = Var(self.new(patches.N, patches.H, patches.W, cs))
        ops.convd_fwd(patches, D, kd, wf, y.act, stat_partials=self._stat_buffer(y, stats))

        def bwd():
            dy = self.grad_of(y.act)
            dw = torch.empty(cs, Q, kd, dtype=torch.float32, device=self.device)
            ops.convd_wgrad(patches, D, kd, dy, dw, cin_real=Q)
            g = self.alloc_grad(wname, w)
            g.copy_(dw[:Cout].view(Cout, kh, kw, Cin, kd).permute(0, 3, 4, 1, 2))
            self._emit(wname, g)
            if need_dx:
                dpatches = self.new(patches.N, patches.H, patches.W, Qp)
                ops.convd_fwd(dy, D, kd, wb, dpatches)
                ops.col2im3d(dpatches, self.grad_of(v.act), D, Cin, k2, s2, p2, accumulate=v.ready)
                v.ready = True
        self.record(bwd)
        return y, D

    # ---------------------------------------------------------------- 1x1x1 and 3x3x3 convolutions
    def conv1x1(self, x, wname, out=None, stats=False):
        w = self.p[wname]
        Cout, Cin = w.shape[0], w.shape[1]
        a = x.act
        assert a.C == Cin
        wf, wb = self._pack(wname, view=(Cout, Cin, 1, 1))
        y = Var(out or self.new(a.N, a.H, a.W, Cout))
        ops.conv_fwd(a, wf, Cout, y.act, taps=1, stat_partials=self._stat_buffer(y, stats))

        def bwd():
            dy = self.grad_of(y.act)
            dw = self.alloc_grad(wname, w)
            ops.conv_wgrad(a, dy, dw, taps=1)
            self._emit(wname, dw)
            ops.conv_fwd(dy, wb, Cin, self.grad_of(a), taps=1, accumulate=x.ready)
            x.ready = True
        self.record(bwd)
        return y

    def conv3d(self, x, D, wname, out=None):
        """Conv3d(kernel 3, padding 1, bias=False) (models/densenet3d.py:27-32)."""
        w = self.p[wname]
        Cout, Cin = w.shape[0], w.shape[1]
        a = x.act
        assert a.C == Cin
        wf, wb = self._pack(wname)
        y = Var(out or self.new(a.N, a.H, a.W, Cout))
        ops.conv3d_fwd(a, D, wf, y.act)

        def bwd():
            dy = self.grad_of(y.act)
            dw = self.alloc_grad(wname, w)
            ops.conv3d_wgrad(a, D, dy, dw)
            self._emit(wname, dw)
            ops.conv3d_fwd(dy, D, wb, self.grad_of(a), accumulate=x.ready)
            x.ready = True
        self.record(bwd)
        return y

    # ---------------------------------------------------------------- pools
    def maxpool3d(self, a, D, out=None):
        """MaxPool3d(3, stride 2, padding 1) (models/densenet3d.py:107-109).  -> (Var, Do)"""
        x = a.act
        N = x.N // D
        Do, Ho, Wo = ops.out_extent((D, x.H, x.W), (3, 3, 3), (2, 2, 2), (1, 1, 1))
        p = Var(out or self.new(N * Do, Ho, Wo, x.C))
        arg = torch.empty(N * Do * Ho * Wo * x.C, dtype=torch.uint8, device=self.device)
        ops.maxpool3d_fwd(x, D, p.act, arg)

        def bwd():
            ops.maxpool3d_bwd(self.grad_of(p.act), arg, self.grad_of(x), D, accumulate=a.ready)
            a.ready = True
        self.record(bwd)
        return p, Do

    def avgpool3d(self, a, D, out=None):
        """AvgPool3d(2, stride 2) (models/densenet3d.py:68).  -> (Var, Do)"""
        x = a.act
        N = x.N // D
        p = Var(out or self.new(N * (D // 2), x.H // 2, x.W // 2, x.C))
        ops.avgpool3d_fwd(x, D, p.act)

        def bwd():
            ops.avgpool3d_bwd(self.grad_of(p.act), self.grad_of(x), D, accumulate=a.ready)
            a.ready = True
        self.record(bwd)
        return p, D // 2

    # ---------------------------------------------------------------- DenseNet pieces
    def dense_block(self, feat, c_in, D, prefix, num_layers, growth):
        """_DenseBlock (models/densenet3d.py:44-52): `feat` is the block's concat buffer, its first c_in channels
        already written; every layer appends `growth` channels in place (the torch.cat of :41).  Returns the Var
        over the whole buffer.

        Gradient order: the block's consumer writes the gradient of the whole buffer first; each layer's first
        BatchNorm then ADDS its share to the prefix it read, so prefix Vars are created ready."""
        c = c_in
        for i in range(num_layers):
            lp = f"{prefix}denselayer{i + 1}."
            xin = Var(Act(feat.buf, feat.c0, c))
            xin.ready = True
            h = self.bn_relu(xin, lp + "norm1")
            h = self.conv1x1(h, lp + "conv1.weight", stats=True)      # its epilogue feeds norm2's batch statistics
            h = self.bn_relu(h, lp + "norm2")
            self.conv3d(h, D, lp + "conv2.weight", out=Act(feat.buf, feat.c0 + c, growth))
            c += growth
        assert c == feat.C
        return Var(feat)

    def transition(self, x, D, prefix, out=None):
        """_Transition (models/densenet3d.py:55-68): BatchNorm3d, ReLU, 1x1x1 convolution to half the channels, AvgPool3d(2)."""
        h = self.bn_relu(x, prefix + "norm")
        h = self.conv1x1(h, prefix + "conv.weight")
        return self.avgpool3d(h, D, out=out)

    def pooled_features(self, x, D, norm_prefix):
        """Final BatchNorm3d -> ReLU -> global average pool (models/densenet3d.py:129,153-157) -> fp32 [N, C]."""
        h = self.bn_relu(x, norm_prefix)
        a = h.act
        N = a.N // D
        feats = ops.rows_mean_fwd(a, N)

        def bwd(dfeats):
            ops.rows_mean_bwd(dfeats.contiguous(), self.grad_of(a))
            h.ready = True
        return feats, bwd

    # ---------------------------------------------------------------- fully connected tail
    def linear_tail(self, feats, feats_bwd, layers, squeeze_last=False):
        """Chain of nn.Linear layers on the pooled features (models/densenet3d.py:139,158; models/resnt.py:34-35,41-44):
        [N, <=1664] fp32 rows.  `layers`: (weight name, bias name) pairs."""
        acts = [feats]
        for wn, bn_ in layers:
            acts.append(ops.linear_fwd(acts[-1], self.p[wn].detach().contiguous(), self.p[bn_].detach()))
        out = acts[-1].squeeze(-1) if squeeze_last else acts[-1]
        self._tail = (acts, feats_bwd, layers, squeeze_last)
        return out

    def head_backward(self, dout):
        acts, feats_bwd, layers, squeeze_last = self._tail
        d = dout.float()
        if squeeze_last:
            d = d.unsqueeze(-1)
        d = d.contiguous()
        for k in range(len(layers) - 1, -1, -1):
            wn, bn_ = layers[k]
            w = self.p[wn].detach().contiguous()
            gw, gb = self.alloc_grad(wn, w), self.alloc_grad(bn_, self.p[bn_])
            d = ops.linear_bwd(acts[k], w, d, gw, gb)
            self._emit(wn, gw)
            self._emit(bn_, gb)
        feats_bwd(d)
