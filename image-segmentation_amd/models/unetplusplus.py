"""Nested U-Net (UNet++), MI355X-native.

Constructor signatures, attribute names and state_dict layout of the reference
(/root/reference/models/unetplusplus.py: conv_block_nested :6-25, Nested_UNet :27-82).  The dense skip
concatenations of :64-79 are never materialised: every node x[i][j] is written by its producer into
a per-resolution NHWC buffer [x[i][0] | x[i][1] | ...], a consumer convolution reads a prefix of that
buffer as its first input segment and the up-sampled deeper node as its second.
"""
import torch
from torch import nn

from segk.net import SegkBlock, SegkNet
from segk.ops import Act

_WIDTHS = (64, 128, 256, 512, 1024)


class conv_block_nested(SegkBlock):
    """(conv3x3 + bias -> BatchNorm -> ReLU) twice."""

    def __init__(self, in_ch, mid_ch, out_ch):
        super().__init__()
        self.activation = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(in_ch, mid_ch, kernel_size=3, padding=1, bias=True)
        self.bn1 = nn.BatchNorm2d(mid_ch)
        self.conv2 = nn.Conv2d(mid_ch, out_ch, kernel_size=3, padding=1, bias=True)
        self.bn2 = nn.BatchNorm2d(out_ch)

    def _forward_host(self, x):
        x = self.activation(self.bn1(self.conv1(x)))
        return self.activation(self.bn2(self.conv2(x)))

    def emit(self, rt, prefix, x, out=None, image=None, pool=False):
        """Append the block to a segk tape.  x: Var / concat description; image: raw NCHW input of the first block;
        pool=True returns (output, maxpool2x2(output)), the pool fused into the last norm's passes."""
        zero = rt.training     # a bias in front of a batch-statistics norm has an identically zero gradient
        if image is not None and rt.can_fuse_first_norm(image, self.conv1.out_channels):
            # convolution + norm + ReLU from the image, the convolution's output recomputed instead of stored
            h = rt.conv_first_bn_relu(image, prefix + "conv1.weight", prefix + "conv1.bias", prefix + "bn1")
        else:
            if image is not None:
                y = rt.conv_first(image, prefix + "conv1.weight", prefix + "conv1.bias", bias_grad_zero=zero, stats=True)
            else:
                y = rt.conv(x, prefix + "conv1.weight", prefix + "conv1.bias", bias_grad_zero=zero, stats=True)
            h = rt.bn_relu(y, prefix + "bn1", one_reader=True)
        y = rt.conv(h, prefix + "conv2.weight", prefix + "conv2.bias", bias_grad_zero=zero, stats=True)
        return rt.bn_relu(y, prefix + "bn2", out=out, pool=pool)

    def _emit_block(self, rt, v):
        return self.emit(rt, "", v)


class Nested_UNet(SegkNet):
    """deep_supervision=False (default) is the reference's network: one 1x1 head `final` on x(0,4), same state_dict.
    deep_supervision=True is an extension the reference does not contain (its file has a single head, :58,:81-82): the
    UNet++ paper's four heads `final1..final4` on x(0,1)..x(0,4); forward returns the list of their logits and the
    training loss is the sum of the per-head losses (`deep_supervision_loss`)."""

    def __init__(self, in_ch=1, out_ch=3, deep_supervision=False):
        super().__init__()
        self.deep_supervision = deep_supervision
        f = _WIDTHS
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.Up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        # registration order = the reference's, so state_dict key order matches
        for i in range(5):
            setattr(self, f"conv{i}_0", conv_block_nested(in_ch if i == 0 else f[i - 1], f[i], f[i]))
        for j in range(1, 5):
            for i in range(5 - j):
                setattr(self, f"conv{i}_{j}", conv_block_nested(f[i] * j + f[i + 1], f[i], f[i]))
        if deep_supervision:
            for j in range(1, 5):
                setattr(self, f"final{j}", nn.Conv2d(f[0], out_ch, kernel_size=1))
        else:
            self.final = nn.Conv2d(f[0], out_ch, kernel_size=1)

    @staticmethod
    def _schedule():
        """Nodes in the reference's evaluation order: after backbone node (i,0) the anti-diagonal it completes."""
        for i in range(5):
            yield i, 0
            for j in range(1, i + 1):
                yield i - j, j

    def _forward_host(self, x):
        node = {}
        for i, j in self._schedule():
            block = getattr(self, f"conv{i}_{j}")
            if j == 0:
                node[i, 0] = block(x if i == 0 else self.pool(node[i - 1, 0]))
            else:
                node[i, j] = block(torch.cat([node[i, k] for k in range(j)] + [self.Up(node[i + 1, j - 1])], 1))
        if self.deep_supervision:
            return [getattr(self, f"final{j}")(node[0, j]) for j in range(1, 5)]
        return self.final(node[0, 4])

    def _pack_outputs(self, logits, feats):
        # the tape hands back (head on x(0,4), [heads on x(0,1..3)]); callers get them shallowest first
        return list(feats) + [logits]

    def _emit(self, rt, x):
        N, cin, H, W = x.shape
        if H % 16 or W % 16:
            raise RuntimeError(f"Nested_UNet needs H and W divisible by 16 (four 2x2 pools), got {H}x{W}")
        f = _WIDTHS
        # level i keeps nodes j = 0 .. 3-i side by side (the ones later convolutions concatenate)
        level = {i: rt.new(N, H >> i, W >> i, f[i] * (4 - i)).buf for i in range(4)}
        node, pooled = {}, {}
        for i, j in self._schedule():
            block = getattr(self, f"conv{i}_{j}")
            out = Act(level[i], f[i] * j, f[i]) if (i < 4 and j < 4 - i) else None
            if j == 0:
                # backbone node: its max pool (input of the next backbone node) comes out of the same passes
                if i == 0 and 9 * cin <= 32:
                    res = block.emit(rt, "conv0_0.", None, out=out, image=x, pool=True)
                elif i == 0:
                    res = block.emit(rt, "conv0_0.", rt.input(x), out=out, pool=True)
                else:
                    res = block.emit(rt, f"conv{i}_0.", pooled[i - 1], out=out, pool=i < 4)
                node[i, 0], pooled[i] = res if i < 4 else (res, None)
            else:
                up = rt.up(node[i + 1, j - 1], align=True)
                node[i, j] = block.emit(rt, f"conv{i}_{j}.", ([node[i, k] for k in range(j)], up), out=out)
        if self.deep_supervision:
            # the head on x(0,4) is the tape's main output; the three others leave as extra outputs whose gradients
            # re-enter through the slices of the level-0 buffer that x(0,1..3) occupy
            last = rt.head(node[0, 4], "final4.weight", "final4.bias")
            return last, [rt.head(node[0, j], f"final{j}.weight", f"final{j}.bias") for j in range(1, 4)]
        return rt.head(node[0, 4], "final.weight", "final.bias")


def deep_supervision_loss(outputs, target, criteria):
    """Sum over the heads of the sum of `criteria` (callables loss(logits, target)): the deep-supervision objective."""
    total = None
    for logits in outputs:
        for crit in criteria:
            term = crit(logits, target)
            total = term if total is None else total + term
    return total
