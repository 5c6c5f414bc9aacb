"""U-Net with a VGG16-BN encoder, MI355X-native.

Same constructor signatures, attribute names and state_dict layout as the reference classes
(/root/reference/models/unetvgg.py: make_decoder_block :6-41, UNetVgg :52-105, UNetVggInOut :158-213),
so checkpoints and callers are interchangeable.  The module tree is made of ordinary torch.nn layers
that only hold the parameters (and serve CPU tensors); for CUDA tensors `forward` hands the
parameters to the segk tape (segk/engine.py), which launches the HIP kernels: implicit-GEMM
convolutions on the matrix cores, fused BatchNorm/InstanceNorm/ReLU/pool/up-sample streams, with the
skip concatenations of :99-103 written in place into shared NHWC buffers.
"""
from collections import OrderedDict

import torch
from torch import nn

from segk.engine import Var
from segk.net import SegkBlock, SegkNet
from segk.ops import Act

# VGG configuration D: first feature index and conv widths of the five encoder stages; stages 2-5 start
# with the 2x2 max pool that precedes them in torchvision's `features` list, whose numbering the
# reference keeps by slicing that list (unetvgg.py:72-76).
_ENCODER = ((0, (64, 64)), (6, (128, 128)), (13, (256, 256, 256)), (23, (512, 512, 512)), (33, (512, 512, 512)))
# decoder blocks: name -> (in, mid, out); `in` = previous decoder output + skip width
_DECODER = OrderedDict(center=(512, 512, 256), dec5=(768, 512, 256), dec4=(768, 512, 256), dec3=(512, 256, 64),
                       dec2=(192, 128, 32))


class DecoderBlock(SegkBlock, nn.Sequential):
    """conv3x3+ReLU followed by one of three tails, then ReLU; child indices as in the reference."""

    def __init__(self, in_channels, middle_channels, out_channels, if_convtrans=False, noupsam=False, upsc=2):
        layers = [nn.Conv2d(in_channels, middle_channels, 3, padding=1), nn.ReLU(inplace=True)]
        if not noupsam and if_convtrans:
            layers.append(nn.ConvTranspose2d(middle_channels, out_channels, kernel_size=4, stride=2, padding=1))
            self_kind = "convtrans"
        else:
            layers += [nn.Conv2d(middle_channels, out_channels, 3, padding=1, bias=False), nn.InstanceNorm2d(out_channels)]
            self_kind = "norm"
            if not noupsam:
                layers.append(nn.Upsample(scale_factor=upsc, mode='bilinear', align_corners=False))
                self_kind = "bilinear"
        layers.append(nn.ReLU(inplace=True))
        nn.Sequential.__init__(self, *layers)
        self.kind, self.upsc = self_kind, upsc

    def _forward_host(self, x):
        return nn.Sequential.forward(self, x)

    def _emit_block(self, rt, v):
        return self.emit(rt, "", v)

    def emit(self, rt, prefix, x, out=None):
        """Append this block to a segk tape; x: Var or list of Vars tiling a concat buffer."""
        h = rt.conv(x, prefix + "0.weight", prefix + "0.bias", relu=True)
        if self.kind == "convtrans":
            return rt.conv_transpose_relu(h, prefix + "2.weight", prefix + "2.bias", out=out)
        if self.kind == "bilinear" and self.upsc != 2:
            raise NotImplementedError("the HIP up-sampling stream is specialised for scale factor 2")
        y = rt.conv(h, prefix + "2.weight", stats="instance")
        return rt.in_up_relu(y, out=out, upsample=self.kind == "bilinear")


def make_decoder_block(in_channels, middle_channels, out_channels, if_convtrans=False, noupsam=False, upsc=2):
    return DecoderBlock(in_channels, middle_channels, out_channels, if_convtrans=if_convtrans, noupsam=noupsam, upsc=upsc)


def _encoder_stage(first, c_in, widths):
    mods = OrderedDict()
    idx = first
    if first:
        mods[str(idx)] = nn.MaxPool2d(kernel_size=2, stride=2)
        idx += 1
    for width in widths:
        mods[str(idx)] = nn.Conv2d(c_in, width, kernel_size=3, padding=1)
        mods[str(idx + 1)] = nn.BatchNorm2d(width)
        mods[str(idx + 2)] = nn.ReLU(inplace=True)
        c_in, idx = width, idx + 3
    return nn.Sequential(mods)


class _UNetVggBase(SegkNet):
    def _build(self, out_classes, if_convtrans, in_classes):
        if in_classes is not None:
            self.conv0 = nn.Conv2d(in_classes, 3, 3, padding=1)
        c = 3
        for k, (first, widths) in enumerate(_ENCODER, start=1):
            setattr(self, f"conv{k}", _encoder_stage(first, c, widths))
            c = widths[-1]
        for name, (ci, cm, co) in _DECODER.items():
            block = make_decoder_block(ci, cm, co, if_convtrans=if_convtrans)
            setattr(self, name, nn.Sequential(nn.MaxPool2d(kernel_size=2, stride=2), block) if name == "center" else block)
        self.dec1 = nn.Sequential(nn.Conv2d(32 + 64, 32, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(32, out_classes, kernel_size=1)

    # -------------------------------------------------------------- CPU tensors: plain torch.nn
    def _forward_host(self, x):
        if hasattr(self, "conv0"):
            x = self.conv0(x)
        skips = []
        for k in range(1, 6):
            x = getattr(self, f"conv{k}")(x)
            skips.append(x)
        x = self.center(x)
        for k in (5, 4, 3, 2, 1):
            x = getattr(self, f"dec{k}")(torch.cat([x, skips[k - 1]], 1))
        return self.final(x)

    # -------------------------------------------------------------- CUDA tensors: segk tape
    def _emit(self, rt, x):
        N, _, H, W = x.shape
        if H % 32 or W % 32:
            raise RuntimeError(f"UNetVgg needs H and W divisible by 32 (five 2x2 pools), got {H}x{W}")
        # concat buffers of models/unetvgg.py:99-103: [decoder output | encoder skip], written in place
        widths = {k: (_DECODER[f"dec{k + 1}" if k < 5 else "center"][2], _ENCODER[k - 1][1][-1]) for k in range(1, 6)}
        cat = {k: rt.new(N, H >> (k - 1), W >> (k - 1), sum(widths[k])).buf for k in range(1, 6)}
        if hasattr(self, "conv0"):
            v = rt.conv_first(x, "conv0.weight", "conv0.bias")          # 3 channels stored as 8
        else:
            v = None
        skips = {}
        pooled = None
        for k, (first, stage_widths) in enumerate(_ENCODER, start=1):
            idx = first + (1 if first else 0)
            if first:
                v = pooled           # the max pool that opens this stage ran fused with the previous stage's last norm
            for j, _ in enumerate(stage_widths):
                wn, bn_ = f"conv{k}.{idx}", f"conv{k}.{idx + 1}"
                if k == 1 and j == 0 and v is None:
                    y = rt.conv_first(x, wn + ".weight", wn + ".bias", bias_grad_zero=self.training, stats=True)
                else:
                    y = rt.conv(v, wn + ".weight", wn + ".bias", bias_grad_zero=self.training, stats=True)
                if j == len(stage_widths) - 1:   # skip tensor (second slice of the concat buffer) + pool for the next stage
                    v, pooled = rt.bn_relu(y, bn_, out=Act(cat[k], widths[k][0], widths[k][1]), pool=True)
                else:
                    v = rt.bn_relu(y, bn_)
                idx += 3
            skips[k] = v
        d = self.center[1].emit(rt, "center.1.", pooled, out=Act(cat[5], 0, widths[5][0]))
        for k in (5, 4, 3, 2):
            block = getattr(self, f"dec{k}")
            d = block.emit(rt, f"dec{k}.", [d, skips[k]], out=Act(cat[k - 1], 0, widths[k - 1][0]))
        e = rt.conv([d, skips[1]], "dec1.0.weight", "dec1.0.bias", relu=True)
        return rt.head(e, "final.weight", "final.bias")


class UNetVgg(_UNetVggBase):
    def __init__(self, pretrained=False, out_classes=3, patch_train=False, if_convtrans=False):
        super().__init__()
        if pretrained:
            raise RuntimeError("pretrained VGG16-BN weights need a download, which is unavailable here; "
                               "load a checkpoint with load_state_dict instead")
        self.patch_train = patch_train
        self._build(out_classes, if_convtrans, None)


class UNetVggInOut(_UNetVggBase):
    """Variant with a learned in_classes -> 3 convolution in front (reference :158-213)."""

    def __init__(self, pretrained=False, in_classes=1, out_classes=3, patch_train=False):
        super().__init__()
        if pretrained:
            raise RuntimeError("pretrained VGG16-BN weights need a download, which is unavailable here")
        self.patch_train = patch_train
        self._build(out_classes, False, in_classes)
