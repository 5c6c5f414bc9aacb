"""U-Net with a VGG16-BN encoder, MI355X-native.

Same constructor signatures, attribute names and state_dict layout as the reference classes
(/root/reference/models/unetvgg.py: make_decoder_block :6-41, UNetVgg :52-105, UNetVgg256 :107-132, UNetVggWholeImg :134-155,
UNetVggInOut :158-213, UNetVggTwo :216-265,
make_encoder_block :268-323, make_center :335-372, UNetNaive :373-440, UNetWithoutCenter :443-506,
UNetNaiveMultiOut :509-574, UNetLimMultiOut :576-648, UNetSimple :651-728, UNetSkip :730-813,
UNetSimpleReduced :815-894, SpiderNet :896-993, UNetTetra :996-1095),
so checkpoints and callers are interchangeable.  The module tree is made of ordinary torch.nn layers
that only hold the parameters (and serve CPU tensors); for CUDA tensors `forward` hands the
parameters to the segk tape (segk/engine.py), which launches the HIP kernels: implicit-GEMM
convolutions on the matrix cores, fused BatchNorm/InstanceNorm/ReLU/pool/up-sample streams, with the
skip concatenations of :99-103 written in place into shared NHWC buffers.
"""
from collections import OrderedDict

import os

import torch
import torch.nn.functional as F
from torch import nn

from segk.engine import Var
from segk.net import SegkBlock, SegkNet
from segk.ops import Act

# VGG configuration D: first feature index and conv widths of the five encoder stages; stages 2-5 start
# with the 2x2 max pool that precedes them in torchvision's `features` list, whose numbering the
# reference keeps by slicing that list (unetvgg.py:72-76).
_INPLACE_CONCAT = bool(os.environ.get("SEGK_INPLACE_CONCAT"))   # experiment: [decoder output | skip] as slices of one buffer
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
        y = rt.conv(h, prefix + "2.weight", stats="instance")
        return rt.in_up_relu(y, out=out, upsample=self.kind == "bilinear", scale=self.upsc)


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
        if _INPLACE_CONCAT:
            widths = {k: (_DECODER[f"dec{k + 1}" if k < 5 else "center"][2], _ENCODER[k - 1][1][-1]) for k in range(1, 6)}
            cat = {k: rt.new(N, H >> (k - 1), W >> (k - 1), sum(widths[k])).buf for k in range(1, 6)}
        # torch.cat of models/unetvgg.py:99-103: never materialised.  The decoder output and the encoder skip stay two
        # tensors of their own (every producer writes whole, line-aligned pixels) and the convolution that reads the
        # concatenation takes them as its two input segments; its data gradient writes each segment's gradient tensor.
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
                if k == 1 and j == 0 and v is None and rt.can_fuse_first_norm(x, stage_widths[0]):
                    v = rt.conv_first_bn_relu(x, wn + ".weight", wn + ".bias", bn_)      # output of the convolution never stored
                    idx += 3
                    continue
                if k == 1 and j == 0 and v is None:
                    y = rt.conv_first(x, wn + ".weight", wn + ".bias", bias_grad_zero=self.training, stats=True)
                else:
                    y = rt.conv(v, wn + ".weight", wn + ".bias", bias_grad_zero=self.training, stats=True)
                if j == len(stage_widths) - 1:   # skip tensor (second slice of the concat buffer) + pool for the next stage
                    v, pooled = rt.bn_relu(y, bn_, out=Act(cat[k], widths[k][0], widths[k][1]) if _INPLACE_CONCAT else None, pool=True)
                else:
                    v = rt.bn_relu(y, bn_, one_reader=True)        # read by the stage's next convolution only
                idx += 3
            skips[k] = v
        if _INPLACE_CONCAT:
            d = self.center[1].emit(rt, "center.1.", pooled, out=Act(cat[5], 0, widths[5][0]))
            for k in (5, 4, 3, 2):
                block = getattr(self, f"dec{k}")
                d = block.emit(rt, f"dec{k}.", [d, skips[k]], out=Act(cat[k - 1], 0, widths[k - 1][0]))
            e = rt.conv([d, skips[1]], "dec1.0.weight", "dec1.0.bias", relu=True)
            return rt.head(e, "final.weight", "final.bias")
        d = self.center[1].emit(rt, "center.1.", pooled)
        for k in (5, 4, 3, 2):
            block = getattr(self, f"dec{k}")
            d = block.emit(rt, f"dec{k}.", (d, skips[k]))
        e = rt.conv((d, skips[1]), "dec1.0.weight", "dec1.0.bias", relu=True)
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


class EncoderBlock(nn.Sequential):
    """Contraction unit of the plain U-Nets (reference make_encoder_block :268-323): a down-sampling layer, then two or
    three conv3x3(bias=False) + InstanceNorm2d + ReLU groups; child indices as in the reference."""

    def __init__(self, in_channels, middle_channels, out_channels, convs3=True, middle_channels2=None, max_pool=False,
                 no_dsam=False, if_interpolate=False, downsc=2):
        if if_interpolate:
            down, kind = nn.Upsample(scale_factor=downsc, mode="bilinear", align_corners=False), "interpolate"
        elif no_dsam:
            down, kind = nn.Conv2d(in_channels, in_channels, kernel_size=3, padding=1), "conv3"
        elif max_pool:
            down, kind = nn.MaxPool2d(kernel_size=2, stride=2), "pool"
        else:
            down, kind = nn.Conv2d(in_channels, in_channels, kernel_size=2, stride=2), "conv2"
        widths = [middle_channels] + ([middle_channels2] if convs3 else []) + [out_channels]
        layers, c = [down], in_channels
        for width in widths:
            layers += [nn.Conv2d(c, width, 3, padding=1, bias=False), nn.InstanceNorm2d(width), nn.ReLU(inplace=True)]
            c = width
        super().__init__(*layers)
        self.kind, self.groups, self.downsc = kind, len(widths), downsc

    def emit(self, rt, prefix, x, out=None):
        """Append this block to a segk tape; the last group writes into `out` (a slice of a concat buffer)."""
        if self.kind == "conv2":
            h = rt.conv_down2(x, prefix + "0.weight", prefix + "0.bias")
        elif self.kind == "conv3":
            h = rt.conv(x, prefix + "0.weight", prefix + "0.bias")
        elif self.kind == "pool":
            h = rt.pool(x)
        else:
            # SpiderNet's branches (reference :925-941) re-scale the network INPUT: `x` is the NCHW image here.  Resizing a
            # one-channel image that needs no gradient is boundary plumbing (a torch op), like the layout conversion.
            assert torch.is_tensor(x), "an interpolating encoder block takes the input image"
            h = F.interpolate(x.detach().float(), scale_factor=self.downsc, mode="bilinear", align_corners=False)
        for j in range(self.groups):
            if torch.is_tensor(h):
                y = rt.conv_first(h, f"{prefix}1.weight", None, stats="instance")
            else:
                y = rt.conv(h, f"{prefix}{1 + 3 * j}.weight", stats="instance")
            h = rt.in_up_relu(y, upsample=False, out=out if j == self.groups - 1 else None)
        return h


def make_encoder_block(in_channels, middle_channels, out_channels, convs3=True, middle_channels2=None, max_pool=False,
                       no_dsam=False, if_interpolate=False, downsc=2):
    return EncoderBlock(in_channels, middle_channels, out_channels, convs3=convs3, middle_channels2=middle_channels2,
                        max_pool=max_pool, no_dsam=no_dsam, if_interpolate=if_interpolate, downsc=downsc)


def _emit_stem(rt, x, out):
    """conv1 of the InstanceNorm U-Nets (reference :395-402, :676-683): two conv3x3(bias=False) + InstanceNorm + ReLU groups
    on the NCHW image; the second writes into `out`."""
    y = rt.conv_first(x, "conv1.0.weight", None, stats="instance")
    h = rt.in_up_relu(y, upsample=False)
    y = rt.conv(h, "conv1.3.weight", stats="instance")
    return rt.in_up_relu(y, upsample=False, out=out)


def _emit_bottom(rt, c4, out, prefix="conv5"):
    """conv5 of reference :471-480 / :690-699: conv, InstanceNorm (no ReLU), conv, ReLU, conv, InstanceNorm, ReLU.
    A bias in front of an InstanceNorm cancels: its gradient is exactly zero."""
    y = rt.conv(c4, f"{prefix}.0.weight", f"{prefix}.0.bias", stats="instance", bias_grad_zero=True)
    h = rt.in_up_relu(y, upsample=False, relu=False)
    h = rt.conv(h, f"{prefix}.2.weight", f"{prefix}.2.bias", relu=True)
    y = rt.conv(h, f"{prefix}.4.weight", f"{prefix}.4.bias", stats="instance", bias_grad_zero=True)
    return rt.in_up_relu(y, upsample=False, out=out)


class CenterBlock(nn.Sequential):
    """Bottleneck of UNetNaive (reference make_center :335-372): down-sampling, conv3x3 + ReLU, then either a transposed
    convolution or conv3x3 + bilinear x2, and a ReLU; widths are the reference's literals."""

    def __init__(self, in_channels, max_pool=True, iftrans2d=False):
        down = nn.MaxPool2d(kernel_size=2, stride=2) if max_pool else nn.Conv2d(in_channels, in_channels, kernel_size=2, stride=2)
        layers = [down, nn.Conv2d(512, 512, kernel_size=(3, 3), padding=(1, 1)), nn.ReLU(inplace=True)]
        if iftrans2d:
            layers += [nn.ConvTranspose2d(512, 256, kernel_size=(4, 4), stride=(2, 2), padding=1)]
        else:
            layers += [nn.Conv2d(512, 256, 3, padding=1), nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)]
        super().__init__(*layers, nn.ReLU(inplace=True))
        self.max_pool, self.iftrans2d = max_pool, iftrans2d

    def emit(self, rt, prefix, x, out=None):
        h = rt.pool(x) if self.max_pool else rt.conv_down2(x, prefix + "0.weight", prefix + "0.bias")
        h = rt.conv(h, prefix + "1.weight", prefix + "1.bias", relu=True)
        if self.iftrans2d:
            return rt.conv_transpose_relu(h, prefix + "3.weight", prefix + "3.bias", out=out)
        y = rt.conv(h, prefix + "3.weight", prefix + "3.bias")
        return rt.up(y, out=out, align=False, relu=True)


def make_center(in_channels, max_pool=True, iftrans2d=False):
    return CenterBlock(in_channels, max_pool=max_pool, iftrans2d=iftrans2d)


def _stem_modules(in_classes, width):
    return nn.Sequential(
        nn.Conv2d(in_classes, width, 3, padding=1, bias=False), nn.InstanceNorm2d(width), nn.ReLU(inplace=True),
        nn.Conv2d(width, width, 3, padding=1, bias=False), nn.InstanceNorm2d(width), nn.ReLU(inplace=True))


class UNetNaive(SegkNet):
    """Five-level InstanceNorm U-Net with a bottleneck (reference :373-440)."""

    def __init__(self, pretrained=False, out_classes=3, patch_train=False, in_classes=1, if_convtrans=False, ifmaxpool=False):
        super().__init__()
        self.patch_train = patch_train
        self.conv1 = _stem_modules(in_classes, 64)
        self.conv2 = make_encoder_block(64, 128, 128, convs3=False)
        self.conv3 = make_encoder_block(128, 256, 256, middle_channels2=256)
        self.conv4 = make_encoder_block(256, 512, 512, middle_channels2=512)
        self.conv5 = make_encoder_block(512, 512, 512, middle_channels2=512)
        self.center = make_center(in_channels=512, max_pool=ifmaxpool, iftrans2d=if_convtrans)
        self.dec5 = make_decoder_block(256 + 512, 512, 256)
        self.dec4 = make_decoder_block(256 + 512, 512, 256)
        self.dec3 = make_decoder_block(256 + 256, 256, 64)
        self.dec2 = make_decoder_block(64 + 128, 128, 32)
        self.dec1 = nn.Sequential(nn.Conv2d(32 + 64, 32, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(32, out_classes, kernel_size=1)

    def _forward_host(self, x):
        skips = [self.conv1(x)]
        for k in range(2, 6):
            skips.append(getattr(self, f"conv{k}")(skips[-1]))
        d = self.center(skips[-1])
        for k in (5, 4, 3, 2, 1):
            d = getattr(self, f"dec{k}")(torch.cat([d, skips[k - 1]], 1))
        return self.final(d)

    def _emit(self, rt, x):
        N, _, H, W = x.shape
        if H % 32 or W % 32:
            raise RuntimeError(f"UNetNaive needs H and W divisible by 32 (five halvings), got {H}x{W}")
        dec_w = {1: 32, 2: 64, 3: 256, 4: 256, 5: 256}          # width of the decoder tensor that leads concat buffer k
        enc_w = {1: 64, 2: 128, 3: 256, 4: 512, 5: 512}
        cat = {k: rt.new(N, H >> (k - 1), W >> (k - 1), dec_w[k] + enc_w[k]).buf for k in range(1, 6)}
        skips = {1: _emit_stem(rt, x, Act(cat[1], dec_w[1], enc_w[1]))}
        for k in range(2, 6):
            skips[k] = getattr(self, f"conv{k}").emit(rt, f"conv{k}.", skips[k - 1], out=Act(cat[k], dec_w[k], enc_w[k]))
        d = self.center.emit(rt, "center.", skips[5], out=Act(cat[5], 0, dec_w[5]))
        for k in (5, 4, 3, 2):
            d = getattr(self, f"dec{k}").emit(rt, f"dec{k}.", [d, skips[k]], out=Act(cat[k - 1], 0, dec_w[k - 1]))
        e = rt.conv([d, skips[1]], "dec1.0.weight", "dec1.0.bias", relu=True)
        return rt.head(e, "final.weight", "final.bias")


class UNetWithoutCenter(SegkNet):
    """Four-level InstanceNorm U-Net whose deepest stage stays at 1/8 resolution (reference :443-506)."""

    def __init__(self, pretrained=False, out_classes=3, patch_train=False, in_classes=1):
        super().__init__()
        self.patch_train = patch_train
        self.conv1 = _stem_modules(in_classes, 64)
        self.conv2 = make_encoder_block(64, 128, 128, convs3=False)
        self.conv3 = make_encoder_block(128, 256, 256, middle_channels2=256)
        self.conv4 = make_encoder_block(256, 512, 512, middle_channels2=512)
        self.conv5 = nn.Sequential(
            nn.Conv2d(512, 512, kernel_size=3, padding=1), nn.InstanceNorm2d(512),
            nn.Conv2d(512, 512, kernel_size=(3, 3), padding=(1, 1)), nn.ReLU(inplace=True),
            nn.Conv2d(512, 256, kernel_size=3, padding=1), nn.InstanceNorm2d(256), nn.ReLU(inplace=True))
        self.dec4 = make_decoder_block(256 + 512, 512, 256)
        self.dec3 = make_decoder_block(256 + 256, 256, 64)
        self.dec2 = make_decoder_block(64 + 128, 128, 32)
        self.dec1 = nn.Sequential(nn.Conv2d(32 + 64, 32, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(32, out_classes, kernel_size=1)

    def _forward_host(self, x):
        skips = [self.conv1(x)]
        for k in range(2, 5):
            skips.append(getattr(self, f"conv{k}")(skips[-1]))
        d = self.conv5(skips[-1])
        for k in (4, 3, 2, 1):
            d = getattr(self, f"dec{k}")(torch.cat([d, skips[k - 1]], 1))
        return self.final(d)

    def _emit(self, rt, x):
        return _emit_unet4(self, rt, x, {1: 64, 2: 128, 3: 256, 4: 512}, {1: 32, 2: 64, 3: 256, 4: 256}, _emit_bottom)[0]


def _emit_bottom_plain(rt, c4, out):
    """conv5 of reference :539-548 / :611-620: two conv3x3(bias=False) + InstanceNorm + ReLU groups."""
    y = rt.conv(c4, "conv5.0.weight", stats="instance")
    h = rt.in_up_relu(y, upsample=False)
    y = rt.conv(h, "conv5.3.weight", stats="instance")
    return rt.in_up_relu(y, upsample=False, out=out)


def _emit_unet4(net, rt, x, enc_w, dec_w, bottom):
    """Shared body of the four-level InstanceNorm U-Nets.  enc_w[k] / dec_w[k]: widths of the encoder skip and of the
    decoder tensor that share concat buffer k (1 = full resolution; the torch.cat of reference :501-504 in place).
    -> (logits, encoder skips {1..4}, decoder tensors {5: conv5, 4: dec4, 3: dec3, 2: dec2, 1: dec1})"""
    N, _, H, W = x.shape
    if H % 8 or W % 8:
        raise RuntimeError(f"{type(net).__name__} needs H and W divisible by 8 (three stride-2 convolutions), got {H}x{W}")
    cat = {k: rt.new(N, H >> (k - 1), W >> (k - 1), dec_w[k] + enc_w[k]).buf for k in range(1, 5)}
    skips = {1: _emit_stem(rt, x, Act(cat[1], dec_w[1], enc_w[1]))}
    for k in range(2, 5):
        skips[k] = getattr(net, f"conv{k}").emit(rt, f"conv{k}.", skips[k - 1], out=Act(cat[k], dec_w[k], enc_w[k]))
    dec = {5: bottom(rt, skips[4], Act(cat[4], 0, dec_w[4]))}
    for k in (4, 3, 2):
        dec[k] = getattr(net, f"dec{k}").emit(rt, f"dec{k}.", [dec[k + 1], skips[k]], out=Act(cat[k - 1], 0, dec_w[k - 1]))
    dec[1] = rt.conv([dec[2], skips[1]], "dec1.0.weight", "dec1.0.bias", relu=True)
    return rt.head(dec[1], "final.weight", "final.bias"), skips, dec


class UNetNaiveMultiOut(SegkNet):
    """UNetWithoutCenter with a bias-free bottom stage; also returns [conv3, dec4] (reference :509-574)."""

    def __init__(self, pretrained=False, out_classes=3, patch_train=False, in_classes=1):
        super().__init__()
        self.patch_train = patch_train
        self.conv1 = _stem_modules(in_classes, 64)
        self.conv2 = make_encoder_block(64, 128, 128, convs3=False)
        self.conv3 = make_encoder_block(128, 256, 256, middle_channels2=256)
        self.conv4 = make_encoder_block(256, 512, 512, middle_channels2=512)
        self.conv5 = nn.Sequential(
            nn.Conv2d(512, 512, kernel_size=3, padding=1, bias=False), nn.InstanceNorm2d(512), nn.ReLU(inplace=True),
            nn.Conv2d(512, 256, kernel_size=3, padding=1, bias=False), nn.InstanceNorm2d(256), nn.ReLU(inplace=True))
        self.dec4 = make_decoder_block(256 + 512, 512, 256)
        self.dec3 = make_decoder_block(256 + 256, 256, 64)
        self.dec2 = make_decoder_block(64 + 128, 128, 32)
        self.dec1 = nn.Sequential(nn.Conv2d(32 + 64, 32, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(32, out_classes, kernel_size=1)

    def _forward_host(self, x):
        skips = [self.conv1(x)]
        for k in range(2, 5):
            skips.append(getattr(self, f"conv{k}")(skips[-1]))
        dec4 = self.dec4(torch.cat([self.conv5(skips[-1]), skips[3]], 1))
        d = self.dec3(torch.cat([dec4, skips[2]], 1))
        d = self.dec2(torch.cat([d, skips[1]], 1))
        return self.final(self.dec1(torch.cat([d, skips[0]], 1))), [skips[2], dec4]

    def _emit(self, rt, x):
        logits, skips, dec = _emit_unet4(self, rt, x, {1: 64, 2: 128, 3: 256, 4: 512}, {1: 32, 2: 64, 3: 256, 4: 256}, _emit_bottom_plain)
        return logits, [rt.export(skips[3]), rt.export(dec[4])]


class UNetLimMultiOut(SegkNet):
    """Four-level U-Net whose layers all have `channel_depth` channels; with multiout also returns conv3 (reference :576-648)."""

    def __init__(self, pretrained=False, out_classes=3, patch_train=False, in_classes=1, channel_depth=16, multiout=False):
        super().__init__()
        cd = channel_depth
        self.multiout, self.patch_train, self.channel_depth = multiout, patch_train, cd
        # InstanceNorm2d widths are the reference's literals; without affine parameters the width is unused
        self.conv1 = nn.Sequential(
            nn.Conv2d(in_classes, cd, 3, padding=1, bias=False), nn.InstanceNorm2d(64), nn.ReLU(inplace=True),
            nn.Conv2d(cd, cd, 3, padding=1, bias=False), nn.InstanceNorm2d(64), nn.ReLU(inplace=True))
        self.conv2 = make_encoder_block(cd, cd, cd, convs3=False)
        self.conv3 = make_encoder_block(cd, cd, cd, middle_channels2=cd)
        self.conv4 = make_encoder_block(cd, cd, cd, middle_channels2=cd)
        self.conv5 = nn.Sequential(
            nn.Conv2d(cd, cd, kernel_size=3, padding=1, bias=False), nn.InstanceNorm2d(cd), nn.ReLU(inplace=True),
            nn.Conv2d(cd, cd // 2, kernel_size=3, padding=1, bias=False), nn.InstanceNorm2d(cd), nn.ReLU(inplace=True))
        self.dec4 = make_decoder_block(cd + cd // 2, cd, cd // 2)
        self.dec3 = make_decoder_block(cd + cd // 2, cd, cd // 2)
        self.dec2 = make_decoder_block(cd + cd // 2, cd, cd // 2)
        self.dec1 = nn.Sequential(nn.Conv2d(cd + cd // 2, cd, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(cd, out_classes, kernel_size=1)

    def _forward_host(self, x):
        skips = [self.conv1(x)]
        for k in range(2, 5):
            skips.append(getattr(self, f"conv{k}")(skips[-1]))
        d = self.conv5(skips[-1])
        for k in (4, 3, 2, 1):
            d = getattr(self, f"dec{k}")(torch.cat([d, skips[k - 1]], 1))
        return (self.final(d), skips[2]) if self.multiout else self.final(d)

    def _pack_outputs(self, logits, feats):
        return logits, feats[0]

    def _emit(self, rt, x):
        cd = self.channel_depth
        if cd % 16:
            raise NotImplementedError("channel_depth must be a multiple of 16 (8-channel storage groups)")
        logits, skips, _ = _emit_unet4(self, rt, x, {k: cd for k in range(1, 5)}, {k: cd // 2 for k in range(1, 5)}, _emit_bottom_plain)
        return (logits, [rt.export(skips[3])]) if self.multiout else logits


class UNetSimple(SegkNet):
    """Four-level U-Net with InstanceNorm and learned stride-2 down-sampling (reference :651-728); `pretrained` is
    accepted and unused, as there."""
    _export_names = ("d1", "d2", "d3", "d4")         # multiout: [dec1, dec2, dec3, dec4] (reference :726)

    def __init__(self, pretrained=False, out_classes=3, patch_train=False, in_classes=1, channelscale=64, multiout=False,
                 noupdownsam=False):
        super().__init__()
        cs = channelscale
        self.multiout, self.patch_train, self.channelscale, self.noupdownsam = multiout, patch_train, cs, noupdownsam
        self.conv1 = nn.Sequential(
            nn.Conv2d(in_classes, cs, 3, padding=1, bias=False), nn.InstanceNorm2d(cs), nn.ReLU(inplace=True),
            nn.Conv2d(cs, cs, 3, padding=1, bias=False), nn.InstanceNorm2d(cs), nn.ReLU(inplace=True))
        self.conv2 = make_encoder_block(cs, 2 * cs, 2 * cs, convs3=False, no_dsam=noupdownsam)
        self.conv3 = make_encoder_block(2 * cs, 4 * cs, 4 * cs, middle_channels2=4 * cs, no_dsam=noupdownsam)
        self.conv4 = make_encoder_block(4 * cs, 8 * cs, 8 * cs, middle_channels2=8 * cs, no_dsam=noupdownsam)
        # the two InstanceNorm2d widths are the reference's literals; without affine parameters the width is unused
        self.conv5 = nn.Sequential(
            nn.Conv2d(8 * cs, 8 * cs, kernel_size=3, padding=1), nn.InstanceNorm2d(512),
            nn.Conv2d(8 * cs, 8 * cs, kernel_size=3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(8 * cs, 8 * cs, kernel_size=3, padding=1), nn.InstanceNorm2d(4 * cs), nn.ReLU(inplace=True))
        self.dec4 = make_decoder_block(16 * cs, 8 * cs, 4 * cs, noupsam=noupdownsam)
        self.dec3 = make_decoder_block(8 * cs, 4 * cs, cs, noupsam=noupdownsam)
        self.dec2 = make_decoder_block(3 * cs, 2 * cs, cs // 2, noupsam=noupdownsam)
        self.dec1 = nn.Sequential(nn.Conv2d(int(1.5 * cs), cs // 2, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(cs // 2, out_classes, kernel_size=1)

    def _forward_host(self, x):
        conv1 = self.conv1(x)
        conv2 = self.conv2(conv1)
        conv3 = self.conv3(conv2)
        conv4 = self.conv4(conv3)
        conv5 = self.conv5(conv4)
        dec4 = self.dec4(torch.cat([conv5, conv4], 1))
        dec3 = self.dec3(torch.cat([dec4, conv3], 1))
        dec2 = self.dec2(torch.cat([dec3, conv2], 1))
        dec1 = self.dec1(torch.cat([dec2, conv1], 1))
        if self.multiout:
            return self.final(dec1), [dec1, dec2, dec3, dec4]
        return self.final(dec1)

    def _emit(self, rt, x):
        N, _, H, W = x.shape
        cs, down = self.channelscale, not self.noupdownsam
        if cs % 16:
            raise NotImplementedError("channelscale must be a multiple of 16 (8-channel storage groups)")
        if down and (H % 8 or W % 8):
            raise RuntimeError(f"UNetSimple needs H and W divisible by 8 (three stride-2 convolutions), got {H}x{W}")

        def level(k, C):
            return rt.new(N, H >> k if down else H, W >> k if down else W, C).buf
        # concat buffers of reference :720-723, written in place
        cat1, cat2 = level(0, cs // 2 + cs), level(1, cs + 2 * cs)            # [dec2 | conv1], [dec3 | conv2]
        cat3, cat4 = level(2, 4 * cs + 4 * cs), level(3, 8 * cs + 8 * cs)     # [dec4 | conv3], [conv5 | conv4]
        c1 = _emit_stem(rt, x, Act(cat1, cs // 2, cs))
        c2 = self.conv2.emit(rt, "conv2.", c1, out=Act(cat2, cs, 2 * cs))
        c3 = self.conv3.emit(rt, "conv3.", c2, out=Act(cat3, 4 * cs, 4 * cs))
        c4 = self.conv4.emit(rt, "conv4.", c3, out=Act(cat4, 8 * cs, 8 * cs))
        c5 = _emit_bottom(rt, c4, Act(cat4, 0, 8 * cs))
        d4 = self.dec4.emit(rt, "dec4.", [c5, c4], out=Act(cat3, 0, 4 * cs))
        d3 = self.dec3.emit(rt, "dec3.", [d4, c3], out=Act(cat2, 0, cs))
        d2 = self.dec2.emit(rt, "dec2.", [d3, c2], out=Act(cat1, 0, cs // 2))
        e = rt.conv([d2, c1], "dec1.0.weight", "dec1.0.bias", relu=True)
        logits = rt.head(e, "final.weight", "final.bias")
        tensors = {"d1": e, "d2": d2, "d3": d3, "d4": d4}
        return (logits, [rt.export(tensors[k]) for k in self._export_names]) if self.multiout else logits


class UNetSkip(UNetSimple):
    """UNetSimple with channelscale 32 by default; with multiout it returns (logits, dec4) (reference :730-813).  The
    reference also resizes three crops of the input (:795,:804,:807) and never uses them; that dead work is left out."""

    def __init__(self, pretrained=False, out_classes=3, patch_train=False, in_classes=1, channelscale=32, multiout=False,
                 noupdownsam=False):
        super().__init__(pretrained=pretrained, out_classes=out_classes, patch_train=patch_train, in_classes=in_classes,
                         channelscale=channelscale, multiout=multiout, noupdownsam=noupdownsam)
        self.xcontr = 3

    def _forward_host(self, x):
        out = super()._forward_host(x)
        return (out[0], out[1][3]) if self.multiout else out

    _export_names = ("d4",)

    def _pack_outputs(self, logits, feats):
        return logits, feats[0]


class UNetSimpleReduced(SegkNet):
    """UNetSimple without its two deepest stages (reference :815-894).  multiout=True names tensors that do not exist in
    the reference's forward (:892) and raises there; it raises here as well."""

    def __init__(self, pretrained=False, out_classes=3, patch_train=False, in_classes=1, channelscale=64, multiout=False,
                 noupdownsam=False):
        super().__init__()
        cs = channelscale
        self.multiout, self.patch_train, self.channelscale, self.noupdownsam = multiout, patch_train, cs, noupdownsam
        self.conv1 = _stem_modules(in_classes, cs)
        self.conv2 = make_encoder_block(cs, 2 * cs, 2 * cs, convs3=False, no_dsam=noupdownsam)
        self.conv3 = make_encoder_block(2 * cs, 4 * cs, 8 * cs, middle_channels2=4 * cs, no_dsam=noupdownsam)
        self.dec4 = make_decoder_block(8 * cs, 8 * cs, 4 * cs, noupsam=True)
        self.dec3 = make_decoder_block(12 * cs, 4 * cs, cs, noupsam=noupdownsam)
        self.dec2 = make_decoder_block(3 * cs, 2 * cs, cs // 2, noupsam=noupdownsam)
        self.dec1 = nn.Sequential(nn.Conv2d(int(1.5 * cs), cs // 2, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(cs // 2, out_classes, kernel_size=1)

    def _check(self):
        if self.multiout:
            raise NameError("name 'conv4' is not defined")       # what the reference's forward does at :892

    def _forward_host(self, x):
        conv1 = self.conv1(x)
        conv2 = self.conv2(conv1)
        conv3 = self.conv3(conv2)
        dec3 = self.dec3(torch.cat([self.dec4(conv3), conv3], 1))
        dec2 = self.dec2(torch.cat([dec3, conv2], 1))
        out = self.final(self.dec1(torch.cat([dec2, conv1], 1)))
        self._check()
        return out

    def _emit(self, rt, x):
        self._check()
        N, _, H, W = x.shape
        cs, down = self.channelscale, not self.noupdownsam
        if cs % 16:
            raise NotImplementedError("channelscale must be a multiple of 16 (8-channel storage groups)")
        if down and (H % 4 or W % 4):
            raise RuntimeError(f"UNetSimpleReduced needs H and W divisible by 4 (two stride-2 convolutions), got {H}x{W}")

        def level(k, C):
            return rt.new(N, H >> k if down else H, W >> k if down else W, C).buf
        cat1, cat2, cat3 = level(0, cs // 2 + cs), level(1, cs + 2 * cs), level(2, 4 * cs + 8 * cs)
        c1 = _emit_stem(rt, x, Act(cat1, cs // 2, cs))
        c2 = self.conv2.emit(rt, "conv2.", c1, out=Act(cat2, cs, 2 * cs))
        c3 = self.conv3.emit(rt, "conv3.", c2, out=Act(cat3, 4 * cs, 8 * cs))
        d4 = self.dec4.emit(rt, "dec4.", c3, out=Act(cat3, 0, 4 * cs))
        d3 = self.dec3.emit(rt, "dec3.", [d4, c3], out=Act(cat2, 0, cs))
        d2 = self.dec2.emit(rt, "dec2.", [d3, c2], out=Act(cat1, 0, cs // 2))
        e = rt.conv([d2, c1], "dec1.0.weight", "dec1.0.bias", relu=True)
        return rt.head(e, "final.weight", "final.bias")


class SpiderNet(SegkNet):
    """Five branches that see the input at 2x, 1x, 1/2, 1/4 and 1/8 resolution, each encoded and decoded back to full
    resolution, concatenated in front of a 1x1 head (reference :896-993).  Every branch re-scales the input image itself;
    no tensor is shared between branches before the head."""
    _BRANCHES = ((4, 3), (3, 2), (2, 1), (1, None), (0, -1))       # (index k of conv{k}/dec{k}, log2 of the down-scaling)

    def __init__(self, out_classes=3, patch_train=False, in_classes=1, channelscale=64, multiout=False, noupdownsam=False):
        super().__init__()
        cs = channelscale
        self.multiout, self.patch_train, self.channelscale, self.noupdownsam = multiout, patch_train, cs, noupdownsam
        enc = dict(no_dsam=noupdownsam, if_interpolate=True)
        self.conv0 = make_encoder_block(in_classes, cs // 2, cs // 2, convs3=False, downsc=2, **enc)
        self.conv1 = _stem_modules(in_classes, cs)
        self.conv2 = make_encoder_block(in_classes, 2 * cs, 2 * cs, convs3=False, downsc=0.5, **enc)
        self.conv3 = make_encoder_block(in_classes, 4 * cs, 4 * cs, middle_channels2=4 * cs, downsc=0.25, **enc)
        self.conv4 = make_encoder_block(in_classes, 8 * cs, 8 * cs, middle_channels2=8 * cs, downsc=1 / 8, **enc)
        # never used by forward (reference :977 is commented out); kept for the state_dict layout
        self.conv5 = nn.Sequential(
            nn.Conv2d(cs + 2 * cs + 4 * cs + 8 * cs, 8 * cs, kernel_size=3, padding=1), nn.InstanceNorm2d(512),
            nn.Conv2d(12 * cs, 8 * cs, kernel_size=(3, 3), padding=(1, 1)), nn.ReLU(inplace=True),
            nn.Conv2d(8 * cs, 8 * cs, kernel_size=3, padding=1), nn.InstanceNorm2d(8 * cs), nn.ReLU(inplace=True))
        self.dec4 = make_decoder_block(8 * cs, 8 * cs, 4 * cs, noupsam=noupdownsam, upsc=8)
        self.dec3 = make_decoder_block(4 * cs, 4 * cs, cs, noupsam=noupdownsam, upsc=4)
        self.dec2 = make_decoder_block(2 * cs, 2 * cs, cs // 2, noupsam=noupdownsam, upsc=2)
        self.dec1 = nn.Sequential(nn.Conv2d(cs, cs // 2, 3, padding=1), nn.ReLU(inplace=True))
        self.dec0 = make_decoder_block(cs // 2, cs // 2, cs // 4, noupsam=noupdownsam, upsc=0.5)
        self.final = nn.Conv2d(4 * cs + cs + cs // 2 + cs // 2 + cs // 4, out_classes, kernel_size=1)

    def _forward_host(self, x):
        decs = [getattr(self, f"dec{k}")(getattr(self, f"conv{k}")(x)) for k in (4, 3, 2, 1, 0)]
        out = self.final(torch.cat(decs, 1))
        return (out, decs[::-1]) if self.multiout else out

    def _emit(self, rt, x):
        N, _, H, W = x.shape
        cs = self.channelscale
        if self.noupdownsam:
            raise NotImplementedError("SpiderNet(noupdownsam=True) mixes resolutions in its final concatenation and fails in the reference too")
        if cs % 32:
            raise NotImplementedError("channelscale must be a multiple of 32 (8-channel storage groups)")
        if H % 8 or W % 8:
            raise RuntimeError(f"SpiderNet needs H and W divisible by 8, got {H}x{W}")
        widths = {4: 4 * cs, 3: cs, 2: cs // 2, 1: cs // 2, 0: cs // 4}
        cat = rt.new(N, H, W, sum(widths.values())).buf             # [dec4 | dec3 | dec2 | dec1 | dec0] (reference :993)
        decs, c0 = {}, 0
        for k in (4, 3, 2, 1, 0):
            out = Act(cat, c0, widths[k])
            c0 += widths[k]
            if k == 1:
                h = _emit_stem(rt, x, None)
                decs[k] = rt.conv(h, "dec1.0.weight", "dec1.0.bias", relu=True, out=out)
            else:
                h = getattr(self, f"conv{k}").emit(rt, f"conv{k}.", x)
                decs[k] = getattr(self, f"dec{k}").emit(rt, f"dec{k}.", h, out=out)
        rt.unused(["conv5.0.weight", "conv5.0.bias", "conv5.2.weight", "conv5.2.bias", "conv5.4.weight", "conv5.4.bias"])
        logits = rt.head_wide([decs[k] for k in (4, 3, 2, 1, 0)], "final.weight", "final.bias")
        return (logits, [rt.export(decs[k]) for k in (0, 1, 2, 3, 4)]) if self.multiout else logits


def _bottom_modules(cs):
    # the two InstanceNorm2d widths are the reference's literals; without affine parameters the width is unused
    return nn.Sequential(
        nn.Conv2d(8 * cs, 8 * cs, kernel_size=3, padding=1), nn.InstanceNorm2d(512),
        nn.Conv2d(8 * cs, 8 * cs, kernel_size=(3, 3), padding=(1, 1)), nn.ReLU(inplace=True),
        nn.Conv2d(8 * cs, 8 * cs, kernel_size=3, padding=1), nn.InstanceNorm2d(4 * cs), nn.ReLU(inplace=True))


class UNetTetra(SegkNet):
    """UNetSimple with a second bottom stage (conv15) and a second head (final1) for edges (reference :996-1095).  Its forward
    decodes the second branch with the FIRST branch's dec4..dec1 (:1090-1093), so those weights receive the sum of both
    branches' gradients and dec14..dec11 are never used; both facts are reproduced.  Returns (final(dec1), final1(dec11))."""

    def __init__(self, pretrained=False, out_classes=3, patch_train=False, in_classes=1, channelscale=64, multiout=False,
                 noupdownsam=False):
        super().__init__()
        cs = channelscale
        self.multiout, self.patch_train, self.channelscale, self.noupdownsam = multiout, patch_train, cs, noupdownsam
        self.conv1 = _stem_modules(in_classes, cs)
        self.conv2 = make_encoder_block(cs, 2 * cs, 2 * cs, convs3=False, no_dsam=noupdownsam)
        self.conv3 = make_encoder_block(2 * cs, 4 * cs, 4 * cs, middle_channels2=4 * cs, no_dsam=noupdownsam)
        self.conv4 = make_encoder_block(4 * cs, 8 * cs, 8 * cs, middle_channels2=8 * cs, no_dsam=noupdownsam)
        self.conv5 = _bottom_modules(cs)
        for tag, head in (("", "final"), ("1", "final1")):
            if tag:
                self.conv15 = _bottom_modules(cs)
            setattr(self, f"dec{tag}4", make_decoder_block(16 * cs, 8 * cs, 4 * cs, noupsam=noupdownsam))
            setattr(self, f"dec{tag}3", make_decoder_block(8 * cs, 4 * cs, cs, noupsam=noupdownsam))
            setattr(self, f"dec{tag}2", make_decoder_block(3 * cs, 2 * cs, cs // 2, noupsam=noupdownsam))
            setattr(self, f"dec{tag}1", nn.Sequential(nn.Conv2d(int(1.5 * cs), cs // 2, 3, padding=1), nn.ReLU(inplace=True)))
            setattr(self, head, nn.Conv2d(cs // 2, out_classes, kernel_size=1))

    def _forward_host(self, x):
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        outs = []
        for bottom, head in ((self.conv5, self.final), (self.conv15, self.final1)):
            d = self.dec4(torch.cat([bottom(c4), c4], 1))
            d = self.dec3(torch.cat([d, c3], 1))
            d = self.dec2(torch.cat([d, c2], 1))
            outs.append(head(self.dec1(torch.cat([d, c1], 1))))
        return outs[0], outs[1]

    def _pack_outputs(self, logits, feats):
        return logits, feats[0]

    def _emit(self, rt, x):
        N, _, H, W = x.shape
        cs, down = self.channelscale, not self.noupdownsam
        if cs % 16:
            raise NotImplementedError("channelscale must be a multiple of 16 (8-channel storage groups)")
        if down and (H % 8 or W % 8):
            raise RuntimeError(f"UNetTetra needs H and W divisible by 8 (three stride-2 convolutions), got {H}x{W}")
        dec_w, enc_w = {1: cs // 2, 2: cs, 3: 4 * cs, 4: 8 * cs}, {1: cs, 2: 2 * cs, 3: 4 * cs, 4: 8 * cs}

        def buffers():
            return {k: rt.new(N, H >> (k - 1) if down else H, W >> (k - 1) if down else W, dec_w[k] + enc_w[k]).buf for k in range(1, 5)}
        cat, cat2 = buffers(), buffers()          # [decoder tensor | encoder skip] per level, one set per branch
        skips = {1: _emit_stem(rt, x, Act(cat[1], dec_w[1], enc_w[1]))}
        for k in range(2, 5):
            skips[k] = getattr(self, f"conv{k}").emit(rt, f"conv{k}.", skips[k - 1], out=Act(cat[k], dec_w[k], enc_w[k]))
        outs = []
        for bufs, bottom, head in ((cat, "conv5", "final"), (cat2, "conv15", "final1")):
            sk = skips if bufs is cat else {k: rt.copy(skips[k], Act(bufs[k], dec_w[k], enc_w[k])) for k in range(1, 5)}
            d = _emit_bottom(rt, skips[4], Act(bufs[4], 0, dec_w[4]), prefix=bottom)
            for k in (4, 3, 2):
                d = getattr(self, f"dec{k}").emit(rt, f"dec{k}.", [d, sk[k]], out=Act(bufs[k - 1], 0, dec_w[k - 1]))
            e = rt.conv([d, sk[1]], "dec1.0.weight", "dec1.0.bias", relu=True)
            outs.append(rt.head(e, f"{head}.weight", f"{head}.bias"))
        rt.unused([n for n, _ in self.named_parameters() if n.startswith(("dec14.", "dec13.", "dec12.", "dec11."))])
        return outs[0], [outs[1]]


# ---------------------------------------------------------------------------------------------------------------------
# Wrappers that re-size the image before and the logits after a VGG U-Net.  The re-sizing acts on NCHW images and logits
# at the network boundary (nearest-neighbour interpolation, adaptive max pooling, cropping, padding: torch ops with
# autograd); the network in between runs on the HIP tape.
# ---------------------------------------------------------------------------------------------------------------------
class UNetVgg256(nn.Module):
    """Input resized to 128x128, three-class UNetVgg, logits max-pooled back to the input size (reference :107-132; the
    `out_classes` argument is accepted and unused, as there)."""

    def __init__(self, out_classes=3):
        super().__init__()
        self.unetvgg = UNetVgg()

    def forward(self, x):
        size = x.shape[-2:]
        out = self.unetvgg(F.interpolate(x, [128, 128]))
        return F.adaptive_max_pool2d(out, size)


class UNetVggWholeImg(nn.Module):
    """Three-class UNetVgg initialised from a checkpoint's 'state_dict' entry (reference :134-155)."""

    def __init__(self, save_model_path, out_classes=3):
        super().__init__()
        self.unetvgg = UNetVgg()
        checkpoint = torch.load(save_model_path, map_location='cpu', weights_only=True)
        self.unetvgg.load_state_dict(checkpoint['state_dict'])

    def forward(self, x):
        return self.unetvgg(x)


class UNetVggTwo(nn.Module):
    """One UNetVggInOut per image half (reference :216-265): optional crop to the labelled region (+15 pixels), each half
    resized to 256x256, logits max-pooled back to the half's size, halves joined, and the crop padded back with the
    reference's own offsets (:264)."""

    def __init__(self):
        super().__init__()
        self.unetvggl = UNetVggInOut()
        self.unetvggr = UNetVggInOut()

    def forward(self, x, y=None):
        if y is not None:
            rows, cols = torch.nonzero(y, as_tuple=True)[1], torch.nonzero(y, as_tuple=True)[2]
            margin, full_h, full_w = 15, y.shape[-2], y.shape[-1]
            top, bottom = max(0, rows.min() - margin), min(rows.max() + margin, full_h)
            left, right = max(0, cols.min() - margin), min(cols.max() + margin, full_w)
            x = x[:, :, top:bottom, left:right].clone()
        height, width = x.shape[-2], x.shape[-1]
        half = int(width / 2)
        outl = self.unetvggl(F.interpolate(x[:, :, :, :half].clone(), [256, 256]))
        outr = self.unetvggr(F.interpolate(x[:, :, :, half:].clone(), [256, 256]))
        out = torch.cat((F.adaptive_max_pool2d(outl, [height, half]), F.adaptive_max_pool2d(outr, [height, width - half])), dim=-1)
        if y is not None:
            out = F.pad(out, (left - 1, full_w - right + 1, top - 1, full_h - bottom + 1))
        return out
