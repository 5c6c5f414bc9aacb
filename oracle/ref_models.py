"""CPU oracle for the segmentation networks.  TEST INFRASTRUCTURE ONLY.

Plain `torch.nn` restatement of the two graded networks with the reference's state_dict layout:

* VGG16-BN encoder + five-stage decoder -- /root/reference/models/unetvgg.py:6-41 (decoder block),
  :52-105 (UNetVgg), :107-132 (UNetVgg256), :158-213 (UNetVggInOut), :216-265 (UNetVggTwo)
* plain InstanceNorm U-Nets             -- /root/reference/models/unetvgg.py:268-323 (encoder block), :335-372 (center),
                                           :373-440 (UNetNaive), :443-506 (UNetWithoutCenter), :509-574 (UNetNaiveMultiOut),
                                           :576-648 (UNetLimMultiOut), :651-728 (UNetSimple), :730-813 (UNetSkip),
                                           :815-894 (UNetSimpleReduced), :896-993 (SpiderNet), :996-1095 (UNetTetra)
* nested U-Net (UNet++)                 -- /root/reference/models/unetplusplus.py:6-25, 27-82
* 3-D DenseNet-BC classifier + DetectMid -- /root/reference/models/densenet3d.py:10-159, 162-186;
                                           /root/reference/models/resnt.py:8-45

Pinned by `oracle/make_golden.py` (bitwise equality with the imported reference in the build
container).  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

# (first VGG feature index, conv widths) of the five encoder stages; a stage after the first opens
# with the max-pool that sits at `first` (unetvgg.py:72-76 slice the torchvision feature list).
_STAGES = [(0, [64, 64]), (6, [128, 128]), (13, [256, 256, 256]), (23, [512, 512, 512]), (33, [512, 512, 512])]


def _encoder_stage(first, c_in, widths):
    mods, idx = OrderedDict(), first
    if first:
        mods[str(idx)] = nn.MaxPool2d(2, 2)
        idx += 1
    for wdt in widths:
        mods[str(idx)] = nn.Conv2d(c_in, wdt, 3, padding=1)
        mods[str(idx + 1)] = nn.BatchNorm2d(wdt)
        mods[str(idx + 2)] = nn.ReLU(inplace=True)
        c_in, idx = wdt, idx + 3
    return nn.Sequential(mods)


def decoder_block(c_in, c_mid, c_out, if_convtrans=False, noupsam=False, upsc=2):
    head = [nn.Conv2d(c_in, c_mid, 3, padding=1), nn.ReLU(inplace=True)]
    if noupsam:
        tail = [nn.Conv2d(c_mid, c_out, 3, padding=1, bias=False), nn.InstanceNorm2d(c_out)]
    elif if_convtrans:
        tail = [nn.ConvTranspose2d(c_mid, c_out, kernel_size=4, stride=2, padding=1)]
    else:
        tail = [nn.Conv2d(c_mid, c_out, 3, padding=1, bias=False), nn.InstanceNorm2d(c_out),
                nn.Upsample(scale_factor=upsc, mode='bilinear', align_corners=False)]
    return nn.Sequential(*head, *tail, nn.ReLU(inplace=True))


class RefUNetVgg(nn.Module):
    def __init__(self, out_classes=3, if_convtrans=False, in_classes=None, patch_train=False):
        super().__init__()
        self.patch_train = patch_train
        if in_classes is not None:          # UNetVggInOut: extra conv in front, bilinear decoder only
            self.conv0 = nn.Conv2d(in_classes, 3, 3, padding=1)
        c = 3
        for k, (first, widths) in enumerate(_STAGES):
            setattr(self, f"conv{k + 1}", _encoder_stage(first, c, widths))
            c = widths[-1]
        ct = if_convtrans
        self.center = nn.Sequential(nn.MaxPool2d(2, 2), decoder_block(512, 512, 256, ct))
        self.dec5 = decoder_block(768, 512, 256, ct)
        self.dec4 = decoder_block(768, 512, 256, ct)
        self.dec3 = decoder_block(512, 256, 64, ct)
        self.dec2 = decoder_block(192, 128, 32, ct)
        self.dec1 = nn.Sequential(nn.Conv2d(96, 32, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(32, out_classes, 1)

    def forward(self, x):
        if hasattr(self, "conv0"):
            x = self.conv0(x)
        skips = []
        for k in range(1, 6):
            x = getattr(self, f"conv{k}")(x)
            skips.append(x)
        x = self.center(x)
        for k in range(5, 0, -1):
            x = getattr(self, f"dec{k}")(torch.cat([x, skips[k - 1]], 1))
        return self.final(x)


def encoder_block(c_in, c_mid, c_out, convs3=True, c_mid2=None, no_dsam=False):
    """unetvgg.py:268-323 with max_pool=False, if_interpolate=False (the settings UNetSimple uses)."""
    down = nn.Conv2d(c_in, c_in, 3, padding=1) if no_dsam else nn.Conv2d(c_in, c_in, kernel_size=2, stride=2)
    mods, c = [down], c_in
    for wdt in [c_mid] + ([c_mid2] if convs3 else []) + [c_out]:
        mods += [nn.Conv2d(c, wdt, 3, padding=1, bias=False), nn.InstanceNorm2d(wdt), nn.ReLU(inplace=True)]
        c = wdt
    return nn.Sequential(*mods)


class RefUNetSimple(nn.Module):
    def __init__(self, out_classes=3, in_classes=1, channelscale=64, noupdownsam=False, multiout=False):
        super().__init__()
        cs, nd, self.multiout = channelscale, noupdownsam, multiout
        self.conv1 = nn.Sequential(nn.Conv2d(in_classes, cs, 3, padding=1, bias=False), nn.InstanceNorm2d(cs), nn.ReLU(inplace=True),
                                   nn.Conv2d(cs, cs, 3, padding=1, bias=False), nn.InstanceNorm2d(cs), nn.ReLU(inplace=True))
        self.conv2 = encoder_block(cs, 2 * cs, 2 * cs, convs3=False, no_dsam=nd)
        self.conv3 = encoder_block(2 * cs, 4 * cs, 4 * cs, c_mid2=4 * cs, no_dsam=nd)
        self.conv4 = encoder_block(4 * cs, 8 * cs, 8 * cs, c_mid2=8 * cs, no_dsam=nd)
        self.conv5 = nn.Sequential(nn.Conv2d(8 * cs, 8 * cs, 3, padding=1), nn.InstanceNorm2d(8 * cs),
                                   nn.Conv2d(8 * cs, 8 * cs, 3, padding=1), nn.ReLU(inplace=True),
                                   nn.Conv2d(8 * cs, 8 * cs, 3, padding=1), nn.InstanceNorm2d(8 * cs), nn.ReLU(inplace=True))
        self.dec4 = decoder_block(16 * cs, 8 * cs, 4 * cs, noupsam=nd)
        self.dec3 = decoder_block(8 * cs, 4 * cs, cs, noupsam=nd)
        self.dec2 = decoder_block(3 * cs, 2 * cs, cs // 2, noupsam=nd)
        self.dec1 = nn.Sequential(nn.Conv2d(cs + cs // 2, cs // 2, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(cs // 2, out_classes, 1)

    def forward(self, x):
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        c5 = self.conv5(c4)
        d4 = self.dec4(torch.cat([c5, c4], 1))
        d3 = self.dec3(torch.cat([d4, c3], 1))
        d2 = self.dec2(torch.cat([d3, c2], 1))
        d1 = self.dec1(torch.cat([d2, c1], 1))
        if self.multiout == "dec4":             # UNetSkip (unetvgg.py:730-813): UNetSimple that returns (logits, dec4)
            return self.final(d1), d4
        return (self.final(d1), [d1, d2, d3, d4]) if self.multiout else self.final(d1)


def _in_stem(in_classes, wdt):
    return nn.Sequential(nn.Conv2d(in_classes, wdt, 3, padding=1, bias=False), nn.InstanceNorm2d(wdt), nn.ReLU(inplace=True),
                         nn.Conv2d(wdt, wdt, 3, padding=1, bias=False), nn.InstanceNorm2d(wdt), nn.ReLU(inplace=True))


def center_block(max_pool, iftrans2d):
    """unetvgg.py:335-372."""
    down = nn.MaxPool2d(2, 2) if max_pool else nn.Conv2d(512, 512, kernel_size=2, stride=2)
    if iftrans2d:
        tail = [nn.ConvTranspose2d(512, 256, kernel_size=4, stride=2, padding=1)]
    else:
        tail = [nn.Conv2d(512, 256, 3, padding=1), nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)]
    return nn.Sequential(down, nn.Conv2d(512, 512, 3, padding=1), nn.ReLU(inplace=True), *tail, nn.ReLU(inplace=True))


class RefUNetNaive(nn.Module):
    def __init__(self, out_classes=3, in_classes=1, if_convtrans=False, ifmaxpool=False, center=True):
        """center=False: UNetWithoutCenter (unetvgg.py:443-506), whose conv5 is a plain stack at 1/8 resolution."""
        super().__init__()
        self.conv1 = _in_stem(in_classes, 64)
        self.conv2 = encoder_block(64, 128, 128, convs3=False)
        self.conv3 = encoder_block(128, 256, 256, c_mid2=256)
        self.conv4 = encoder_block(256, 512, 512, c_mid2=512)
        if center:
            self.conv5 = encoder_block(512, 512, 512, c_mid2=512)
            self.center = center_block(ifmaxpool, if_convtrans)
            self.dec5 = decoder_block(768, 512, 256)
        else:
            self.conv5 = nn.Sequential(nn.Conv2d(512, 512, 3, padding=1), nn.InstanceNorm2d(512),
                                       nn.Conv2d(512, 512, 3, padding=1), nn.ReLU(inplace=True),
                                       nn.Conv2d(512, 256, 3, padding=1), nn.InstanceNorm2d(256), nn.ReLU(inplace=True))
        self.dec4 = decoder_block(768, 512, 256)
        self.dec3 = decoder_block(512, 256, 64)
        self.dec2 = decoder_block(192, 128, 32)
        self.dec1 = nn.Sequential(nn.Conv2d(96, 32, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(32, out_classes, 1)

    def forward(self, x):
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        c5 = self.conv5(c4)
        if hasattr(self, "center"):
            d = self.dec5(torch.cat([self.center(c5), c5], 1))
        else:
            d = c5
        d = self.dec4(torch.cat([d, c4], 1))
        d = self.dec3(torch.cat([d, c3], 1))
        d = self.dec2(torch.cat([d, c2], 1))
        return self.final(self.dec1(torch.cat([d, c1], 1)))


class RefUNetFourLevel(nn.Module):
    """UNetNaiveMultiOut (unetvgg.py:509-574; widths 64..512, returns [conv3, dec4]) and UNetLimMultiOut (:576-648; every
    layer `channel_depth` wide, returns conv3 when multiout): bias-free conv-IN-ReLU bottom stage."""

    def __init__(self, out_classes=3, in_classes=1, channel_depth=None, multiout=True):
        super().__init__()
        lim = channel_depth is not None
        self.lim, self.multiout = lim, multiout
        e = [channel_depth] * 4 if lim else [64, 128, 256, 512]
        bottom = channel_depth // 2 if lim else 256
        dec_out = [channel_depth // 2] * 3 if lim else [256, 64, 32]          # dec4, dec3, dec2
        dec_mid = [channel_depth] * 3 if lim else [512, 256, 128]
        self.conv1 = _in_stem(in_classes, e[0])
        self.conv2 = encoder_block(e[0], e[1], e[1], convs3=False)
        self.conv3 = encoder_block(e[1], e[2], e[2], c_mid2=e[2])
        self.conv4 = encoder_block(e[2], e[3], e[3], c_mid2=e[3])
        self.conv5 = nn.Sequential(nn.Conv2d(e[3], e[3], 3, padding=1, bias=False), nn.InstanceNorm2d(e[3]), nn.ReLU(inplace=True),
                                   nn.Conv2d(e[3], bottom, 3, padding=1, bias=False), nn.InstanceNorm2d(bottom), nn.ReLU(inplace=True))
        self.dec4 = decoder_block(bottom + e[3], dec_mid[0], dec_out[0])
        self.dec3 = decoder_block(dec_out[0] + e[2], dec_mid[1], dec_out[1])
        self.dec2 = decoder_block(dec_out[1] + e[1], dec_mid[2], dec_out[2])
        last = channel_depth if lim else 32
        self.dec1 = nn.Sequential(nn.Conv2d(dec_out[2] + e[0], last, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(last, out_classes, 1)

    def forward(self, x):
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        d4 = self.dec4(torch.cat([self.conv5(c4), c4], 1))
        d = self.dec3(torch.cat([d4, c3], 1))
        d = self.dec2(torch.cat([d, c2], 1))
        y = self.final(self.dec1(torch.cat([d, c1], 1)))
        if not self.multiout:
            return y
        return (y, c3) if self.lim else (y, [c3, d4])


class RefUNetSimpleReduced(nn.Module):
    """unetvgg.py:815-894 (multiout=False)."""

    def __init__(self, out_classes=3, in_classes=1, channelscale=64, noupdownsam=False):
        super().__init__()
        cs, nd = channelscale, noupdownsam
        self.conv1 = _in_stem(in_classes, cs)
        self.conv2 = encoder_block(cs, 2 * cs, 2 * cs, convs3=False, no_dsam=nd)
        self.conv3 = encoder_block(2 * cs, 4 * cs, 8 * cs, c_mid2=4 * cs, no_dsam=nd)
        self.dec4 = decoder_block(8 * cs, 8 * cs, 4 * cs, noupsam=True)
        self.dec3 = decoder_block(12 * cs, 4 * cs, cs, noupsam=nd)
        self.dec2 = decoder_block(3 * cs, 2 * cs, cs // 2, noupsam=nd)
        self.dec1 = nn.Sequential(nn.Conv2d(cs + cs // 2, cs // 2, 3, padding=1), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(cs // 2, out_classes, 1)

    def forward(self, x):
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        d = self.dec3(torch.cat([self.dec4(c3), c3], 1))
        d = self.dec2(torch.cat([d, c2], 1))
        return self.final(self.dec1(torch.cat([d, c1], 1)))


def _interp_encoder(in_classes, c_mid, c_out, scale, convs3=True, c_mid2=None):
    """unetvgg.py:268-323 with if_interpolate=True: the block opens with a bilinear re-scaling of its input."""
    mods, c = [nn.Upsample(scale_factor=scale, mode="bilinear", align_corners=False)], in_classes
    for wdt in [c_mid] + ([c_mid2] if convs3 else []) + [c_out]:
        mods += [nn.Conv2d(c, wdt, 3, padding=1, bias=False), nn.InstanceNorm2d(wdt), nn.ReLU(inplace=True)]
        c = wdt
    return nn.Sequential(*mods)


class RefSpiderNet(nn.Module):
    """unetvgg.py:896-993; conv5 exists in the state_dict and is never called."""

    def __init__(self, out_classes=3, in_classes=1, channelscale=64, multiout=False):
        super().__init__()
        cs, self.multiout = channelscale, multiout
        self.conv0 = _interp_encoder(in_classes, cs // 2, cs // 2, 2, convs3=False)
        self.conv1 = _in_stem(in_classes, cs)
        self.conv2 = _interp_encoder(in_classes, 2 * cs, 2 * cs, 0.5, convs3=False)
        self.conv3 = _interp_encoder(in_classes, 4 * cs, 4 * cs, 0.25, c_mid2=4 * cs)
        self.conv4 = _interp_encoder(in_classes, 8 * cs, 8 * cs, 1 / 8, c_mid2=8 * cs)
        self.conv5 = nn.Sequential(nn.Conv2d(15 * cs, 8 * cs, 3, padding=1), nn.InstanceNorm2d(8 * cs),
                                   nn.Conv2d(12 * cs, 8 * cs, 3, padding=1), nn.ReLU(inplace=True),
                                   nn.Conv2d(8 * cs, 8 * cs, 3, padding=1), nn.InstanceNorm2d(8 * cs), nn.ReLU(inplace=True))
        self.dec4 = decoder_block(8 * cs, 8 * cs, 4 * cs, upsc=8)
        self.dec3 = decoder_block(4 * cs, 4 * cs, cs, upsc=4)
        self.dec2 = decoder_block(2 * cs, 2 * cs, cs // 2, upsc=2)
        self.dec1 = nn.Sequential(nn.Conv2d(cs, cs // 2, 3, padding=1), nn.ReLU(inplace=True))
        self.dec0 = decoder_block(cs // 2, cs // 2, cs // 4, upsc=0.5)
        self.final = nn.Conv2d(4 * cs + cs + cs // 2 + cs // 2 + cs // 4, out_classes, 1)

    def forward(self, x):
        d = [getattr(self, f"dec{k}")(getattr(self, f"conv{k}")(x)) for k in range(5)]
        y = self.final(torch.cat(d[::-1], 1))
        return (y, d) if self.multiout else y


class RefUNetTetra(RefUNetSimple):
    """unetvgg.py:996-1095: a second bottom stage and head; the second branch is decoded with the FIRST branch's decoder
    blocks (:1090-1093), the second set (dec14..dec11) exists in the state_dict and is never called."""

    def __init__(self, out_classes=3, in_classes=1, channelscale=64):
        super().__init__(out_classes=out_classes, in_classes=in_classes, channelscale=channelscale)
        cs = channelscale
        self.conv15 = nn.Sequential(nn.Conv2d(8 * cs, 8 * cs, 3, padding=1), nn.InstanceNorm2d(8 * cs),
                                    nn.Conv2d(8 * cs, 8 * cs, 3, padding=1), nn.ReLU(inplace=True),
                                    nn.Conv2d(8 * cs, 8 * cs, 3, padding=1), nn.InstanceNorm2d(8 * cs), nn.ReLU(inplace=True))
        self.dec14 = decoder_block(16 * cs, 8 * cs, 4 * cs)
        self.dec13 = decoder_block(8 * cs, 4 * cs, cs)
        self.dec12 = decoder_block(3 * cs, 2 * cs, cs // 2)
        self.dec11 = nn.Sequential(nn.Conv2d(cs + cs // 2, cs // 2, 3, padding=1), nn.ReLU(inplace=True))
        self.final1 = nn.Conv2d(cs // 2, out_classes, 1)

    def forward(self, x):
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


class RefUNetVgg256(nn.Module):
    """unetvgg.py:107-132."""

    def __init__(self):
        super().__init__()
        self.unetvgg = RefUNetVgg()

    def forward(self, x):
        out = self.unetvgg(F.interpolate(x, [128, 128]))
        return nn.AdaptiveMaxPool2d(x.shape[-2:])(out)


class RefUNetVggTwo(nn.Module):
    """unetvgg.py:216-265."""

    def __init__(self):
        super().__init__()
        self.unetvggl = RefUNetVgg(in_classes=1)
        self.unetvggr = RefUNetVgg(in_classes=1)

    def forward(self, x, y=None):
        if y is not None:
            nz = torch.nonzero(y, as_tuple=True)
            h0, h1 = max(0, nz[1].min() - 15), min(nz[1].max() + 15, y.shape[-2])
            w0, w1 = max(0, nz[2].min() - 15), min(nz[2].max() + 15, y.shape[-1])
            x = x[:, :, h0:h1, w0:w1].clone()
        hgt, wid = x.shape[-2], x.shape[-1]
        mid = int(wid / 2)
        ol = self.unetvggl(F.interpolate(x[..., :mid].clone(), [256, 256]))
        orr = self.unetvggr(F.interpolate(x[..., mid:].clone(), [256, 256]))
        out = torch.cat((nn.AdaptiveMaxPool2d([hgt, mid])(ol), nn.AdaptiveMaxPool2d([hgt, wid - mid])(orr)), dim=-1)
        if y is not None:
            out = F.pad(out, (w0 - 1, y.shape[-1] - w1 + 1, h0 - 1, y.shape[-2] - h1 + 1))
        return out


class RefDoubleConv(nn.Module):
    def __init__(self, c_in, c_mid, c_out):
        super().__init__()
        self.activation = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(c_in, c_mid, 3, padding=1, bias=True)
        self.bn1 = nn.BatchNorm2d(c_mid)
        self.conv2 = nn.Conv2d(c_mid, c_out, 3, padding=1, bias=True)
        self.bn2 = nn.BatchNorm2d(c_out)

    def forward(self, x):
        x = self.activation(self.bn1(self.conv1(x)))
        return self.activation(self.bn2(self.conv2(x)))


class RefNestedUNet(nn.Module):
    """deep_supervision=True (heads final1..final4 on x(0,1)..x(0,4), list of logits) restates the UNet++ paper's deep
    supervision in plain torch; the reference file has a single head, so that mode is NOT pinned by the reference."""

    def __init__(self, in_ch=1, out_ch=3, deep_supervision=False):
        super().__init__()
        self.deep_supervision = deep_supervision
        f = [64, 128, 256, 512, 1024]
        self.pool = nn.MaxPool2d(2, 2)
        self.Up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        for i in range(5):                                   # backbone column j = 0
            setattr(self, f"conv{i}_0", RefDoubleConv(in_ch if i == 0 else f[i - 1], f[i], f[i]))
        for j in range(1, 5):                                # nested columns
            for i in range(5 - j):
                setattr(self, f"conv{i}_{j}", RefDoubleConv(f[i] * j + f[i + 1], f[i], f[i]))
        if deep_supervision:
            for j in range(1, 5):
                setattr(self, f"final{j}", nn.Conv2d(f[0], out_ch, 1))
        else:
            self.final = nn.Conv2d(f[0], out_ch, 1)

    def forward(self, x):
        node = {}
        for i in range(5):
            node[i, 0] = getattr(self, f"conv{i}_0")(x if i == 0 else self.pool(node[i - 1, 0]))
            # every anti-diagonal i+j = const is completed as soon as node (i,0) exists
            for j in range(1, i + 1):
                r = i - j
                parts = [node[r, k] for k in range(j)] + [self.Up(node[r + 1, j - 1])]
                node[r, j] = getattr(self, f"conv{r}_{j}")(torch.cat(parts, 1))
        if self.deep_supervision:
            return [getattr(self, f"final{j}")(node[0, j]) for j in range(1, 5)]
        return self.final(node[0, 4])


# ---------------------------------------------------------------------------------------------
# Deterministic, torch-RNG-independent initialiser shared by the oracle, the tests and the fixtures
# ---------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------------
# 3-D DenseNet-BC (densenet3d.py) and the middle-slice detector on top of it (resnt.py:8-45)
# ---------------------------------------------------------------------------------------------------
def _bn_relu_conv3d(c_in, c_out, k, names):
    """BatchNorm3d -> ReLU -> Conv3d(bias=False, 'same' padding) under the given three child names."""
    return [(names[0], nn.BatchNorm3d(c_in)), (names[1], nn.ReLU(inplace=True)),
            (names[2], nn.Conv3d(c_in, c_out, kernel_size=k, stride=1, padding=k // 2, bias=False))]


class _RefDenseLayer(nn.Sequential):
    """densenet3d.py:10-41: bottleneck 1x1x1 to bn_size*growth channels, 3x3x3 to growth channels, concat with the input."""

    def __init__(self, c_in, growth, bn_size):
        super().__init__(OrderedDict(_bn_relu_conv3d(c_in, bn_size * growth, 1, ("norm1", "relu1", "conv1")) +
                                     _bn_relu_conv3d(bn_size * growth, growth, 3, ("norm2", "relu2", "conv2"))))

    def forward(self, x):
        return torch.cat([x, super().forward(x)], 1)


class RefDenseNet3D(nn.Module):
    """densenet3d.py:71-159 with drop_rate=0 (the reference default; DetectMid never changes it)."""

    def __init__(self, n_input_channels=3, conv1_t_size=7, conv1_t_stride=1, no_max_pool=False, growth_rate=32,
                 block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4, num_classes=1000):
        super().__init__()
        stem = OrderedDict()
        stem["conv1"] = nn.Conv3d(n_input_channels, num_init_features, kernel_size=(conv1_t_size, 7, 7),
                                  stride=(conv1_t_stride, 2, 2), padding=(conv1_t_size // 2, 3, 3), bias=False)
        stem["norm1"] = nn.BatchNorm3d(num_init_features)
        stem["relu1"] = nn.ReLU(inplace=True)
        if not no_max_pool:
            stem["pool1"] = nn.MaxPool3d(kernel_size=3, stride=2, padding=1)
        self.features = nn.Sequential(stem)
        c = num_init_features
        for b, n_layers in enumerate(block_config, start=1):
            block = nn.Sequential(OrderedDict((f"denselayer{j + 1}", _RefDenseLayer(c + j * growth_rate, growth_rate, bn_size))
                                              for j in range(n_layers)))
            self.features.add_module(f"denseblock{b}", block)
            c += n_layers * growth_rate
            if b != len(block_config):
                trans = nn.Sequential(OrderedDict(_bn_relu_conv3d(c, c // 2, 1, ("norm", "relu", "conv")) +
                                                  [("pool", nn.AvgPool3d(kernel_size=2, stride=2))]))
                self.features.add_module(f"transition{b}", trans)
                c //= 2
        self.features.add_module("norm5", nn.BatchNorm3d(c))
        self.classifier = nn.Linear(c, num_classes)

    def forward(self, x):
        f = self.features(x)
        out = torch.nn.functional.relu(f, inplace=True)
        out = torch.nn.functional.adaptive_avg_pool3d(out, output_size=(1, 1, 1)).view(f.size(0), -1)
        return self.classifier(out)


_DENSE_CONFIGS = {121: (6, 12, 24, 16), 169: (6, 12, 32, 32), 201: (6, 12, 48, 32), 264: (6, 12, 64, 48)}


class RefDetectMid(nn.Module):
    """resnt.py:8-45: Conv3d(in_classes, 3, (3,7,7), stride (1,2,2)) -> DenseNet-169 (3-D) -> Linear(1000, 80)
    (-> Linear(80, out_classes), squeezed, when if_regr)."""

    def __init__(self, in_classes=1, out_classes=1, if_regr=False, model_depth=169):
        super().__init__()
        self.rsnet = RefDenseNet3D(block_config=_DENSE_CONFIGS[model_depth])
        self.patch_train = False
        self.first = nn.Conv3d(in_classes, 3, kernel_size=(3, 7, 7), stride=(1, 2, 2), padding=(1, 3, 3), bias=False)
        self.last = nn.Linear(1000, 80)
        self.lastregr = nn.Linear(80, out_classes)
        self.if_regr = if_regr

    def forward(self, x):
        out = self.last(self.rsnet(self.first(x)))
        if self.if_regr:
            out = self.lastregr(out).squeeze(dim=-1)
        return out


def synthetic_volumes(n, depth, side, tag="vol"):
    """Z-scored single-channel volumes and a middle-slice class per volume (< 80, train_midlslice.py:25-33)."""
    x = hash_uniform(tag + ":vol", (n, 1, depth, side, side), -1.7320508, 1.7320508)
    t = (hash_uniform(tag + ":mid", (n,), 0.0, 1.0) * min(depth, 80)).long().clamp_(0, 79)
    return x, t


def _splitmix64(x):
    import numpy as np
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_uniform(tag, shape, lo=-1.0, hi=1.0):
    """Uniform [lo,hi) values that depend only on `tag` and the flat element index."""
    import zlib
    import numpy as np
    n = 1
    for s in shape:
        n *= int(s)
    seed = np.uint64(zlib.crc32(tag.encode()) * 2654435761 % (1 << 63))
    with np.errstate(over='ignore'):
        bits = _splitmix64(np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + seed)
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def hash_init_(module, salt="w"):
    """Fill a module's parameters and BN buffers by key name: conv weights ~ U(+-sqrt(3/fan_in))
    (unit-variance preserving), biases small, BN gamma in [0.75,1.25], beta small."""
    sd = module.state_dict()
    for key, t in sd.items():
        tag = f"{salt}:{key}"
        if key.endswith("num_batches_tracked"):
            t.zero_()
        elif key.endswith("running_mean"):
            t.copy_(hash_uniform(tag, t.shape, -0.1, 0.1))
        elif key.endswith("running_var"):
            t.copy_(hash_uniform(tag, t.shape, 0.8, 1.2))
        elif t.dim() in (4, 5):
            fan_in = t[0].numel()
            bound = (6.0 / fan_in) ** 0.5      # He-uniform keeps activations O(1) through ReLU stacks
            t.copy_(hash_uniform(tag, t.shape, -bound, bound))
        elif t.dim() == 2:                     # nn.Linear
            bound = (3.0 / t.shape[1]) ** 0.5
            t.copy_(hash_uniform(tag, t.shape, -bound, bound))
        elif key.endswith("weight"):           # BN gamma
            t.copy_(hash_uniform(tag, t.shape, 0.75, 1.25))
        else:                                  # biases / BN beta
            t.copy_(hash_uniform(tag, t.shape, -0.1, 0.1))
    module.load_state_dict(sd)
    return module


def synthetic_batch(n, side, n_classes, in_ch=3, tag="batch"):
    """Grey z-scored slice replicated to `in_ch` channels (read_roi.py:183-187 convention) and a label
    map with small filled ellipses; every 4th image is all background (exercises the fg filter)."""
    import numpy as np
    grey = hash_uniform(tag + ":img", (n, 1, side, side), -1.7320508, 1.7320508)
    img = grey.expand(n, in_ch, side, side).contiguous()
    yy, xx = np.mgrid[0:side, 0:side].astype(np.float32)
    lab = np.zeros((n, side, side), dtype=np.int64)
    for b in range(n):
        if b % 4 == 3:
            continue
        for c in range(1, n_classes):
            cy = side * (0.35 + 0.1 * c + 0.03 * b)
            cx = side * (0.30 + 0.15 * c - 0.02 * b)
            ry, rx = side * (0.06 + 0.01 * c), side * (0.09 + 0.01 * b)
            lab[b][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = c
    return img, torch.from_numpy(lab)
