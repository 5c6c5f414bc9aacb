"""3-D DenseNet-BC classifier, MI355X-native.

Same constructor signature, module names and state_dict layout as the reference
(/root/reference/models/densenet3d.py: _DenseLayer :10-41, _DenseBlock :44-52, _Transition :55-68, DenseNet
:71-159, generate_model :162-186).  The torch.nn layers only hold the parameters (and serve CPU tensors); CUDA
tensors run on the segk tape (segk/engine3d.py): the strided stem convolution as a patch gather + 1x1 MFMA
convolution, BatchNorm3d/ReLU as channel-wise streams over the slice-major buffer, 1x1x1 and 3x3x3 convolutions on
the implicit-GEMM kernels (the latter as three depth taps), and every torch.cat of :41 written in place into the
block's concat buffer.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

from segk.engine3d import Runtime3D
from segk.net import SegkNet
from segk.ops import Act


def _bn_relu_conv(c_in, c_out, kernel, suffix="", **conv_kw):
    """The pre-activation unit every part of the network is made of, as (name, module) pairs: BatchNorm3d -> ReLU ->
    bias-free Conv3d.  `suffix` numbers the names inside a dense layer (norm1/relu1/conv1, norm2/...)."""
    return [("norm" + suffix, nn.BatchNorm3d(c_in)), ("relu" + suffix, nn.ReLU(inplace=True)),
            ("conv" + suffix, nn.Conv3d(c_in, c_out, kernel_size=kernel, stride=1, bias=False, **conv_kw))]


class _Growth(nn.Sequential):
    """One dense layer: a 1x1x1 bottleneck to bn_size * growth channels, a 3x3x3 convolution to `growth` new channels,
    which are appended to the input (reference :10-41).  State-dict names norm1/conv1/norm2/conv2."""

    def __init__(self, c_in, growth, bn_size, drop_rate):
        mid = bn_size * growth
        super().__init__(OrderedDict(_bn_relu_conv(c_in, mid, 1, "1") + _bn_relu_conv(mid, growth, 3, "2", padding=1)))
        self.drop_rate = drop_rate

    def forward(self, x):
        fresh = nn.Sequential.forward(self, x)
        if self.drop_rate > 0:
            fresh = F.dropout(fresh, p=self.drop_rate, training=self.training)
        return torch.cat((x, fresh), dim=1)


def _dense_block(n_layers, c_in, bn_size, growth, drop_rate):
    """`denselayer1..n`, layer k reading the c_in + (k-1) * growth channels gathered so far (reference :44-52)."""
    return nn.Sequential(OrderedDict((f"denselayer{k + 1}", _Growth(c_in + k * growth, growth, bn_size, drop_rate))
                                     for k in range(n_layers)))


def _transition(c_in, c_out):
    """Channel halving + 2x2x2 average pool between dense blocks (reference :55-68)."""
    return nn.Sequential(OrderedDict(_bn_relu_conv(c_in, c_out, 1) + [("pool", nn.AvgPool3d(kernel_size=2, stride=2))]))


_INIT = {nn.Conv3d: lambda m: nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu'),
         nn.BatchNorm3d: lambda m: (nn.init.ones_(m.weight), nn.init.zeros_(m.bias)),
         nn.Linear: lambda m: nn.init.zeros_(m.bias)}


class DenseNet(SegkNet):
    """Densenet-BC; arguments as in the reference (:83-93)."""
    _runtime_cls = Runtime3D

    def __init__(self, n_input_channels=3, conv1_t_size=7, conv1_t_stride=1, no_max_pool=False, growth_rate=32,
                 block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4, drop_rate=0, num_classes=1000):
        super().__init__()
        self.stem = dict(kernel=(conv1_t_size, 7, 7), stride=(conv1_t_stride, 2, 2), pad=(conv1_t_size // 2, 3, 3))
        self.no_max_pool, self.growth_rate, self.block_config, self.drop_rate = no_max_pool, growth_rate, tuple(block_config), drop_rate
        # the whole feature extractor as one name -> module table (names and order are the checkpoint layout)
        table = [("conv1", nn.Conv3d(n_input_channels, num_init_features, kernel_size=self.stem["kernel"], stride=self.stem["stride"],
                                     padding=self.stem["pad"], bias=False)),
                 ("norm1", nn.BatchNorm3d(num_init_features)),
                 ("relu1", nn.ReLU(inplace=True))]
        if not no_max_pool:
            table.append(("pool1", nn.MaxPool3d(kernel_size=3, stride=2, padding=1)))
        self.block_io = []                       # (channels in, channels out) of every dense block
        width = num_init_features
        last = len(self.block_config) - 1
        for b, n_layers in enumerate(self.block_config):
            grown = width + n_layers * growth_rate
            table.append((f"denseblock{b + 1}", _dense_block(n_layers, width, bn_size, growth_rate, drop_rate)))
            self.block_io.append((width, grown))
            width = grown
            if b != last:
                table.append((f"transition{b + 1}", _transition(width, width // 2)))
                width //= 2
        table.append(("norm5", nn.BatchNorm3d(width)))
        self.features = nn.Sequential(OrderedDict(table))
        self.classifier = nn.Linear(width, num_classes)
        for m in self.modules():
            init = _INIT.get(type(m))
            if init is not None:
                init(m)

    # -------------------------------------------------------------- CPU tensors: plain torch.nn
    def _forward_host(self, x):
        features = self.features(x)
        out = F.relu(features, inplace=True)
        out = F.adaptive_avg_pool3d(out, output_size=(1, 1, 1)).view(features.size(0), -1)
        return self.classifier(out)

    # -------------------------------------------------------------- CUDA tensors: segk tape
    def emit_features(self, rt, v, D, prefix, need_dx):
        """Everything up to the pooled feature vector.  v: Var over [N*D,H,W,8] whose first channels are the input.
        -> (fp32 [N, C] features, their backward hook)"""
        if self.drop_rate > 0 and self.training:
            raise NotImplementedError("dropout inside the dense layers is not implemented on the HIP path (reference default drop_rate=0)")
        f = prefix + "features."
        N = v.act.N // D
        y, D = rt.conv_patches(v, D, f + "conv1.weight", self.stem["kernel"], self.stem["stride"], self.stem["pad"], need_dx, stats=True)
        H, W = y.act.H, y.act.W
        c_in, c_out = self.block_io[0]
        if self.no_max_pool:
            feat = rt.new(N * D, H, W, c_out)
            rt.bn_relu(y, f + "norm1", out=Act(feat.buf, 0, c_in))
        else:
            a = rt.bn_relu(y, f + "norm1")
            Do, Ho, Wo = (D - 1) // 2 + 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1
            feat = rt.new(N * Do, Ho, Wo, c_out)
            _, D = rt.maxpool3d(a, D, out=Act(feat.buf, 0, c_in))
        for i, num_layers in enumerate(self.block_config):
            c_in, c_out = self.block_io[i]
            x = rt.dense_block(feat, c_in, D, f"{f}denseblock{i + 1}.", num_layers, self.growth_rate)
            if i != len(self.block_config) - 1:
                nc_in, nc_out = self.block_io[i + 1]
                if D < 2 or feat.H < 2 or feat.W < 2:
                    raise RuntimeError("input volume too small for the transitions' 2x2x2 average pools")
                nxt = rt.new(N * (D // 2), feat.H // 2, feat.W // 2, nc_out)
                _, D = rt.transition(x, D, f"{f}transition{i + 1}.", out=Act(nxt.buf, 0, nc_in))
                feat = nxt
        return rt.pooled_features(x, D, f + "norm5")

    def _emit(self, rt, x):
        v, D = rt.input_volume(x)
        feats, fb = self.emit_features(rt, v, D, "", need_dx=False)
        return rt.linear_tail(feats, fb, [("classifier.weight", "classifier.bias")])


def generate_model(model_depth, **kwargs):
    assert model_depth in [121, 169, 201, 264]
    config = {121: (6, 12, 24, 16), 169: (6, 12, 32, 32), 201: (6, 12, 48, 32), 264: (6, 12, 64, 48)}[model_depth]
    return DenseNet(num_init_features=64, growth_rate=32, block_config=config, **kwargs)
