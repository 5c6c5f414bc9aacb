"""Middle-slice detector: a learned 1->3 channel stem in front of the 3-D DenseNet-169, MI355X-native.

Same constructor signature, attribute names and state_dict layout as /root/reference/models/resnt.py:8-45
(DetectMid).  `DetectSide` (:48-85) wraps torchvision's 2-D resnet18, which is not part of this build.
"""
from torch import nn

from models.densenet3d import generate_model as dg
from segk.engine3d import Runtime3D
from segk.net import SegkNet


class DetectMid(SegkNet):
    _runtime_cls = Runtime3D
    _FIRST = dict(kernel=(3, 7, 7), stride=(1, 2, 2), pad=(1, 3, 3))

    def __init__(self, pretrained=False, in_classes=1, out_classes=1, patch_train=False, if_convtrans=False, if_regr=False,
                 model_depth=169):
        super().__init__()
        self.rsnet = dg(model_depth)
        self.patch_train = False
        self.first = nn.Conv3d(in_classes, 3, kernel_size=self._FIRST["kernel"], stride=self._FIRST["stride"],
                               padding=self._FIRST["pad"], bias=False)
        self.last = nn.Linear(1000, 80)
        self.lastregr = nn.Linear(80, out_classes)   # last regression layer
        self.if_regr = if_regr

    def _forward_host(self, x):
        out = self.last(self.rsnet(self.first(x)))
        if self.if_regr:
            out = self.lastregr(out).squeeze(dim=-1)
        return out

    def _emit(self, rt, x):
        v, D = rt.input_volume(x)
        y, D = rt.conv_patches(v, D, "first.weight", self._FIRST["kernel"], self._FIRST["stride"], self._FIRST["pad"], need_dx=False)
        feats, fb = self.rsnet.emit_features(rt, y, D, "rsnet.", need_dx=True)
        layers = [("rsnet.classifier.weight", "rsnet.classifier.bias"), ("last.weight", "last.bias")]
        if self.if_regr:
            layers.append(("lastregr.weight", "lastregr.bias"))
        else:
            rt.unused(["lastregr.weight", "lastregr.bias"])
        return rt.linear_tail(feats, fb, layers, squeeze_last=self.if_regr)


class DetectSide(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("DetectSide wraps torchvision.models.resnet18 (2-D), which this build does not provide")
