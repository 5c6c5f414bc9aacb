"""`vgg16_bn` layer list (VGG configuration D with BatchNorm): 13 conv3x3+BN+ReLU, 5 max-pools,
44 feature modules in total, the length the slices at models/unetvgg.py:72-79 assume."""
from torch import nn

_CFG_D = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


class _VGG(nn.Module):
    def __init__(self):
        super().__init__()
        layers, c = [], 3
        for v in _CFG_D:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)


def vgg16_bn(pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained weights need a download; unavailable offline")
    return _VGG()
