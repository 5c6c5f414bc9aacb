"""Minimal stand-in for the absent `torchvision` package (test infrastructure only).

Only what /root/reference/models/unetvgg.py:4,70,177 touches: `models.vgg16_bn(pretrained).features`.
"""
from . import models  # noqa: F401
