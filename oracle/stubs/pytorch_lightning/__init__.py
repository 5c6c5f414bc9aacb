"""Stand-in for `pytorch_lightning`: the reference's Nested_UNet only uses LightningModule as an
nn.Module base class (/root/reference/models/unetplusplus.py:27)."""
from torch import nn


class LightningModule(nn.Module):
    pass
