"""Synthetic 2-D segmentation slices with the statistics the reference's readers produce
(/root/reference/roi_train/read_roi.py:141-142,183-187: z-scored grey slice, replicated to 3 channels; int64 label
map with a few small structures; some slices without foreground)."""
import torch
from torch.utils.data import Dataset


class SyntheticSlices(Dataset):
    def __init__(self, num_img=64, side=384, n_classes=4, in_channels=3, seed=20):
        self.n, self.side, self.c, self.in_ch, self.seed = num_img, side, n_classes, in_channels, seed
        yy, xx = torch.meshgrid(torch.arange(side), torch.arange(side), indexing="ij")
        self._yy, self._xx = yy.float(), xx.float()

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        s = self.side
        target = torch.zeros(s, s, dtype=torch.int64)
        grey = torch.randn(1, s, s, generator=g)
        if i % 8 != 7:                                           # 1 in 8 slices is all background
            for c in range(1, self.c):
                cy, cx = s * (0.4 + 0.07 * c + 0.02 * float(torch.rand(1, generator=g))), s * (0.3 + 0.12 * c)
                ry, rx = s * 0.06, s * 0.08
                inside = ((self._yy - cy) / ry) ** 2 + ((self._xx - cx) / rx) ** 2 <= 1.0
                target[inside] = c
                grey[0][inside] += 1.0 + 0.5 * c                 # the structures are visible in the image
        grey = (grey - grey.mean()) / grey.std()
        return grey.expand(self.in_ch, s, s).contiguous(), target
