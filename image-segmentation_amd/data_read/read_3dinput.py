"""Stand-in for the reference's 3-D volume reader (`Meniscus3DMidSlice(data_loc, num_img=None, in_channels=1)`, call
site /root/reference/train_midlslice.py:25): yields `(volume [in_channels, D, H, W] float32, mid_index int64 < 80)`
from a seeded generator instead of the private MRI files under `data_loc`.  `SEGK_SYNTH_3D=n,depth,side` overrides the
number and size of the generated volumes (the reference's script passes neither, /root/reference/train_midlslice.py:25)."""
import os

import torch
from torch.utils.data import Dataset


class Meniscus3DMidSlice(Dataset):
    def __init__(self, data_loc=None, num_img=None, in_channels=1, depth=64, side=256, seed=20):
        if os.environ.get("SEGK_SYNTH_3D"):
            n_env, depth, side = (int(v) for v in os.environ["SEGK_SYNTH_3D"].split(","))
            num_img = num_img or n_env
        self.data_loc, self.n, self.in_ch, self.depth, self.side, self.seed = data_loc, num_img or 16, in_channels, depth, side, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 7919 + i)
        mid = int(torch.randint(self.depth // 4, min(80, 3 * self.depth // 4), (1,), generator=g))
        vol = torch.randn(self.in_ch, self.depth, self.side, self.side, generator=g)
        vol[:, mid] += 1.5                                       # the labelled slice is brighter
        return vol, torch.tensor(mid, dtype=torch.int64)
