"""`train_model`: the lightning branch of the reference's `train_general` (/root/reference/pl_utils/pl_train.py:6-44)."""
import os
import shutil

import pytorch_lightning as pl
import torch
from pytorch_lightning.callbacks import ModelCheckpoint


def train_model(lightn_model, checkpoint_path, del_pre_files=True, device=None, num_epochs=11):
    """Empty `checkpoint_path` (when asked), train for `num_epochs`, keep the checkpoint with the best `val_acc`."""
    os.makedirs(checkpoint_path, exist_ok=True)
    if del_pre_files:
        for name in os.listdir(checkpoint_path):
            path = os.path.join(checkpoint_path, name)
            if os.path.isfile(path) or os.path.islink(path):
                os.unlink(path)
            elif os.path.isdir(path):
                shutil.rmtree(path)
    checkpoint_callback = ModelCheckpoint(monitor="val_acc", filepath=checkpoint_path, mode="max")
    num_gpus = 1 if (device is not None and torch.device(device).type == 'cuda') else None
    trainer = pl.Trainer(gpus=num_gpus, max_epochs=num_epochs, default_root_dir=checkpoint_path,
                         checkpoint_callback=checkpoint_callback, val_check_interval=1)
    trainer.fit(lightn_model)
    return trainer


from .pl_test import test_model  # noqa: E402,F401  (kept importable from here; its home is pl_utils/pl_test.py as in the reference)
