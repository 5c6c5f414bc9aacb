"""Minimal stand-in for the pytorch-lightning 0.8-era API surface the reference's harness touches
(/root/reference/pl_utils/pl_train.py:33-43 `ModelCheckpoint(monitor, filepath, mode)`,
`Trainer(gpus, max_epochs, default_root_dir, checkpoint_callback, val_check_interval)`, `.fit`, `.test`;
/root/reference/pl_utils/pl_mod.py:6 `LightningModule` with training_step / validation_step /
validation_epoch_end / test_step / test_epoch_end / configure_optimizers / *_dataloader).

`install()` registers it as `pytorch_lightning` ONLY when the real package cannot be imported.  It is a single-device
loop: no loggers, no early stopping, no distributed strategy -- the data-parallel path of this build is
segk.flat.FlatModel.
"""
import os
import sys
import types

import torch
from torch import nn


class LightningModule(nn.Module):
    def prepare_data(self):
        pass

    def validation_epoch_end(self, outputs):
        return {}

    def test_epoch_end(self, outputs):
        return {}


class ModelCheckpoint:
    """Keeps the single best checkpoint `epoch=<E>.ckpt` (dict with 'epoch', 'state_dict', the monitored value) in `filepath`."""

    def __init__(self, filepath=None, monitor="val_loss", mode="min", **_):
        self.filepath, self.monitor, self.mode = filepath, monitor, mode
        self.best, self.best_path = None, None

    def update(self, module, epoch, metrics):
        if self.filepath is None or self.monitor not in metrics:
            return
        value = float(metrics[self.monitor])
        better = self.best is None or (value > self.best if self.mode == "max" else value < self.best)
        if not better:
            return
        os.makedirs(self.filepath, exist_ok=True)
        if self.best_path and os.path.exists(self.best_path):
            os.unlink(self.best_path)
        self.best, self.best_path = value, os.path.join(self.filepath, f"epoch={epoch}.ckpt")
        torch.save({"epoch": epoch, self.monitor: value,
                    "state_dict": {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}}, self.best_path)


class Trainer:
    def __init__(self, gpus=None, max_epochs=1, default_root_dir=None, checkpoint_callback=None, val_check_interval=1, **_):
        self.device = torch.device("cuda" if gpus else "cpu")
        self.max_epochs, self.root = max_epochs, default_root_dir
        self.checkpoint_callback = checkpoint_callback if isinstance(checkpoint_callback, ModelCheckpoint) else None
        self.callback_metrics = {}

    def _to(self, batch):
        return tuple(b.to(self.device) if torch.is_tensor(b) else b for b in batch)

    def fit(self, module):
        module.to(self.device)
        module.prepare_data()
        optimizer = module.configure_optimizers()
        for epoch in range(self.max_epochs):
            module.train()
            for i, batch in enumerate(module.train_dataloader()):
                optimizer.zero_grad()
                out = module.training_step(self._to(batch), i)
                loss = out["loss"] if isinstance(out, dict) else out
                loss.backward()
                optimizer.step()
                self.callback_metrics["loss"] = float(loss.detach())
            loader = module.val_dataloader()
            if loader is not None and len(loader):
                module.eval()
                with torch.no_grad():
                    outs = [module.validation_step(self._to(batch), i) for i, batch in enumerate(loader)]
                    res = module.validation_epoch_end(outs) or {}
                self.callback_metrics.update({k: float(v) for k, v in res.items() if not isinstance(v, dict)})
                if self.checkpoint_callback is not None:
                    self.checkpoint_callback.update(module, epoch, self.callback_metrics)
        return 1

    def test(self, module=None):
        module.to(self.device)
        module.eval()
        with torch.no_grad():
            outs = [module.test_step(self._to(batch), i) for i, batch in enumerate(module.test_dataloader())]
            res = module.test_epoch_end(outs) or {}
        self.callback_metrics.update({k: float(v) for k, v in res.items() if not isinstance(v, dict)})
        return res


def install():
    """Make `import pytorch_lightning` resolve to this stand-in unless the real package is importable."""
    try:
        import pytorch_lightning  # noqa: F401
        return False
    except ImportError:
        pass
    me = sys.modules[__name__]
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule, pl.Trainer, pl.__segk_stand_in__ = LightningModule, Trainer, True
    callbacks = types.ModuleType("pytorch_lightning.callbacks")
    callbacks.ModelCheckpoint = ModelCheckpoint
    pl.callbacks = callbacks
    sys.modules["pytorch_lightning"], sys.modules["pytorch_lightning.callbacks"] = pl, callbacks
    return me is not None
