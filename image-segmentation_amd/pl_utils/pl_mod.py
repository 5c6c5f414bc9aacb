"""`SegmModelPL`: the LightningModule the reference wraps its networks in
(/root/reference/pl_utils/pl_mod.py:6-119): summed (optionally weighted) loss list for training, validation and
test steps, the accuracy function on validation/test, Adam(lr) from `configure_optimizers`, the three dataloaders."""
import pytorch_lightning as pl
import torch
from torch import optim
from torch.utils.data import DataLoader


class SegmModelPL(pl.LightningModule):
    def __init__(self, model, train_set, val_set, test_set, batch_size, test_batch_size, loss_criterion: list,
                 loss_sum_weights: list = None, val_acc_func=None, optimizer=optim.Adam, lr=1e-4, experiment_name=""):
        super().__init__()
        self.model = model
        self.train_set, self.val_set, self.test_set = train_set, val_set, test_set
        self.batch_size, self.test_batch_size = batch_size, test_batch_size
        self.loss_criterion, self.loss_sum_weights, self.val_acc_func = loss_criterion, loss_sum_weights, val_acc_func
        self.optimizer, self.lr, self.experiment_name = optimizer, lr, experiment_name
        self.kwargs = {}

    def forward(self, x):
        return self.model(x)

    def _loss(self, pred, segm):
        total = 0
        for i, loss in enumerate(self.loss_criterion):
            each = loss(pred, segm)
            if self.loss_sum_weights is not None:
                each = self.loss_sum_weights[i] * each
            total = total + each
        return total

    def training_step(self, batch, batch_nb):
        img, segm = batch
        loss = self._loss(self.forward(img), segm)
        return {'loss': loss, 'log': {f'train_loss_{self.experiment_name}': loss}}

    def validation_step(self, batch, batch_nb):
        img, segm = batch
        pred = self.forward(img)
        return {'val_loss_epoch': self._loss(pred, segm), 'val_acc_': self.val_acc_func(pred, segm)}

    def validation_epoch_end(self, outputs):
        avg_loss = torch.stack([torch.as_tensor(x['val_loss_epoch']) for x in outputs]).mean()
        avg_acc = torch.stack([torch.as_tensor(x['val_acc_']) for x in outputs]).mean()
        logs = {f'avg_val_loss_{self.experiment_name}': avg_loss, f'val_acc_avg_{self.experiment_name}': avg_acc}
        return {'val_loss': avg_loss, 'val_acc': avg_acc, 'log': logs}

    def test_step(self, batch, batch_nb):
        img, segm = batch
        pred = self.forward(img)
        return {'test_loss': self._loss(pred, segm), 'test_acc_': self.val_acc_func(pred, segm)}

    def test_epoch_end(self, outputs):
        avg_loss = torch.stack([torch.as_tensor(x['test_loss']) for x in outputs]).mean()
        avg_acc = torch.stack([torch.as_tensor(x['test_acc_']) for x in outputs]).mean()
        logs = {f'test_loss_{self.experiment_name}': avg_loss, f'test_acc_{self.experiment_name}': avg_acc}
        return {'test_loss': avg_loss, 'test_acc': avg_acc, 'log': logs, 'progress_bar': logs}

    def configure_optimizers(self):
        return self.optimizer(self.parameters(), self.lr)

    def train_dataloader(self):
        return DataLoader(self.train_set, batch_size=self.batch_size, shuffle=True, **self.kwargs)

    def val_dataloader(self):
        return DataLoader(self.val_set, batch_size=self.batch_size, shuffle=False, **self.kwargs)

    def test_dataloader(self):
        return DataLoader(self.test_set, batch_size=self.test_batch_size, shuffle=False, **self.kwargs)
