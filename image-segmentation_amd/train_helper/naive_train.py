"""Plain training / validation loop with the semantics of the reference's `naive_train`
(/root/reference/perlosstrain/naive_train.py:389-454 train, :456-518 evaluate, :567-626 naive_train,
:37-58 save_checkpoint; fixed-lr variant /root/reference/detect_region/naive_train.py:190-191).

Kept from the reference: a NEW Adam is created every epoch (its moment estimates restart); with `edge_mode`
the learning rate is overridden by the epoch (1e-2 below epoch 10, 1e-3 after, 1e-5 after epoch 20); the summed
loss of the criterion list is back-propagated; validation runs in eval mode under no_grad every `valid_after`
epochs; a checkpoint `{'epoch', 'arch', 'state_dict'}` is written when the validation loss improves, replacing
the previous one.  Not kept: tqdm progress bars, matplotlib curves.

`fused=True` (default on CUDA) re-homes the parameters into one flat buffer and steps them with the fused Adam
kernel (same update rule as torch.optim.Adam); `fused=False` uses torch.optim.Adam on `model.parameters()`.
"""
import os

import torch
from torch import optim
from torch.utils.data import DataLoader


class AverageMeter:
    """Running average weighted by batch size (reference :19-35)."""

    def __init__(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def save_checkpoint(state, is_best, save_model_dir, del_pre_model=True):
    """Write `checkpoint_epoch{E}_{arch}.pth.tar` when `is_best`, deleting earlier epoch files first (reference :37-58)."""
    filename = os.path.join(save_model_dir, f'checkpoint_epoch{state["epoch"]}_{state["arch"]}.pth.tar')
    if is_best:
        if del_pre_model:
            for name in os.listdir(save_model_dir):
                if "epoch" in name:
                    os.unlink(os.path.join(save_model_dir, name))
        torch.save(state, filename)
        return filename
    return None


def _outputs(model, images):
    out = model(images)
    return out[0] if isinstance(out, tuple) else out       # multi-output research variants return (logits, extras)


def _epoch_lr(epoch, lr, edge_mode):
    if edge_mode:
        lr = 1e-2 if epoch < 10 else 1e-3
        if epoch > 20:
            lr = 1e-5
    return lr


def train(train_loader, model, criterion, epoch, num_epochs, device, batch_size, optimizer=None, lr=1e-4,
          edge_mode=False, fused=None, log=None):
    """One epoch; returns the batch-size-weighted mean of the summed loss."""
    model.train()
    losses = AverageMeter()
    lr = _epoch_lr(epoch, lr, edge_mode)
    if fused is None:
        fused = torch.device(device).type == "cuda" and hasattr(model, "set_compute_dtype")
    if optimizer is None:
        if fused:
            from segk.flat import FlatModel
            from segk.optim import Adam
            flat = getattr(model, "_segk_flat", None)
            if flat is None:
                flat = model._segk_flat = FlatModel(model)
            optimizer = Adam(flat, lr=lr)                   # fresh moments every epoch, as the reference's new optim.Adam
        else:
            optimizer = optim.Adam(model.parameters(), lr=lr)
    for i, (images, target) in enumerate(train_loader):
        images, target = images.to(device), target.to(device)
        optimizer.zero_grad()
        outputs = _outputs(model, images)
        aggr_loss = 0
        for loss in criterion:
            aggr_loss = aggr_loss + loss(outputs, target)
        losses.update(float(aggr_loss.detach()), images.size(0))
        aggr_loss.backward()
        optimizer.step()
        if log is not None:
            log(f"[TRAIN] epoch {epoch + 1}/{num_epochs} batch {i}: loss {losses.val:.4f} / {losses.avg:.4f} (epoch avg)")
    return losses.avg


def evaluate(val_loader, model, criterion, val_acc_func, epoch, num_epochs, device, log=None):
    """Validation pass: (mean loss, mean of the per-batch accuracy values)."""
    model.eval()
    losses, accs = AverageMeter(), AverageMeter()
    with torch.no_grad():
        for images, target in val_loader:
            images, target = images.to(device), target.to(device)
            outputs = _outputs(model, images)
            aggr_loss = 0
            for loss in criterion:
                aggr_loss = aggr_loss + loss(outputs, target)
            losses.update(float(aggr_loss), images.size(0))
            accs.update(float(val_acc_func(outputs, target)))   # the metric already averages over the batch
    if log is not None:
        log(f"[VALID] epoch {epoch + 1}/{num_epochs}: loss {losses.avg:.4f} dice {accs.avg:.4f}")
    return losses.avg, accs.avg


def naive_train(train_set, valid_set, model, criterion, val_acc_func, num_epochs, batch_size, device, out_path, kwargs,
                experiment_name="", valid_after=4, lr=1e-4, edge_mode=False, use_saved=False, fused=None, log=None):
    """Train `model` in place; returns {"train_loss": [...], "val_loss": [...], "val_acc": [...], "checkpoint": path}."""
    os.makedirs(out_path, exist_ok=True)
    if use_saved:                                                 # resume from the newest checkpoint in out_path
        names = sorted(n for n in os.listdir(out_path) if "epoch" in n)
        if names:
            state = torch.load(os.path.join(out_path, names[-1]), map_location="cpu", weights_only=True)
            model.load_state_dict(state["state_dict"])
    best_loss = 1000
    metrics = {"train_loss": [], "val_loss": [], "val_acc": [], "checkpoint": None}
    train_loader = DataLoader(train_set, batch_size=batch_size, **kwargs)
    val_loader = DataLoader(valid_set, batch_size=batch_size, shuffle=False, **kwargs)
    for epoch in range(num_epochs):
        metrics["train_loss"].append(train(train_loader, model, criterion, epoch, num_epochs, device, batch_size, lr=lr,
                                           edge_mode=edge_mode, fused=fused, log=log))
        if not epoch % valid_after:
            val_loss, val_acc = evaluate(val_loader, model, criterion, val_acc_func, epoch, num_epochs, device, log=log)
            metrics["val_loss"].append(val_loss)
            metrics["val_acc"].append(val_acc)
            is_best = val_loss < best_loss
            best_loss = min(val_loss, best_loss)
            saved = save_checkpoint({'epoch': epoch + 1, 'arch': experiment_name,
                                     'state_dict': {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}},
                                    is_best, out_path)
            metrics["checkpoint"] = saved or metrics["checkpoint"]
    return metrics
