"""`train_general`: split a dataset, train, validate, test -- the wrapper the reference's scripts call
(/root/reference/detect_region/common.py:18-116, /root/reference/perlosstrain/common.py:17-108; call site
/root/reference/train_midlslice.py:30-33).  Same signature and defaults as the detect_region variant.

Differences, on purpose: figure export after the test pass (matplotlib) is replaced by the returned metrics
(`model.segk_metrics`); the default network is `UNetVgg()`.  The pytorch-lightning branch (`is_naive_train=False`)
runs on the installed pytorch_lightning, or on the minimal stand-in of segk_compat.lightning when that is absent.

Two environment knobs exist so that the reference's entry scripts, which hard-code 2000 epochs and the authors' home
directories (/root/reference/train_midlslice.py:16-17,24), can be run unedited as a check: `SEGK_MAX_EPOCHS=n` caps
`num_epochs`, `SEGK_PATH_REMAP=old_prefix=new_prefix` re-roots `out_path`.
"""
import os

import torch
from torch.utils.data import DataLoader

from loss.dice import dice_coefficient, dice_loss_sq
from models.unetvgg import UNetVgg
from train_helper.naive_train import evaluate, naive_train


def train_general(dataset, out_path, expt_name="", val_set_perc=0.15, test_set_perc=.05, device=None,
                  num_epochs=1, is_naive_train=False, mode=None, val_set=None, test_set=None, valid_after=4,
                  net_model=None, batch_size=3, test_batch_size=7, loss_criterion=None, val_acc_func=None,
                  sgm_train=True, lr=1e-4, use_saved=False):
    if os.environ.get("SEGK_MAX_EPOCHS"):
        num_epochs = min(num_epochs, int(os.environ["SEGK_MAX_EPOCHS"]))
    if os.environ.get("SEGK_PATH_REMAP"):
        old, new = os.environ["SEGK_PATH_REMAP"].split("=", 1)
        if str(out_path).startswith(old):
            out_path = new + str(out_path)[len(old):]
    if mode is None:
        val_size = int(val_set_perc * len(dataset))
        test_size = int(test_set_perc * len(dataset))
        train_size = len(dataset) - test_size - val_size
        train_set, val_set, test_set = torch.utils.data.random_split(dataset, [train_size, val_size, test_size])
    else:
        train_set = dataset
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    device = torch.device(device)
    kwargs = {}
    if net_model is None:
        net_model = UNetVgg()
    if loss_criterion is None:
        loss_criterion = [dice_loss_sq]
    if val_acc_func is None:
        val_acc_func = dice_coefficient
    if not is_naive_train:
        from pl_utils.pl_mod import SegmModelPL
        from pl_utils.pl_test import test_model
        from pl_utils.pl_train import train_model
        lightn_mod = SegmModelPL(net_model, train_set, val_set, test_set, batch_size, test_batch_size, loss_criterion=loss_criterion,
                                 val_acc_func=val_acc_func, lr=lr, experiment_name=expt_name).to(device)
        trainer = train_model(lightn_mod, out_path, del_pre_files=True, device=device, num_epochs=num_epochs)
        metrics = dict(trainer.callback_metrics)
        if test_set_perc and test_set is not None and len(test_set):
            test_model(lightn_mod, out_path, trainer)
            metrics.update(trainer.callback_metrics)
        lightn_mod.segk_metrics = metrics
        return lightn_mod
    net_model = net_model.to(device)
    metrics = naive_train(train_set, val_set, net_model, loss_criterion, val_acc_func, num_epochs, batch_size, device, out_path,
                          kwargs, experiment_name=expt_name, valid_after=valid_after, lr=lr, use_saved=use_saved)
    if test_set_perc and test_set is not None and len(test_set) and sgm_train:
        loader = DataLoader(test_set, batch_size=test_batch_size, shuffle=False, **kwargs)
        metrics["test_loss"], metrics["test_acc"] = evaluate(loader, net_model, loss_criterion, val_acc_func, num_epochs - 1,
                                                             num_epochs, device)
    net_model.segk_metrics = metrics
    return net_model
